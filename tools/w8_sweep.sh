for s in "64 1024 6 64" "64 256 9 64" "64 64 12 64"; do
  for w in 4 8; do ND_ATTN_WAVES=$w timeout -k 10 120 python tools/attn_bench.py $s 2>&1 | grep attention; done
done
