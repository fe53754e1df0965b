export WINO=1
for s in "64 32 32 384 384" "64 64 64 192 192" "64 16 16 576 576" "64 8 8 768 768" "64 64 64 384 192"; do
  timeout -k 10 120 python tools/conv_bench.py $s 3 7,8,10 20 2>&1 | grep -E "^shape|n/a|diff" | sort -u
done
