export WINO=1
for s in "64 64 64 192 192" "64 32 32 384 384" "64 64 64 384 384" "64 16 16 576 576" "64 32 32 768 384"; do
 for g in 1 0 2 4 99; do
  echo "ngroup=$g"; ND_NGROUP=$g timeout -k 10 120 python tools/conv_bench.py $s 3 5 20 2>&1 | grep shape
 done
done
echo "8x8 v2"; for g in 1 0 99; do ND_NGROUP=$g timeout -k 10 120 python tools/conv_bench.py 64 8 8 768 768 3 2 20 2>&1 | grep shape; done
