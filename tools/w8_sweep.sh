export WINO=1
for s in "64 32 32 384 384" "64 64 64 192 192"; do
  timeout -k 10 120 python tools/conv_bench.py $s 3 8,9,8,9,7,8,9 20 2>&1 | grep -E "^shape|n/a"
done
