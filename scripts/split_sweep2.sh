#!/bin/bash
# encode call against the LDS-table kernel's share of the batch's cost (K4LZ4_COST_PCT; the floor of one residency stays): bench batch
TAG=${1:-split}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for p in 36 40 44 48 50; do
  echo -n "K4LZ4_COST_PCT=$p " | tee -a $OUT/split.txt
  K4LZ4_COST_PCT=$p timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*' | tee -a $OUT/split.txt
done
for f in 6 7 8; do
  echo -n "K4LZ4_LDS_FLOOR=$f K4LZ4_COST_PCT=40 " | tee -a $OUT/split.txt
  K4LZ4_LDS_FLOOR=$f K4LZ4_COST_PCT=40 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*' | tee -a $OUT/split.txt
done
