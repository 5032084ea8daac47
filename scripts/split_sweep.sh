#!/bin/bash
# bench encode rate against the share of the batch that goes to the LDS-table kernel (K4LZ4_SPLIT_PCT)
OUT=gpurun_out/${1:-split}
mkdir -p $OUT
for pct in ${2:-40 44 48 52 56 62 70}; do
  echo -n "split $pct " | tee -a $OUT/split.txt
  K4LZ4_SPLIT_PCT=$pct timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*' | tee -a $OUT/split.txt
done
