#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
bash scripts/r2c.sh $1 | grep -E "block (0|2|4|9)|launch" 
L=k4os/compression/lz4_amd/libk4lz4.so
for v in $2; do
  cp ab/$v.so $L
  echo -n "$v " | tee -a $OUT/variants.txt
  timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/variants.txt
done
