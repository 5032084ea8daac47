#!/bin/bash
# encode rate of one build (ab/v_$1.so) over a list of K4LZ4_SPLIT_PCT values.  Usage: scripts/variants_split.sh name tag "44 48 52"
V=$1; TAG=${2:-split}; OUT=gpurun_out/$TAG; mkdir -p $OUT
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so; cp ab/v_$V.so $L
for s in $3; do
  echo -n "$V split=$s " | tee -a $OUT/split.txt
  K4LZ4_SPLIT_PCT=$s timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*' | tee -a $OUT/split.txt
done
cp /tmp/keep.so $L
