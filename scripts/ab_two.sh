#!/bin/bash
# bench batch A/B of every ab/v_*.so, REPS times (4096 and 1024 blocks), then per-class decode times of the last one
TAG=${1:-ab2}; REPS=${2:-3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in $(seq $REPS); do
  for f in ab/v_*.so; do
    cp $f $L
    echo -n "$(basename $f .so) " | tee -a $OUT/ab.txt
    timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"decode_GiBs_per_gpu[^,]*' | tee -a $OUT/ab.txt
  done
done
cp /tmp/keep.so $L
