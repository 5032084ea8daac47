#!/bin/bash
# round 6: ab/v_*.so on the bench batch, REPS times alternating: encode / decode rate and the median encode launch
TAG=${1:-r6ab2}; REPS=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in $(seq $REPS); do for f in ab/v_*.so; do cp $f $L; echo -n "$(basename $f .so) " | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"encode_GiBs_per_gpu[^,]*,[^,]*\|"median_launch_ms": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/ab.txt; echo | tee -a $OUT/ab.txt; done; done
cp /tmp/keep.so $L
