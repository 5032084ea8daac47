#!/bin/bash
# round 6: every ab/v_*.so on one box: bench batch at 4096 / 1024 blocks (REPS times) and the pair probe of the second-generation decoder
TAG=${1:-r6var}; REPS=${2:-2}; PROBE=${3:-1}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for f in ab/v_*.so; do
  cp $f $L
  for r in $(seq $REPS); do
    for nb in 4096 1024; do
      echo -n "$(basename $f .so) blocks=$nb " | tee -a $OUT/bench.txt
      timeout 300 python bench.py --steps 20 --warmup 3 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/bench.txt
    done
  done
  if [ "$PROBE" = 1 ]; then
    echo "== $(basename $f .so) 4096 blocks" >> $OUT/probe.txt
    timeout 300 python scripts/x_probe.py 2>&1 | grep -v amdgpu >> $OUT/probe.txt
    echo "== $(basename $f .so) 256 blocks" >> $OUT/probe.txt
    K4_BLOCKS=256 timeout 300 python scripts/x_probe.py 2>&1 | grep -v amdgpu >> $OUT/probe.txt
  fi
done
cp /tmp/keep.so $L
