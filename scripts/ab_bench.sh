#!/bin/bash
# A/B two builds of libk4lz4.so on the same GPU box (box-to-box spread is ~2 %): ab/A.so, ab/B.so, alternating.
# Usage: scripts/ab_bench.sh [reps]
REPS=${1:-3}
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in $(seq $REPS); do
  for v in A B; do
    cp ab/$v.so $L
    echo -n "$v "; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | grep -o "encode_GiBs_per_gpu[^,]*,[^,]*"
  done
done
cp /tmp/keep.so $L
