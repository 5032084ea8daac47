#!/bin/bash
# stamp_probe.py for every build ab/v_*.so.  Usage: scripts/variants_stamp.sh [tag]
TAG=${1:-vars}
OUT=gpurun_out/$TAG
mkdir -p $OUT
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for f in ab/v_*.so; do
  v=$(basename $f .so)
  cp $f $L
  timeout 300 python scripts/stamp_probe.py 2>&1 | grep -v amdgpu > $OUT/stamp_$v.txt
done
cp /tmp/keep.so $L
