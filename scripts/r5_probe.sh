#!/bin/bash
# parse-kernel phase probe for the profiling builds in ab_prof/ (full batch and a 512-block batch)
TAG=${1:-r51}
OUT=gpurun_out/$TAG
mkdir -p $OUT
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for f in ab_prof/v_*.so; do
  cp $f $L
  for nb in ${SIZES:-4096 512}; do
    echo "== $(basename $f .so) blocks=$nb waves=${K4LZ4_PARSE_WAVES:-16}" | tee -a $OUT/probe.txt
    K4_BLOCKS=$nb timeout 300 python scripts/parse_probe.py 2>&1 | tail -40 | tee -a $OUT/probe.txt
  done
done
cp /tmp/keep.so $L
