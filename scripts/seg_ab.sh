#!/bin/bash
# A/B of the segment kernels' register budget (ab/v_seg5.so: waves_per_eu(5,6), no scratch; ab/v_seg6.so: (6,6), spills to scratch)
# on the ragged configurations, plus two environment switches on the bench batch.  Usage: scripts/seg_ab.sh tag
TAG=${1:-segab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in 1 2; do
  for v in seg5 seg6; do
    cp ab/v_$v.so $L
    echo "== $v rep $r" | tee -a $OUT/seg_ab.txt
    timeout 600 python tests/tools/config4_pickle.py 2>/dev/null | tail -1 | cut -c1-700 | tee -a $OUT/seg_ab.txt
    timeout 300 python tests/tools/gpu_big_messages.py 2>/dev/null | tail -1 | cut -c1-700 | tee -a $OUT/seg_ab.txt
  done
done
cp /tmp/keep.so $L
for h in 12 16; do
  echo -n "K4LZ4_HOP2_MAX=$h " | tee -a $OUT/env_ab.txt
  K4LZ4_HOP2_MAX=$h timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/env_ab.txt
done
