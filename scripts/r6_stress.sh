#!/bin/bash
# round 6: randomised GPU stress of what the round changed -- big blocks (byU32) and ragged pickles with segments through the two-step
# encoder, the persistent launch, HC level 3 from records, decode / mutate with the zeroed offset-0 bytes through host pointers
TAG=${1:-r6stress}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python tests/tools/gpu_stress_encode.py 3 21 400 big
  timeout 900 python tests/tools/gpu_stress_encode.py 2 22 300 device big
  timeout 900 python tests/tools/gpu_stress_encode.py 2 23 6000
  timeout 1200 python tests/tools/gpu_stress_all.py 3 31 bigpickle flags many envelopes
  timeout 1200 python tests/tools/gpu_stress_all.py 2 32 decode mutate pickle hc sizes
  timeout 1200 python tests/tools/gpu_stress_all.py 2 33 hclevels frames partial dict
  timeout 900 python tests/tools/gpu_large_blocks.py ) 2>&1 | grep -v amdgpu.ids | tee $OUT/stress.txt | tail -40
