#!/bin/bash
# rank 0's share of configs[3] and the 13 big messages against the segments' warm-up (K4LZ4_SEG_WARM): since round 4 a boundary that
# does not verify costs one piece's range again, not the rest of the message
TAG=${1:-warm}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for w in 393216 262144 196608 131072; do
  echo -n "K4LZ4_SEG_WARM=$w " | tee -a $OUT/warm.txt
  K4LZ4_SEG_WARM=$w timeout 600 python tests/tools/config4_pickle.py 2>/dev/null | tail -1 | grep -o '"pickle_ms[^,]*' | tr '\n' ' ' | tee -a $OUT/warm.txt
  K4LZ4_SEG_WARM=$w timeout 300 python tests/tools/gpu_big_messages.py 2>/dev/null | tail -1 | grep -o '"batch_pickle_ms[^,]*\|"one_4MiB[^,]*\|"envelopes_differing[^]]*]' | tr '\n' ' ' | tee -a $OUT/warm.txt; echo | tee -a $OUT/warm.txt
done
