#!/bin/bash
# round 6, configs[3]: rank 0's share, the 13 big messages and one 4 MiB message through the two-step encoder with segments (default)
# and through the one-kernel encoders (K4LZ4_NO_PARSE_SEG); then the pickle / segment parity tests
TAG=${1:-r6cfg4}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2; do
  for v in two_step one_kernel; do
    if [ $v = one_kernel ]; then export K4LZ4_NO_PARSE_SEG=1; else unset K4LZ4_NO_PARSE_SEG; fi
    echo -n "$v share " | tee -a $OUT/cfg4.txt; timeout 900 python tests/tools/config4_pickle.py 2>&1 | tail -1 | cut -c1-400 | tee -a $OUT/cfg4.txt
    echo -n "$v big " | tee -a $OUT/cfg4.txt; timeout 300 python tests/tools/gpu_big_messages.py 2>&1 | tail -1 | cut -c1-400 | tee -a $OUT/cfg4.txt
  done
done
unset K4LZ4_NO_PARSE_SEG
timeout 1800 python -m pytest tests -m gpu -x -q -k "pickle or Pickle or segment or config4 or configs_3 or big_messages or strong or ragged" 2>&1 | tail -4 | tee $OUT/pytest.txt
