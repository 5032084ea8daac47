#!/bin/bash
# round 6: every ab/v_*.so through configs[4] (tests/tools/config5_hc.py), twice, then the kernel split of each
TAG=${1:-r6hcv}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in 1 2; do for f in ab/v_*.so; do cp $f $L; echo -n "$(basename $f .so) " | tee -a $OUT/hc.txt; timeout 600 python tests/tools/config5_hc.py 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/hc.txt; done; done
for f in ab/v_*.so; do cp $f $L; n=$(basename $f .so)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$n -o hc -- python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py > /dev/null 2>&1)
  echo "== $n" | tee -a $OUT/kernels.txt; find $OUT/prof_$n -name "*kernel_stats.csv" | head -1 | xargs head -5 | cut -c1-100 | tee -a $OUT/kernels.txt
done
cp /tmp/keep.so $L
