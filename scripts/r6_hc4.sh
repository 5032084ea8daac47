#!/bin/bash
# round 6, HC: ab/v_*.so x K4LZ4_HC_SEGS through configs[4] (twice), kernel split of each variant at SEGS (default 2)
TAG=${1:-r6hc4}; SEGS=${2:-"2 4"}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in 1 2; do for f in ab/v_*.so; do cp $f $L; for sg in $SEGS; do echo -n "$(basename $f .so) segs=$sg " | tee -a $OUT/hc.txt; K4LZ4_HC_SEGS=$sg timeout 600 python tests/tools/config5_hc.py 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/hc.txt; done; done; done
for f in ab/v_*.so; do cp $f $L; n=$(basename $f .so); for sg in $SEGS; do
  (cd /tmp && K4LZ4_HC_SEGS=$sg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_${n}_s$sg -o hc -- python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py > /dev/null 2>&1)
  echo "== $n segs=$sg" | tee -a $OUT/kernels.txt; find $OUT/prof_${n}_s$sg -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-100 | tee -a $OUT/kernels.txt
done; done
cp /tmp/keep.so $L
