#!/bin/bash
# Per-kernel times (rocprofv3 kernel trace) of every build ab/v_*.so on the bench batch.  Usage: scripts/variants_prof.sh [tag] [blocks]
TAG=${1:-varp}
NB=${2:-4096}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
cd /tmp
for f in $GRAFT_REPO_ROOT/ab/v_*.so; do
  v=$(basename $f .so)
  cp $f $L
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --blocks $NB --no-cpu-baseline --no-verify --no-host-path > $OUT/$v.log 2>&1
  echo "== $v" | tee -a $OUT/kernels.txt
  grep -h "k4_encode\|k4_decode\|k4_cost" $(find $OUT/$v -name "*kernel_stats.csv") | awk -F'","' '{printf "%-60s calls %s avg_us %.1f\n", substr($1,2,58), $2, $4/1000}' | tee -a $OUT/kernels.txt
  rm -rf $OUT/$v
done
cp /tmp/keep.so $L
