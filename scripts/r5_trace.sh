#!/bin/bash
# kernel trace of the bench with the build ab/$2.so (default: the tree's)
TAG=${1:-r60}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
[ -n "$2" ] && cp ab/$2.so $L
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-host-path > $GRAFT_REPO_ROOT/$OUT/trace_bench.log 2>&1 )
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-140 {} | head -9' | tee -a $OUT/trace.txt
cp /tmp/keep.so $L
