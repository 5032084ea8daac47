#!/bin/bash
# HC visit: configs[4] (L03, and L06 for the general path) with the tree's library, per-kernel times from a kernel trace.
TAG=${1:-r3hc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/tools/config5_hc.py 2>&1 | grep -v amdgpu.ids | tee $OUT/config5_L3.json
K4_LEVEL=6 K4_BLOCKS=1024 timeout 600 python tests/tools/config5_hc.py 2>&1 | grep -v amdgpu.ids | tee $OUT/config5_L6.json
( cd /tmp && K4_BLOCKS=4096 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "k4_hc|Name" $f | cut -d, -f1-5 | head
