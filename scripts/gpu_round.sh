#!/bin/bash
# One GPU-box visit: gpu tests, smoke, bench, rocprofv3 kernel-trace stats.  Everything lands in
# gpurun_out/ (merged back by gpurun).  Usage: scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt
nproc > $OUT/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1; echo "bench exit $?" >> $OUT/bench.log
python scripts/host_path_rate.py > $OUT/host_path.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-host-path > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1 )
find $OUT/prof -name "*stats*" | head > $OUT/prof_files.txt
tail -5 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -3; tail -2 $OUT/bench.log
