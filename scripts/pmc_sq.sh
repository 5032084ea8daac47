#!/bin/bash
# SQ counter passes only (instruction counts, wave cycles, wait states) over a short bench run.  Usage: scripts/pmc_sq.sh tag
TAG=${1:-pmcsq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-host-path"   # (the host-pointer leg launches halves: they would dilute the per-launch averages)
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
cd $GRAFT_REPO_ROOT && python scripts/pmc_summary.py gpurun_out/$TAG | grep -A1 "decode\|==" | grep -v "^--"
