#!/bin/bash
# How busy the units a wave shares with its neighbours are during the bench kernels: vector memory address unit (TA), LDS,
# the scalar and vector issue ports.  Separate rocprofv3 --pmc passes (no tracing besides the kernel trace).  Usage: scripts/pmc_units.sh tag
TAG=${1:-pmcu}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-host-path"
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run ta TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE
run sq SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py gpurun_out/$TAG /tmp/pmc_units_traffic.json > gpurun_out/$TAG/pmc_units.txt 2>&1
grep -A6 "^== " gpurun_out/$TAG/pmc_units.txt | grep -v "cost_kernel\|order_kernel" | cut -c1-600
