#!/bin/bash
# rocprofv3 PMC passes over the bench workload (separate passes: SQ 8 slots, TCC FETCH/WRITE apart), plus the same two TCC
# passes over scripts/ubench/pmc_calib (kernels of known byte counts) so that FETCH_SIZE / WRITE_SIZE can be turned into
# bytes for THIS access pattern.  Writes profiles-ready files under gpurun_out/<tag>/ and pmc_traffic.json with the hash of
# the kernel sources it was measured on (bench.py refuses traffic figures from other sources).
# Usage: scripts/pmc_round.sh [tag]
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-host-path"   # (the host-pointer leg launches halves: they would dilute the per-launch averages)
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
cal() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o $name -- $GRAFT_REPO_ROOT/scripts/ubench/pmc_calib > $OUT/$name.log 2>&1; }
run sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
# the L2's requests to DRAM counted in 32-byte units: on the calibration kernels these two reproduce the known byte counts
# exactly (1 GiB streamed = 33 554 768 x 32 B read, 33 554 432 x 32 B written; a scattered 16-byte read costs a 128-byte
# line, a scattered 2-byte write a 32-byte sector), which FETCH_SIZE / WRITE_SIZE do not
# the L2's own hit rate (round 6: the review asked for it beside the traffic): hits / (hits + misses), over all channels
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run dramrd TCC_EA0_RDREQ_DRAM_32B_sum
run dramwr TCC_EA0_WRREQ_WRITE_DRAM_32B_sum
cal caldramrd TCC_EA0_RDREQ_DRAM_32B_sum
cal caldramwr TCC_EA0_WRREQ_WRITE_DRAM_32B_sum
cal calfetch FETCH_SIZE
cal calwrite WRITE_SIZE
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py gpurun_out/$TAG gpurun_out/$TAG/pmc_traffic.json > gpurun_out/$TAG/pmc_summary.txt 2>&1
tail -30 gpurun_out/$TAG/pmc_summary.txt
