#!/bin/bash
# round 6, HC: k4_hc_cand_kernel with fewer workgroups per CU (K4LZ4_HC_CAND_DYNLDS bytes of unused LDS each): is it the number of loads in flight?
TAG=${1:-r6hc5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for lds in 0 20000 40000 80000; do
  (cd /tmp && K4LZ4_HC_CAND_DYNLDS=$lds timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$lds -o hc -- python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py > /dev/null 2>&1)
  echo -n "dynlds=$lds " | tee -a $OUT/kernels.txt; find $OUT/prof_$lds -name "*kernel_stats.csv" | head -1 | xargs grep cand_kernel | cut -c1-100 | tee -a $OUT/kernels.txt
done
