#!/bin/bash
# round 6, HC: the candidate records out of LDS (default) against k4_hc_cand_kernel from memory (K4LZ4_HC_CAND_MEM=1): configs[4] twice each, kernel split, HC tests
TAG=${1:-r6hc6}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2; do for m in lds mem; do if [ $m = mem ]; then export K4LZ4_HC_CAND_MEM=1; else unset K4LZ4_HC_CAND_MEM; fi
  echo -n "$m " | tee -a $OUT/hc.txt; timeout 600 python tests/tools/config5_hc.py 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/hc.txt; done; done
for m in lds mem; do if [ $m = mem ]; then export K4LZ4_HC_CAND_MEM=1; else unset K4LZ4_HC_CAND_MEM; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$m -o hc -- python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py > /dev/null 2>&1)
  echo "== $m" | tee -a $OUT/kernels.txt; find $OUT/prof_$m -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-100 | tee -a $OUT/kernels.txt
done
unset K4LZ4_HC_CAND_MEM
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py tests/test_gpu_configs_full.py -x -q -k "hc or HC or level or optimal" 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 600 python tests/tools/gpu_stress_all.py 3 21 hc 2>&1 | tail -1 | tee $OUT/stress.txt
