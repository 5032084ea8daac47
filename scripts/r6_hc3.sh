#!/bin/bash
# round 6, HC: the level-3 parse with 1 / 2 / 4 waves per block (K4LZ4_HC_SEGS) through configs[4], kernel split of each, smaller batches, then every HC test
TAG=${1:-r6hc3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2; do for sg in 1 2 4; do echo -n "segs=$sg " | tee -a $OUT/hc.txt; K4LZ4_HC_SEGS=$sg timeout 600 python tests/tools/config5_hc.py 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/hc.txt; done; done
for nb in 2048 1024 512; do for sg in 1 0; do echo -n "blocks=$nb segs=$sg " | tee -a $OUT/hc.txt; K4_BLOCKS=$nb K4LZ4_HC_SEGS=$sg timeout 600 python tests/tools/config5_hc.py 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/hc.txt; done; done
for sg in 1 2 4; do
  (cd /tmp && K4LZ4_HC_SEGS=$sg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_s$sg -o hc -- python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py > /dev/null 2>&1)
  echo "== segs=$sg" | tee -a $OUT/kernels.txt; find $OUT/prof_s$sg -name "*kernel_stats.csv" | head -1 | xargs head -5 | cut -c1-100 | tee -a $OUT/kernels.txt
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py tests/test_gpu_configs_full.py -x -q -k "hc or HC or level or optimal" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python tests/tools/gpu_stress_all.py 2 7 hc 2>&1 | tail -3 | tee $OUT/stress.txt
K4LZ4_HC_SEGS=4 timeout 600 python tests/tools/gpu_stress_all.py 2 8 hc 2>&1 | tail -3 | tee -a $OUT/stress.txt
