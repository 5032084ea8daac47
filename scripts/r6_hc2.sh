#!/bin/bash
# round 6, HC: ab/v_*.so through configs[4] (twice, alternating) with the kernel split of each, then the tree's library through every HC test
TAG=${1:-r6hc2}
bash scripts/r6_hc_var.sh $TAG
OUT=gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py tests/test_gpu_configs_full.py -x -q -k "hc or HC or level or optimal" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python tests/tools/gpu_stress_all.py 2 7 hc 2>&1 | tail -3 | tee $OUT/stress.txt
