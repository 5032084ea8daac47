#!/bin/bash
# The round's closing visit: tests, smoke, bench, kernel trace, per-block stamps, PMC traffic (for exactly these sources), the other
# BASELINE configurations, big messages with and without segments, the one-batch multi-rank driver.  Usage: scripts/final_round.sh tag
TAG=${1:-final}
bash scripts/gpu_round.sh $TAG
OUT=gpurun_out/$TAG
python scripts/stamp_probe.py 2>&1 | grep -v amdgpu > $OUT/stamp.txt
bash scripts/r15_configs.sh $TAG > $OUT/configs.log 2>&1
timeout 300 python tests/tools/gpu_big_messages.py 2>&1 | tail -1 > $OUT/big_messages.json
K4LZ4_NO_SEGMENTS=1 timeout 300 python tests/tools/gpu_big_messages.py 2>&1 | tail -1 >> $OUT/big_messages.json
timeout 600 python tests/tools/config5_hc.py 2>&1 | tail -1 > $OUT/config5_hc.json
for nb in 256 512 1024 2048 4096 8192 16384; do echo -n "blocks $nb "; timeout 200 python bench.py --steps 10 --warmup 2 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*\|"encode_GiBs_per_gpu[^,]*,[^,]*'| tr '\n' ' '; echo; done > $OUT/block_count_scaling.txt
for nb in 65536 262144; do echo -n "4 KiB blocks $nb "; timeout 300 python bench.py --steps 3 --warmup 1 --blocks $nb --block-size 4096 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*\|"encode_GiBs_per_gpu[^,]*,[^,]*'| tr '\n' ' '; echo; done >> $OUT/block_count_scaling.txt
bash scripts/pmc_round.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc_round.log 2>&1
tail -3 gpurun_out/${TAG}_pmc/pmc_summary.txt; cat $OUT/big_messages.json; cat $OUT/block_count_scaling.txt; tail -4 $OUT/configs.log | cut -c1-400
