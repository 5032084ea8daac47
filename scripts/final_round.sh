#!/bin/bash
# The round's closing visit: tests, smoke, bench, kernel trace, PMC traffic (for exactly these sources), the other
# BASELINE configurations, the one-batch multi-rank driver on one rank.  Usage: scripts/final_round.sh tag
TAG=${1:-final}
bash scripts/gpu_round.sh $TAG
bash scripts/pmc_round.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc_round.log 2>&1
OUT=gpurun_out/$TAG
timeout 300 python scripts/config3_decode.py > $OUT/config3_decode.json 2>$OUT/config3_decode.err
timeout 600 python bench.py --strong --messages 20000 > $OUT/strong.log 2>&1
timeout 600 python tests/tools/config4_pickle.py > $OUT/config4_pickle.json 2>$OUT/config4_pickle.err
for nb in 256 512 1024 2048 4096; do echo -n "blocks $nb "; timeout 200 python bench.py --steps 10 --warmup 2 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*\|"encode_GiBs_per_gpu[^,]*,[^,]*'| tr '\n' ' '; echo; done > $OUT/block_count_scaling.txt
tail -3 gpurun_out/${TAG}_pmc/pmc_summary.txt; tail -2 $OUT/strong.log | cut -c1-600; cat $OUT/block_count_scaling.txt; cat $OUT/config3_decode.json | cut -c1-500
