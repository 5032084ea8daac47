#!/bin/bash
# Round-3 baseline visit: gpu suite, smoke, bench, kernel trace, phase probes of both kernels, block-count scaling.
TAG=${1:-r15a}
bash scripts/gpu_round.sh $TAG
OUT=gpurun_out/$TAG
timeout 300 python scripts/phase_probe.py > $OUT/phase_probe.txt 2>&1
timeout 300 python scripts/pair_probe.py > $OUT/pair_probe.txt 2>&1
for nb in 256 512 1024 2048 4096; do echo -n "blocks $nb "; timeout 200 python bench.py --steps 10 --warmup 2 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"ms_per_step[^,]*\|"encode_GiBs_per_gpu[^,]*,[^,]*'| tr '\n' ' '; echo; done > $OUT/block_count_scaling.txt
cat $OUT/block_count_scaling.txt
