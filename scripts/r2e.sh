#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
cp ab/B.so k4os/compression/lz4_amd/libk4lz4.so
timeout 300 python scripts/pair_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pair_probe.txt
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee $OUT/bench_short.txt
