#!/bin/bash
# every ab/v*.so on the bench batch (4096 / 512 blocks), REPS times; the last one also at other LDS-table shares
TAG=${1:-r3ab}; REPS=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so; cp $L /tmp/keep.so
run() { timeout 300 python bench.py --steps 10 --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
for r in $(seq $REPS); do for f in ab/v*.so; do cp $f $L; for nb in 4096 512; do echo -n "$(basename $f .so) blocks=$nb " | tee -a $OUT/variants.txt; run $nb | tee -a $OUT/variants.txt; done; done; done
f=$(ls ab/v*.so | tail -1); cp $f $L
for pct in 44 50 52 56; do echo -n "$(basename $f .so) split=$pct " | tee -a $OUT/variants.txt; K4LZ4_SPLIT_PCT=$pct run 4096 | tee -a $OUT/variants.txt; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
cp /tmp/keep.so $L
