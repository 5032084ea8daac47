#!/bin/bash
# round 6: batches beyond one residency through the persistent launch and through launches of one residency (K4LZ4_NO_PERSIST); small blocks
TAG=${1:-r6big}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bench() { timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --blocks $1 --block-size ${BS:-65536} --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
for r in 1 2; do
for nb in 4096 8192 12288 16384; do
  for v in persist launches; do
    if [ $v = launches ]; then export K4LZ4_NO_PERSIST=1; else unset K4LZ4_NO_PERSIST; fi
    echo -n "$v blocks=$nb " | tee -a $OUT/big.txt; bench $nb | tee -a $OUT/big.txt
  done
done
done
for nb in 65536 262144; do
  for v in persist launches; do
    if [ $v = launches ]; then export K4LZ4_NO_PERSIST=1; else unset K4LZ4_NO_PERSIST; fi
    echo -n "$v 4KiB blocks=$nb " | tee -a $OUT/small.txt; BS=4096 STEPS=3 bench $nb | tee -a $OUT/small.txt
  done
done
unset K4LZ4_NO_PERSIST
