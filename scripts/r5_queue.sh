#!/bin/bash
TAG=${1:-r57}
OUT=gpurun_out/$TAG
mkdir -p $OUT
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
cp ab/v_a_cur.so $L
bench() { timeout 300 python bench.py --steps 10 --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
for r in 1 2; do
echo -n "static waves=16 " | tee -a $OUT/q.txt; bench 4096 | tee -a $OUT/q.txt
for w in 9 10 11 12 13 14 15; do echo -n "queue waves=$w " | tee -a $OUT/q.txt; K4LZ4_PARSE_QUEUE=1 K4LZ4_PARSE_WAVES=$w bench 4096 | tee -a $OUT/q.txt; done
done
K4LZ4_PARSE_QUEUE=1 K4LZ4_PARSE_WAVES=12 timeout 600 python bench.py --steps 5 --warmup 2 --no-host-path --no-cpu-baseline 2>&1 | tail -1 | grep -o '"bit_exact[^,]*' | tee -a $OUT/q.txt
cp /tmp/keep.so $L
