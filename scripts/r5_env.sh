#!/bin/bash
# one build (ab/$2.so), several environments: "name:VAR=val,VAR=val" ...
TAG=$1; LIBV=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
cp ab/$LIBV.so $L
bench() { timeout 300 python bench.py --steps 10 --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
for r in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  for nb in 4096 512; do
    echo -n "$name blocks=$nb " | tee -a $OUT/env.txt
    ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; bench $nb ) | tee -a $OUT/env.txt
  done
done
done
if [ -n "$VERIFYENV" ]; then ( export $VERIFYENV; timeout 600 python bench.py --steps 5 --warmup 2 --no-host-path --no-cpu-baseline 2>&1 | tail -1 | grep -o '"bit_exact[^,]*' | tee -a $OUT/env.txt ); fi
cp /tmp/keep.so $L
