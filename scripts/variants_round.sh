#!/bin/bash
# One GPU-box visit timing every build ab/v_*.so on the bench batch (and on a 512-block batch, where a block's own
# latency shows): encode / decode GiB/s per build, alternating, REPS times.  Usage: scripts/variants_round.sh [tag] [reps] [sizes]
TAG=${1:-var}
REPS=${2:-2}
SIZES=${3:-"4096 512"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in $(seq $REPS); do
  for f in ab/v_*.so; do
    cp $f $L
    for nb in $SIZES; do
      echo -n "$(basename $f .so) blocks=$nb " | tee -a $OUT/variants.txt
      timeout 300 python bench.py --steps 10 --warmup 2 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/variants.txt
    done
  done
done
cp /tmp/keep.so $L
if [ -n "$SPLITS" ]; then
  for s in $SPLITS; do
    echo -n "split=$s " | tee -a $OUT/variants.txt
    K4LZ4_SPLIT_PCT=$s timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/variants.txt
  done
fi
