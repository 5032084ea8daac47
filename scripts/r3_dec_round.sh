#!/bin/bash
# Round-3 decoder visit: gpu suite on the tree's library, every ab/[vd]*.so on the bench batch (4096 / 512 blocks) and on
# configs[2] (decode only, 4 KiB blocks), then the pair probe with ab/p*_nodrain.so.  Usage: scripts/r3_dec_round.sh [tag] [reps]
TAG=${1:-r3dec}
REPS=${2:-2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
cp $L /tmp/keep.so
run() { timeout 300 python bench.py --steps 10 --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
for r in $(seq $REPS); do
  for f in ab/v*.so ab/d*.so; do
    cp $f $L
    for nb in 4096 512; do
      echo -n "$(basename $f .so) blocks=$nb " | tee -a $OUT/variants.txt
      run $nb | tee -a $OUT/variants.txt
    done
  done
done
for f in ab/v*.so ab/d*.so; do
  cp $f $L
  echo "== $(basename $f .so)" | tee -a $OUT/config3.txt
  K4_BLOCKS=${K4_BLOCKS:-262144} timeout 300 python scripts/config3_decode.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $OUT/config3.txt
done
for f in ab/p*_nodrain.so; do
  cp $f $L
  echo "== $(basename $f .so)"
  timeout 300 python scripts/pair_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pair_probe_$(basename $f .so).txt
done
cp /tmp/keep.so $L
