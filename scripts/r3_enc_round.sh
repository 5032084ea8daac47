#!/bin/bash
# Round-3 encoder visit: gpu suite on the tree's library, then every ab/v_*.so on the bench batch (4096 and 512 blocks),
# the new builds also with other LDS-table / global-table splits, then the encoder phase probe.  Usage: scripts/r3_enc_round.sh [tag] [reps]
TAG=${1:-r3enc}
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
  for f in ab/v*.so; do
    cp $f $L
    for nb in 4096 512; do
      echo -n "$(basename $f .so) blocks=$nb " | tee -a $OUT/variants.txt
      run $nb | tee -a $OUT/variants.txt
    done
  done
done
for f in ab/v1*.so ab/v2*.so; do
  cp $f $L
  for pct in 30 40 56 64 100; do
    echo -n "$(basename $f .so) split=$pct blocks=4096 " | tee -a $OUT/variants.txt
    K4LZ4_SPLIT_PCT=$pct run 4096 | tee -a $OUT/variants.txt
  done
done
cp /tmp/keep.so $L
K4_ONLY_ENCODE=1 timeout 300 python scripts/phase_probe.py > $OUT/phase_probe.txt 2>&1
sed -n '/ENCODE/,/DECODE/p' $OUT/phase_probe.txt | head -24
