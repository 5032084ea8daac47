#!/bin/bash
# A/B of the builds in ab/ (encode GiB/s at 4096 and 512 blocks, REPS times) + phase probe of the builds in ab_prof/
TAG=${1:-r53}
REPS=${2:-2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
bench() { timeout 300 python bench.py --steps 10 --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
if [ -n "$WITH_OLD" ]; then for nb in 4096 512; do echo -n "old blocks=$nb " | tee -a $OUT/ab.txt; K4LZ4_NO_PARSE=1 bench $nb | tee -a $OUT/ab.txt; done; fi
for r in $(seq $REPS); do
for f in ab/v_*.so; do
  cp $f $L
  for w in ${WAVES:-16}; do
    for nb in ${SIZES:-4096 512}; do echo -n "$(basename $f .so) waves=$w blocks=$nb " | tee -a $OUT/ab.txt; K4LZ4_PARSE_WAVES=$w bench $nb | tee -a $OUT/ab.txt; done
  done
done
done
if [ -n "$VERIFY" ]; then cp ab/$VERIFY.so $L; timeout 600 python bench.py --steps 5 --warmup 2 --no-host-path --no-cpu-baseline 2>&1 | tail -1 | grep -o '"bit_exact[^,]*' | tee -a $OUT/ab.txt; fi
cp /tmp/keep.so $L
if ls ab_prof/v_*.so >/dev/null 2>&1; then SIZES="${PSIZES:-512 4096}" bash scripts/r5_probe.sh $TAG; fi
