#!/bin/bash
# One GPU-box visit comparing two builds of libk4lz4.so (ab/A.so = before, ab/B.so = after) on the same box:
# gpu tests with B, bench (decode/encode rates) alternating A/B, configs[2] decode-only, phase probe with B.
# Usage: scripts/ab_round.sh [tag] [reps]
TAG=${1:-ab}
REPS=${2:-2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp ab/B.so $L
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
for r in $(seq $REPS); do
  for v in A B; do
    cp ab/$v.so $L
    echo -n "$v " | tee -a $OUT/ab_bench.txt
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/ab_bench.txt
  done
done
for v in A B; do
  cp ab/$v.so $L
  echo "== $v" | tee -a $OUT/ab_config3.txt
  K4_BLOCKS=${K4_BLOCKS:-262144} timeout 300 python scripts/config3_decode.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_config3.txt
done
cp ab/B.so $L
timeout 300 python scripts/phase_probe.py > $OUT/phase_probe.txt 2>&1
sed -n '/DECODE/,$p' $OUT/phase_probe.txt
