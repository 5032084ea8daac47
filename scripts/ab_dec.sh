#!/bin/bash
# decode A/B on one box: every ab/v_*.so, bench batch at 4096 / 1024 blocks, REPS times; then the GPU parity tests + configs[2] with the last one
TAG=${1:-abdec}; REPS=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in $(seq $REPS); do
  for f in ab/v_*.so; do
    cp $f $L
    for nb in 4096 1024; do
      echo -n "$(basename $f .so) blocks=$nb " | tee -a $OUT/bench.txt
      timeout 300 python bench.py --steps 20 --warmup 3 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/bench.txt
    done
  done
done
last=$(ls ab/v_*.so | tail -1); cp $last $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py tests/test_gpu_frames.py -x -q 2>&1 | tail -2 | tee -a $OUT/pytest_last.txt
K4_BLOCKS=262144 timeout 300 python scripts/config3_decode.py 2>/dev/null | cut -c1-300 | tee -a $OUT/config3_last.txt
timeout 300 python tests/tools/config4_pickle.py 2>/dev/null | tail -1 | cut -c1-400 | tee -a $OUT/config4_last.txt
cp /tmp/keep.so $L
