#!/bin/bash
# one visit, three builds (ab/v_a_old.so, v_b_head.so, v_c_dec32.so): the ragged configurations for a vs b, the bench batch for b vs c
TAG=${1:-ab3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in 1 2 3; do
  for v in a_old b_head; do
    cp ab/v_$v.so $L
    echo -n "$v " | tee -a $OUT/ragged.txt
    timeout 600 python tests/tools/config4_pickle.py 2>/dev/null | tail -1 | grep -o '"pickle_ms[^,]*,[^,]*' | tee -a $OUT/ragged.txt
  done
done
for r in 1 2 3; do
  for v in b_head c_dec32; do
    cp ab/v_$v.so $L
    for nb in 4096 1024; do
      echo -n "$v blocks=$nb " | tee -a $OUT/bench.txt
      timeout 300 python bench.py --steps 20 --warmup 3 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/bench.txt
    done
  done
done
cp ab/v_c_dec32.so $L
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 | tee -a $OUT/pytest_dec32.txt
K4_BLOCKS=262144 timeout 300 python scripts/config3_decode.py 2>/dev/null | cut -c1-300 | tee -a $OUT/config3_dec32.txt
cp ab/v_b_head.so $L
K4_BLOCKS=262144 timeout 300 python scripts/config3_decode.py 2>/dev/null | cut -c1-300 | tee -a $OUT/config3_head.txt
cp /tmp/keep.so $L
