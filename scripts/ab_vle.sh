#!/bin/bash
# A/B of the wave-wide length-field reader (ab/v_a_base.so vs ab/v_b_vle.so): bench batch, configs[2] (text and random), configs[3] share
TAG=${1:-abvle}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for r in 1 2; do
  for f in ab/v_a_base.so ab/v_b_vle.so; do
    cp $f $L
    echo -n "$(basename $f .so) " | tee -a $OUT/ab.txt
    timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"decode_GiBs_per_gpu[^,]*' | tee -a $OUT/ab.txt
    K4_BLOCKS=1048576 timeout 300 python scripts/config3_decode.py 2>/dev/null | grep -o '"variant[^,]*\|"decode_GiBs[^,]*\|"frac_of_8TBs[^,]*' | tr '\n' ' ' | tee -a $OUT/ab.txt; echo | tee -a $OUT/ab.txt
  done
done
for f in ab/v_a_base.so ab/v_b_vle.so; do
  cp $f $L
  echo -n "$(basename $f .so) " | tee -a $OUT/ab.txt
  timeout 300 python tests/tools/config4_pickle.py 2>/dev/null | tail -1 | grep -o '"unpickle_ms[^,]*' | tee -a $OUT/ab.txt
done
cp ab/v_b_vle.so $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py tests/test_gpu_frames.py -x -q 2>&1 | tail -2 | tee -a $OUT/pytest_vle.txt
cp /tmp/keep.so $L
