#!/bin/bash
TAG=${1:-r6l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
bench() { timeout 600 python bench.py --steps 10 --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
for r in 1 2 3; do for f in ab/v_*.so; do cp $f $L; for nb in 4096 1024; do echo -n "$(basename $f .so) blocks=$nb " | tee -a $OUT/ab.txt; bench $nb | tee -a $OUT/ab.txt; done; done; done
cp /tmp/keep.so $L
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py -x -q -k "encod or bench_batch or fast or ragged or stress or borderline or span or residency or accel or pickle or Pickle" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 300 python tests/tools/gpu_big_messages.py 2>&1 | tail -1 | cut -c1-600 | tee $OUT/big_messages.txt
K4LZ4_NO_PARSE_BIG=1 timeout 300 python tests/tools/gpu_big_messages.py 2>&1 | tail -1 | cut -c1-600 | tee -a $OUT/big_messages.txt
