#!/bin/bash
# round 6: the second-generation pair decoder (k4lz4_decode2.hpp) against the first on one box: parity tests, then the bench batch
# at 4096 / 2048 / 1024 blocks with and without K4LZ4_NO_XDEC, REPS times.  Usage: scripts/r6_dec.sh tag [reps]
TAG=${1:-r6dec}; REPS=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py -x -q -k "decod or mutat or long_length or offset or bench_batch or stream or roundtrip or config" 2>&1 | tail -5 | tee $OUT/pytest.txt
for r in $(seq $REPS); do
  for v in new old; do
    for nb in 4096 2048 1024; do
      echo -n "$v blocks=$nb " | tee -a $OUT/bench.txt
      if [ $v = old ]; then export K4LZ4_NO_XDEC=1; else unset K4LZ4_NO_XDEC; fi
      timeout 300 python bench.py --steps 20 --warmup 3 --blocks $nb --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*' | tee -a $OUT/bench.txt
    done
  done
done
unset K4LZ4_NO_XDEC
