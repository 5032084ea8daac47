#!/bin/bash
# round 6, encoder: ab/v_*.so at 4096 blocks (REPS times, alternating), then with the tree's library batches beyond one residency
# through the persistent launch and through launches of one residency (K4LZ4_NO_PERSIST), then the encoder's parity tests
TAG=${1:-r6enc}; REPS=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
bench() { timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
if ls ab/v_*.so >/dev/null 2>&1; then
for r in $(seq $REPS); do
  for f in ab/v_*.so; do cp $f $L; echo -n "$(basename $f .so) blocks=4096 " | tee -a $OUT/ab.txt; bench 4096 | tee -a $OUT/ab.txt; done
done
fi
cp /tmp/keep.so $L
for nb in 6144 8192 16384; do
  for v in persist launches; do
    if [ $v = launches ]; then export K4LZ4_NO_PERSIST=1; else unset K4LZ4_NO_PERSIST; fi
    echo -n "$v blocks=$nb " | tee -a $OUT/big.txt; STEPS=5 bench $nb | tee -a $OUT/big.txt
  done
done
unset K4LZ4_NO_PERSIST
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py -x -q -k "encod or bench_batch or fast or ragged or stress or borderline or span" 2>&1 | tail -4 | tee $OUT/pytest.txt
