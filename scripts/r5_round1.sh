#!/bin/bash
# Round-5 first GPU visit for the two-kernel encoder (k4lz4_parse.hpp): parity, then K x waves sweep, then a kernel trace.
TAG=${1:-r50}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
echo "== parity (default build)" | tee $OUT/log.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_parity.py -m gpu -x -q 2>&1 | tail -15 | tee -a $OUT/log.txt
echo "== bench with verification (default build)" | tee -a $OUT/log.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-host-path > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json | tee -a $OUT/log.txt
bench() { timeout 300 python bench.py --steps 10 --warmup 2 --blocks $1 --no-cpu-baseline --no-verify --no-host-path 2>&1 | tail -1 | grep -o '"encode_GiBs_per_gpu[^,]*,[^,]*'; }
echo "== old kernels (K4LZ4_NO_PARSE)" | tee -a $OUT/log.txt
for nb in 4096 512; do echo -n "old blocks=$nb " | tee -a $OUT/log.txt; K4LZ4_NO_PARSE=1 bench $nb | tee -a $OUT/log.txt; done
for r in 1 2; do
for f in ab/v_k*.so; do
  cp $f $L
  for w in 16 12 9; do
    echo -n "$(basename $f .so) waves=$w blocks=4096 " | tee -a $OUT/log.txt; K4LZ4_PARSE_WAVES=$w bench 4096 | tee -a $OUT/log.txt
  done
  echo -n "$(basename $f .so) blocks=512 " | tee -a $OUT/log.txt; bench 512 | tee -a $OUT/log.txt
done
done
cp /tmp/keep.so $L
echo "== kernel trace (default build)" | tee -a $OUT/log.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-host-path > $GRAFT_REPO_ROOT/$OUT/trace_bench.log 2>&1 )
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-160 {} | head -12' | tee -a $OUT/log.txt
