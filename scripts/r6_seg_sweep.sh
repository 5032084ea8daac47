#!/bin/bash
# round 6: the segment plan's sizes once more, for the two-step encoder's runs (rank 0's share of configs[3], the 13 big messages, one 4 MiB message)
TAG=${1:-r6sweep}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  echo -n "$name share " | tee -a $OUT/sweep.txt; env "$@" timeout 900 python tests/tools/config4_pickle.py 2>&1 | tail -1 | grep -o '"pickle_ms[^,]*' | tr '\n' ' ' | tee -a $OUT/sweep.txt
  echo -n " big " | tee -a $OUT/sweep.txt; env "$@" timeout 300 python tests/tools/gpu_big_messages.py 2>&1 | tail -1 | grep -o '"batch_pickle_ms[^,]*\|"one_4MiB[^,]*\|differing[^]]*]' | tr '\n' ' ' | tee -a $OUT/sweep.txt; echo | tee -a $OUT/sweep.txt
}
if [ -n "$SWEEP2" ]; then
for r in 1 2; do
run default K4_X=1
run tmax1280 K4LZ4_SEG_TARGET_MAX=1310720
run tmax1408 K4LZ4_SEG_TARGET_MAX=1441792
run tmax1536 K4LZ4_SEG_TARGET_MAX=1572864
run tmax1792 K4LZ4_SEG_TARGET_MAX=1835008
run tmax1408div3000 K4LZ4_SEG_TARGET_MAX=1441792 K4LZ4_SEG_DIV=3000
run tmax1408div4000 K4LZ4_SEG_TARGET_MAX=1441792 K4LZ4_SEG_DIV=4000
done
exit 0
fi
run default K4_X=1
run div5000 K4LZ4_SEG_DIV=5000
run div7000 K4LZ4_SEG_DIV=7000
run div2500 K4LZ4_SEG_DIV=2500
run tmax896 K4LZ4_SEG_TARGET_MAX=917504
run tmax1408 K4LZ4_SEG_TARGET_MAX=1441792
run warm256 K4LZ4_SEG_WARM=262144
run warm320 K4LZ4_SEG_WARM=327680
run t512 K4LZ4_SEG_TARGET=524288 K4LZ4_SEG_TARGET_MAX=917504
run min768 K4LZ4_SEG_MIN=786432
run default2 K4_X=1
