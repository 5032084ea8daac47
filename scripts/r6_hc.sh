#!/bin/bash
# round 6, configs[4]: L03 as shipped (sequence records, parse beside the candidate kernel), without the overlap, and with LZ4HC_encodeSequence inside the parse loop (K4LZ4_NO_HC_RECORDS), the HC parity tests, the kernel split
TAG=${1:-r6hc}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2; do
  for v in records inline; do
    unset K4LZ4_NO_HC_RECORDS K4LZ4_NO_HC_OVERLAP
    if [ $v = inline ]; then export K4LZ4_NO_HC_RECORDS=1 K4LZ4_NO_HC_OVERLAP=1; fi
    if [ $v = records ]; then export K4LZ4_NO_HC_OVERLAP=1; fi
    echo -n "$v " | tee -a $OUT/hc.txt; timeout 600 python tests/tools/config5_hc.py 2>&1 | tail -1 | cut -c1-330 | tee -a $OUT/hc.txt
  done
done
unset K4LZ4_NO_HC_RECORDS K4LZ4_NO_HC_OVERLAP
timeout 1500 python -m pytest tests -m gpu -x -q -k "hc or HC or level or config5 or configs_4" 2>&1 | tail -3 | tee $OUT/pytest.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o hc -- python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-120
