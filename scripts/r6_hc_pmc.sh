#!/bin/bash
# round 6, HC: SQ / L2 counter passes over configs[4] (tests/tools/config5_hc.py), per kernel and launch.  Usage: scripts/r6_hc_pmc.sh tag
TAG=${1:-r6hcpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tests/tools/config5_hc.py"
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run dramrd TCC_EA0_RDREQ_DRAM_32B_sum
run dramwr TCC_EA0_WRREQ_WRITE_DRAM_32B_sum
cd $GRAFT_REPO_ROOT
python - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k4_hc" not in k: continue
        acc[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print("  %-34s per launch %16.0f   launches %d   mean ns %10.0f" % (c, sum(x for x, _ in v) / len(v), len(v), sum(t for _, t in v) / len(v)))
PY
cat $OUT/summary.txt
