#!/bin/bash
# SQ instruction counters of the encoder kernels for every build ab/v_*.so (one rocprofv3 pass each).  Usage: scripts/pmc_variants.sh tag
TAG=${1:-pmcvar}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
cd /tmp
for f in $GRAFT_REPO_ROOT/ab/v_*.so; do
  name=$(basename $f .so)
  cp $f $L
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-host-path > $OUT/$name.log 2>&1
  echo "== $name"
  python3 - "$OUT/$name" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'encode_fast' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
for k in acc:
    print('  ',k.split('(')[0][-28:], {c:int(v/max(n[k],1)) for c,v in sorted(acc[k].items())})
PY
done
cp /tmp/keep.so $L
