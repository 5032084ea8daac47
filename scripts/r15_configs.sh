#!/bin/bash
# The other BASELINE configurations and the multi-rank driver, one visit.  Usage: scripts/r15_configs.sh tag
TAG=${1:-r15cfg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/config3_decode.py > $OUT/config3_decode.json 2>$OUT/config3_decode.err
timeout 600 python bench.py --strong --messages 20000 > $OUT/strong.log 2>&1
K4LZ4_RANK_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --strong --gpus 2 --messages 4000 > $OUT/strong_2ranks.log 2>&1
timeout 600 python tests/tools/config4_pickle.py > $OUT/config4_pickle.json 2>$OUT/config4_pickle.err
tail -1 $OUT/strong.log | cut -c1-900; tail -1 $OUT/strong_2ranks.log | cut -c1-900; cat $OUT/config4_pickle.json | cut -c1-600; cat $OUT/config3_decode.json | cut -c1-600
