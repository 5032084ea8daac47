#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
python scripts/ubench/parse_only_input.py /tmp/parse_only.bin > $OUT/parse_only.txt
scripts/ubench/parse_only /tmp/parse_only.bin 1 >> $OUT/parse_only.txt 2>&1
echo "--- 342x (4104 blocks)" >> $OUT/parse_only.txt
scripts/ubench/parse_only /tmp/parse_only.bin 342 | grep launch >> $OUT/parse_only.txt 2>&1
cat $OUT/parse_only.txt
bash scripts/ab_round.sh $1 1
