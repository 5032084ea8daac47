#!/bin/bash
# phase probe (encoder part) with every library given.  Usage: scripts/r3_probe_ab.sh tag p0 p1 ...
TAG=$1; shift
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  cp ab/$v.so $L
  echo "#### $v"
  timeout 300 python scripts/phase_probe.py > gpurun_out/$TAG/probe_$v.txt 2>&1
  sed -n '/ENCODE/,/DECODE/p' gpurun_out/$TAG/probe_$v.txt | grep -v "rounds .* not following" | head -20
done
cp /tmp/keep.so $L
