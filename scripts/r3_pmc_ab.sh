#!/bin/bash
# SQ counter passes for every ab/v*.so given (instruction counts and wait states per kernel).  Usage: scripts/r3_pmc_ab.sh tag v0_old v1_new ...
TAG=$1; shift
L=k4os/compression/lz4_amd/libk4lz4.so
cp $L /tmp/keep.so
for v in "$@"; do
  cp ab/$v.so $L
  echo "#### $v"
  bash scripts/pmc_sq.sh ${TAG}_$v 2>&1 | grep -A1 "encode_fast\|==" | grep -v "^--" | cut -c1-400
done
cp /tmp/keep.so $L
