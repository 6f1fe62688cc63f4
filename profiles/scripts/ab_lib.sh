#!/bin/bash
# A/B of several builds of libax_whisper.so in one call on one device: every profiles/microbench/tmp/lib*.so
# (baseline / variant builds, not tracked) against the in-tree build, two rounds. usage: ab_lib.sh <python script and args>
set -e
L=whisper.axera_amd/lib/libax_whisper.so
cp $L /tmp/lib_intree.so
for r in 1 2; do
  for v in profiles/microbench/tmp/lib*.so /tmp/lib_intree.so; do
    cp $v $L; echo "== $(basename $v)"; timeout -k 10 300 python "$@"
  done
done
cp /tmp/lib_intree.so $L
