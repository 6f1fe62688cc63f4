#!/bin/bash
# A/B of two builds of libax_whisper.so in one call on one device: profiles/microbench/tmp/libA.so (baseline build,
# not tracked) against the in-tree build. usage: ab_lib.sh <python script and args>
set -e
L=whisper.axera_amd/lib/libax_whisper.so
cp $L /tmp/libB.so
for r in 1 2; do
  cp profiles/microbench/tmp/libA.so $L; echo "== A (baseline)"; timeout -k 10 300 python "$@"
  cp /tmp/libB.so $L; echo "== B (in-tree)"; timeout -k 10 300 python "$@"
done
