#!/bin/bash
# Multi-branch decoder steps with the branches on disjoint CU sets (AX_WHISPER_CU_SPLIT=1, launch_step) against the one multi-branch
# graph (=0): Whisper-small, step at t = 224, A/B/A/B per clip count
for B in ${@:-64 32 24 48}; do
  for rep in 1 2; do
    for f in 1 0; do
      echo -n "B=$B CU_SPLIT=$f: "; AX_WHISPER_CU_SPLIT=$f timeout -k 10 200 python3 profiles/scripts/ab_step.py $B | sed 's/.*decode_step/decode_step/'
    done
  done
done
