#!/bin/bash
# Clip-block query fold on / off (AX_WHISPER_CBLOCK_QFOLD), decoder step of Whisper-small at several clip counts, A/B/A/B in one call;
# then the folded step's cross-attention key splits at 64 clips (AX_WHISPER_CROSS_SPLIT_FOLD)
set -e
for B in ${@:-4 8 16 64}; do
  for rep in 1 2; do
    for f in 1 0; do
      echo -n "QFOLD=$f: "; AX_WHISPER_CBLOCK_QFOLD=$f timeout -k 10 200 python3 profiles/scripts/ab_step.py $B
    done
  done
done
for sp in 1 2 3; do
  echo -n "64 clips, folded, CROSS_SPLIT_FOLD=$sp: "; AX_WHISPER_CROSS_SPLIT_FOLD=$sp timeout -k 10 200 python3 profiles/scripts/ab_step.py 64
done
for sp in 1 2 3; do
  echo -n "32 clips, folded, CROSS_SPLIT_FOLD=$sp: "; AX_WHISPER_CROSS_SPLIT_FOLD=$sp timeout -k 10 200 python3 profiles/scripts/ab_step.py 32
done
