#!/bin/bash
# Key splits of the folded cross-attention launch (AX_WHISPER_CROSS_SPLIT_FOLD) by clip count, decoder step of Whisper-small at t = 224
for B in 4 6 8 12 16 20; do
  for sp in 0 1 2 3 4 6; do
    echo -n "B=$B split=$sp (0 = the rule): "; AX_WHISPER_CROSS_SPLIT_FOLD=$sp timeout -k 10 120 python3 profiles/scripts/ab_step.py $B | sed 's/.*decode_step/decode_step/'
  done
done
