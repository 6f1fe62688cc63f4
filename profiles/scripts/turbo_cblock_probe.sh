#!/bin/bash
# turbo dims (d_model 1280), 16 clips: the clip-block sequence (AX_WHISPER_BATCHED_LN=2) against the split-K sequence (default)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  echo -n "split-K (default): "; timeout -k 10 200 python3 profiles/scripts/ab_step.py 16 turbo fp16
  echo -n "clip-block (BATCHED_LN=2): "; AX_WHISPER_BATCHED_LN=2 timeout -k 10 200 python3 profiles/scripts/ab_step.py 16 turbo fp16
done
