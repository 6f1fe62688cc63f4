#!/bin/bash
# turbo dims (d_model 1280), 16 clips: the clip-block sequence (AX_WHISPER_BATCHED_LN=2) against the split-K sequence (default), and
# the per-kernel durations of the clip-block step under rocprofv3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  echo -n "split-K (default): "; timeout -k 10 200 python3 profiles/scripts/ab_step.py 16 turbo fp16
  echo -n "clip-block (BATCHED_LN=2): "; AX_WHISPER_BATCHED_LN=2 timeout -k 10 200 python3 profiles/scripts/ab_step.py 16 turbo fp16
done
OUT=$PWD/gpurun_out/turbo_cb
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export AX_WHISPER_BATCHED_LN=2 && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $R/profiles/scripts/ab_step.py 16 turbo fp16 > $OUT/run.log 2>&1)
f=$(find $OUT/raw -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats_cblock.csv; rm -rf $OUT/raw
cut -c1-170 $OUT/kernel_stats_cblock.csv | head -12
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $R/profiles/scripts/ab_step.py 16 turbo fp16 > $OUT/run2.log 2>&1)
f=$(find $OUT/raw -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats_splitk.csv; rm -rf $OUT/raw
cut -c1-170 $OUT/kernel_stats_splitk.csv | head -12
