#!/bin/bash
# per-kernel durations of the encoder at batch 64 (rocprofv3 --kernel-trace --stats over profiles/enc_driver.py)
OUT=$PWD/gpurun_out/enc_stats
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/profiles/enc_driver.py ${1:-64} > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/raw -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv && rm -rf $OUT/raw
cut -c1-200 $OUT/kernel_stats.csv | head -16
