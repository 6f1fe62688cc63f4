#!/bin/bash
# Collects the round's evidence on the MI355X box: bench lines, rocprofv3 kernel stats, PMC traffic (see profiles/README.md).
set -o pipefail
OUT=$PWD/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_b1.json 2> $OUT/bench_b1.err || exit 1
python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b64.json 2> $OUT/bench_b64.err || exit 1
python bench.py --model turbo --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_turbo_b1.json 2> $OUT/bench_turbo_b1.err || exit 1
AX_WHISPER_DECODE=graph python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_b1_graph.json 2> $OUT/bench_b1_graph.err || exit 1
AX_WHISPER_PERSIST_PROF=$OUT/persist_phases.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python profiles/persist_prof.py $OUT/persist_phases.txt > $OUT/persist_phases_summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_b1.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc/fetch -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 1 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc/write -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 1 > $OUT/pmc_write.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summarize.py $OUT/pmc small_b1 > $OUT/pmc_summary.txt
cp profiles/r01_pmc_traffic.json $OUT/
find $OUT/stats_b1 -name "*kernel_stats.csv" -exec cp {} $OUT/b1_kernel_stats.csv \;
rm -rf $OUT/pmc $OUT/stats_b1
ls -la $OUT
