#!/bin/bash
# Collects round 2's evidence on the MI355X box: bench lines, rocprofv3 kernel stats, PMC traffic (see profiles/README.md).
# usage: collect_r02.sh b1 | b64   (two gpurun calls: each part stays inside one call's time limit). Results land in
# gpurun_out/prof/; copy what is to be judged into profiles/ as r02_*.
set -o pipefail
OUT=$PWD/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$1" != "b64" ]; then
python bench.py > $OUT/bench_b1.json 2> $OUT/bench_b1.err || exit 1
python bench.py --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-batch64 > $OUT/bench_b1_fp16.json 2> $OUT/bench_b1_fp16.err || exit 1
python bench.py --model turbo --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_turbo_fp16_b1.json 2> $OUT/bench_turbo_b1.err || exit 1
AX_WHISPER_DECODE=graph python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch64 > $OUT/bench_b1_graph.json 2> $OUT/bench_b1_graph.err || exit 1
AX_WHISPER_PERSIST_PROF=$OUT/persist_phases.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch64 > /dev/null 2>&1
python profiles/persist_prof.py $OUT/persist_phases.txt > $OUT/persist_phases_summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch64 > $OUT/stats_b1.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc/fetch -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 1 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc/write -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 1 > $OUT/pmc_write.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summarize.py $OUT/pmc small_b1 r02_pmc_traffic.json > $OUT/pmc_summary.txt
find $OUT/stats_b1 -name "*kernel_stats.csv" -exec cp {} $OUT/b1_kernel_stats.csv \;
rm -rf $OUT/pmc $OUT/stats_b1
cp profiles/r02_pmc_traffic.json $OUT/r02_pmc_traffic_b1.json; ls -la $OUT; exit 0
fi
# ---- batch 64 (BASELINE configs[2]) and turbo fp16 at batch 16 (configs[3])
python bench.py --model turbo --batch 16 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_turbo_fp16_b16.json 2> $OUT/bench_turbo_b16.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b64 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/stats_b64.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc/fetch -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 64 > $OUT/pmc_fetch64.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc/write -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 64 > $OUT/pmc_write64.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summarize.py $OUT/pmc small_b64 r02_pmc_traffic.json > $OUT/pmc_summary64.txt
find $OUT/stats_b64 -name "*kernel_stats.csv" -exec cp {} $OUT/b64_kernel_stats.csv \;
rm -rf $OUT/pmc $OUT/stats_b64
python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b64.json 2> $OUT/bench_b64.err || exit 1
cp profiles/r02_pmc_traffic.json $OUT/r02_pmc_traffic_b64.json
ls -la $OUT
