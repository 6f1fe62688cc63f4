#!/bin/bash
# Round 3's evidence on the MI355X box (see profiles/README.md). usage: collect_r03.sh b1 | b64 (one gpurun call each).
# Results land in gpurun_out/prof/; what is to be judged is copied into profiles/ as r03_*.
set -o pipefail
OUT=$PWD/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
cp profiles/r03_pmc_traffic.json $OUT/pmc_before.json 2>/dev/null
if [ "$1" != "b64" ]; then
# the driver's line: configs[1] as value + batch64 (configs[2]) + turbo_fp16_b16 (configs[3]) + config0 + cpu_baseline
python bench.py > $OUT/bench_b1.json 2> $OUT/bench_b1.err || exit 1
# kernel stats of the headline workload ONLY (--no-extras: every decode_persistent_kernel launch in the trace is one
# 448-step clip of the timed workload, so AverageNs IS the launch the roofline is quoted on)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-extras > $OUT/stats_b1.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc/fetch -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 1 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc/write -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 1 > $OUT/pmc_write.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summarize.py $OUT/pmc small_b1 r03_pmc_traffic.json > $OUT/pmc_summary.txt
find $OUT/stats_b1 -name "*kernel_stats.csv" -exec cp {} $OUT/b1_kernel_stats.csv \;
rm -rf $OUT/pmc $OUT/stats_b1
cp profiles/r03_pmc_traffic.json $OUT/r03_pmc_traffic_b1.json; ls -la $OUT; exit 0
fi
# ---- batch 64 (configs[2]): the step as it runs in production (2 graph branches; the profiler serialises them) and with ONE
# branch (AX_WHISPER_DECODE_BRANCHES=1: every attention launch covers all 64 clips, so its duration in the summary is the
# duration of the launch the roofline is quoted on)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b64 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 1 --no-extras > $OUT/stats_b64.log 2>&1 || exit 1
AX_WHISPER_DECODE_BRANCHES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b64_1br -- python3 $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 1 --no-extras > $OUT/stats_b64_1br.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc/fetch -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 64 > $OUT/pmc_fetch64.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc/write -- python3 $GRAFT_REPO_ROOT/profiles/pmc_driver.py 64 > $OUT/pmc_write64.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summarize.py $OUT/pmc small_b64 r03_pmc_traffic.json > $OUT/pmc_summary64.txt
find $OUT/stats_b64 -name "*kernel_stats.csv" -exec cp {} $OUT/b64_kernel_stats.csv \;
find $OUT/stats_b64_1br -name "*kernel_stats.csv" -exec cp {} $OUT/b64_1branch_kernel_stats.csv \;
rm -rf $OUT/pmc $OUT/stats_b64 $OUT/stats_b64_1br
AX_WHISPER_DECODE_BRANCHES=1 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b64_1branch.json 2> $OUT/bench_b64_1br.err || exit 1
python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b64.json 2> $OUT/bench_b64.err || exit 1
python bench.py --model turbo --batch 16 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_turbo_fp16_b16.json 2> $OUT/bench_turbo_b16.err || exit 1
python bench.py --model turbo --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_turbo_fp16_b1.json 2> $OUT/bench_turbo_b1.err || exit 1
cp profiles/r03_pmc_traffic.json $OUT/r03_pmc_traffic_b64.json
ls -la $OUT
