#!/bin/bash
# Round 6's evidence on the MI355X box (profiles/README.md). usage: collect_r06.sh b1 | b2 | b3 | b64 | extra   (one gpurun call each)
# Results land in gpurun_out/prof6/; what is to be judged is copied into profiles/ as r06_* (profiles/scripts/install_r06.sh).
set -o pipefail
OUT=$PWD/gpurun_out/prof6
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pmc() {  # pmc <key> <driver args...>: FETCH_SIZE and WRITE_SIZE in separate passes, each with --kernel-trace only
  key=$1; shift
  rm -rf $OUT/pmc
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc/fetch -- python3 $R/profiles/pmc_driver.py "$@" > $OUT/pmc_fetch_$key.log 2>&1) || return 1
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc/write -- python3 $R/profiles/pmc_driver.py "$@" > $OUT/pmc_write_$key.log 2>&1) || return 1
  python3 profiles/pmc_summarize.py $OUT/pmc $key r06_pmc_traffic.json > $OUT/pmc_summary_$key.txt
  rm -rf $OUT/pmc
}
stats() {  # stats <name> <env assignments or ''> <bench args...>
  name=$1; envs=$2; shift 2
  # (the environment is set in this subshell: under rocprofv3 the program itself follows "--", never env / bash -c)
  (cd /tmp && export $envs && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $R/bench.py "$@" > $OUT/stats_$name.log 2>&1) || return 1
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/stats_$name
}
case "$1" in
b1)
  python bench.py > $OUT/bench_b1.json 2> $OUT/bench_b1.err || exit 1
  stats b1 "AXW_NOP=1" --steps 4 --warmup 2 --no-extras || exit 1
  pmc small_b1 1 || exit 1
  # in-kernel timeline of the persistent launch (per-phase sums + the absolute timeline of one mid-utterance layer)
  AX_WHISPER_PERSIST_PROF=$OUT/pp.txt python bench.py --steps 1 --warmup 1 --no-extras > $OUT/bench_b1_prof.json 2> $OUT/bench_b1_prof.err || exit 1
  python profiles/persist_prof.py $OUT/pp.txt > $OUT/persist_phases_summary.txt || exit 1
  ;;
b2)
  # two clips per call = one two-clip persistent launch (decode_persistent2.hip): kernel stats, PMC traffic, in-kernel timelines
  stats b2 "AXW_NOP=1" --batch 2 --steps 4 --warmup 2 --no-extras || exit 1
  pmc small_b2 2 || exit 1
  AX_WHISPER_PERSIST_PROF=$OUT/pp2_c0.txt python bench.py --batch 2 --steps 1 --warmup 1 --no-extras > $OUT/bench_b2_prof.json 2> $OUT/bench_b2_prof.err || exit 1
  python profiles/persist_prof.py $OUT/pp2_c0.txt > $OUT/persist2_phases_summary.txt || exit 1
  AX_WHISPER_PERSIST_PROF_CLIP=1 AX_WHISPER_PERSIST_PROF=$OUT/pp2_c1.txt python bench.py --batch 2 --steps 1 --warmup 1 --no-extras >> $OUT/bench_b2_prof.json 2>> $OUT/bench_b2_prof.err || exit 1
  python profiles/persist_prof.py $OUT/pp2_c1.txt > $OUT/persist2_phases_summary_clip1.txt || exit 1
  python bench.py --batch 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $OUT/bench_b2.json 2> $OUT/bench_b2.err || exit 1
  ;;
b3)
  # three clips per call = one three-clip persistent launch: kernel stats, PMC traffic, the line bench.py prints, a soak of ragged triples
  stats b3 "AXW_NOP=1" --batch 3 --steps 4 --warmup 2 --no-extras || exit 1
  pmc small_b3 3 || exit 1
  python bench.py --batch 3 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $OUT/bench_b3.json 2> $OUT/bench_b3.err || exit 1
  python profiles/scripts/soak_persistent2.py 40 small 3 > $OUT/soak_persistent3.txt 2>&1 || exit 1
  ;;
b64)
  stats b64 "AXW_NOP=1" --batch 64 --steps 1 --warmup 1 --no-extras || exit 1
  stats b64_1branch "AX_WHISPER_DECODE_BRANCHES=1" --batch 64 --steps 1 --warmup 1 --no-extras || exit 1
  pmc small_b64 64 || exit 1
  # the attention launches on one time axis, two branches in flight (production) and one branch
  AX_WHISPER_ATTN_STAMP=$OUT/stamps_b64.csv python profiles/scripts/attn_stamp.py 64 small > $OUT/attn_stamps_b64.txt 2> $OUT/attn_stamps_b64.err || exit 1
  AX_WHISPER_DECODE_BRANCHES=1 AX_WHISPER_ATTN_STAMP=$OUT/stamps_b64_1br.csv python profiles/scripts/attn_stamp.py 64 small > $OUT/attn_stamps_b64_1branch.txt 2> $OUT/attn_stamps_b64_1br.err || exit 1
  python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b64.json 2> $OUT/bench_b64.err || exit 1
  ;;
extra)
  pmc turbo_b16 16 turbo fp16 || exit 1
  pmc small_b256 256 small || exit 1
  AX_WHISPER_ATTN_STAMP=$OUT/stamps_b256.csv python profiles/scripts/attn_stamp.py 256 small > $OUT/attn_stamps_b256.txt 2> $OUT/attn_stamps_b256.err || exit 1
  AX_WHISPER_ATTN_STAMP=$OUT/stamps_turbo16.csv python profiles/scripts/attn_stamp.py 16 turbo fp16 > $OUT/attn_stamps_turbo_b16.txt 2> $OUT/attn_stamps_turbo16.err || exit 1
  python bench.py --model turbo --batch 16 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_turbo_fp16_b16.json 2> $OUT/bench_turbo_b16.err || exit 1
  python bench.py --model turbo --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_turbo_fp16_b1.json 2> $OUT/bench_turbo_b1.err || exit 1
  ;;
*) echo "usage: collect_r06.sh b1|b2|b3|b64|extra"; exit 2;;
esac
cp profiles/r06_pmc_traffic.json $OUT/r06_pmc_traffic.json
ls -la $OUT
