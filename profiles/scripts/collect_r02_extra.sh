#!/bin/bash
# Round-2 evidence beside collect_r02.sh (one GPU call): encoder GEMM microbenchmark per tile structure and with the
# epilogue compiled out, per-kernel durations and MFMA utilisation of the encoder at batch 64, the two soaks.
# Results land in gpurun_out/extra/; copy what is to be judged into profiles/ as r02_*.
OUT=$PWD/gpurun_out/extra
rm -rf $OUT; mkdir -p $OUT
( cd profiles/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../whisper.axera_amd/csrc gemm_shapes.cpp -o /tmp/gemm_shapes 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAXW_GEMM_NO_EPILOGUE -I../../include -I../../whisper.axera_amd/csrc gemm_shapes.cpp -o /tmp/gemm_noepi 2>/dev/null ) || exit 1
{
  echo "# profiles/microbench/gemm_shapes.cpp on one MI355X, random operands, 20 launches per shape; tile selection 3 = 256x256 two-stage"
  echo "# 32x32x16 loop (round 1), 4 = 256x256 phased 16x16x32 k-loop, 5 = the same k-loop as one k-tile stream per CU (the default"
  echo "# for launches of at least 256 square tiles), 0 = what launch_gemm picks. Last block: tile 4 built with -DAXW_GEMM_NO_EPILOGUE"
  echo "# (k-loop alone; nothing is written, so its spot check reports MISMATCH)."
  for t in 3 4 5 0; do timeout -k 10 120 /tmp/gemm_shapes 20 $t; done
  echo "# -DAXW_GEMM_NO_EPILOGUE"
  timeout -k 10 120 /tmp/gemm_noepi 20 4
} > $OUT/gemm_shapes.txt 2>&1
python profiles/encoder_bench.py 1 16 64 > $OUT/encoder_bench.txt 2>&1
bash profiles/scripts/enc_stats.sh 64 > /dev/null 2>&1 && cp gpurun_out/enc_stats/kernel_stats.csv $OUT/enc_b64_kernel_stats.csv
bash profiles/scripts/mfma_util.sh > $OUT/mfma_util.log 2>&1; cp gpurun_out/mfma/raw_summary.txt gpurun_out/mfma/derived_summary.txt $OUT/ 2>/dev/null
timeout -k 10 400 python profiles/scripts/soak_persistent.py 80 > $OUT/soak_persistent.txt 2>&1
timeout -k 10 400 python profiles/scripts/soak_batched.py 64 > $OUT/soak_batched.txt 2>&1
ls -la $OUT
