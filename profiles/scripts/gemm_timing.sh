#!/bin/bash
# Where a tile's time goes in the stream GEMM (gemm256ps_bf16_kernel): one stamped launch per shape (-DAXW_GEMM_TIMING).
set -e
cd profiles/microbench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -DAXW_GEMM_TIMING -I../../include -I../../whisper.axera_amd/csrc gemm_shapes.cpp -o /tmp/gemm_timing
timeout -k 10 120 /tmp/gemm_timing 10 5
