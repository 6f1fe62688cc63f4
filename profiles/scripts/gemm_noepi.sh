#!/bin/bash
# k-loop alone (no epilogue: outputs are not written, the spot check reports MISMATCH) for the phased 256x256 kernel
cd profiles/microbench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAXW_GEMM_NO_EPILOGUE -I../../include -I../../whisper.axera_amd/csrc gemm_shapes.cpp -o /tmp/gemm_noepi 2>/dev/null
timeout -k 10 120 /tmp/gemm_noepi 20 4
