#!/bin/bash
# A/B of the encoder GEMM tile structures in one process sequence on one device (3 = 256x256 two-stage, 4 = 256x256 phased)
set -e
cd profiles/microbench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../whisper.axera_amd/csrc gemm_shapes.cpp -o /tmp/gemm_shapes
for r in 1 2; do for t in ${TILES:-3 4}; do timeout -k 10 120 /tmp/gemm_shapes 20 $t; done; done
