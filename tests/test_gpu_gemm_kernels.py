"""GPU: the encoder GEMM kernels one by one against an fp64 host reference (profiles/microbench/gemm_shapes.cpp compiles
csrc/gemm.hip unchanged and spot-checks 4000 random outputs per shape: bias, GELU, in-place residual and Q / K / V^T
epilogues at the encoder's own shapes at 64 clips, 4096^3, and a one-clip launch).

launch_gemm picks a kernel by tile count, so the end-to-end parity tests only ever run the choices it makes for Whisper's
dimensions (128x128 at one clip, the 256x128 ring at a few clips, the one-stream-per-CU 256x256 kernel at batch); here
every kernel is forced on every shape."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gemm_shapes(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gemm") / "gemm_shapes")
    src = os.path.join(ROOT, "profiles", "microbench", "gemm_shapes.cpp")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "whisper.axera_amd", "csrc"), src, "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


# 0: the launcher's own choice; 1: 128x128; 2: 256x128 ring; 5: 256x256, one k-tile stream per CU
@pytest.mark.parametrize("tile", [0, 1, 2, 5])
def test_gemm_kernel_vs_fp64_reference(gemm_shapes, tile):
    r = subprocess.run([gemm_shapes, "2", str(tile)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if "TFLOP/s" in ln]
    assert len(lines) == 7, r.stdout
    for ln in lines:
        assert ln.rstrip().endswith("ok"), ln
