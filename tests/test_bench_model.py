"""CPU: the algorithmic-byte model that bench.py's roofline uses equals SURVEY §8(d)'s per-unit figures."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))


def test_decode_step_bytes_match_survey():
    import bench
    import modelgen

    w, c, s = bench.decode_step_bytes(modelgen.DIMS["small"], 1, 0)
    assert abs(w / 1e6 - 277.8) < 0.1      # weights once per step for the whole batch (bf16)
    assert abs(c / 1e6 - 55.30) < 0.01     # cross K/V per clip
    assert abs(s / 1e3 - 36.9) < 0.05      # self K/V per clip and cached position
    w64, c64, s64 = bench.decode_step_bytes(modelgen.DIMS["small"], 64, 223)
    assert w64 == w and c64 == 64 * c and s64 == 64 * 224 * s
    assert abs((w64 + c64) / 1e9 - 3.82) < 0.01  # "3.82 GB/step @B=64" before the self K/V term
    wt, ct, _ = bench.decode_step_bytes(modelgen.DIMS["turbo"], 1, 0)
    assert abs(wt / 1e6 - 316.3) < 0.1 and abs(ct / 1e6 - 30.72) < 0.01


def test_bench_peaks_are_the_dense_figures():
    import bench

    assert bench.HBM_PEAK_GBS == 8000.0       # MI355X_MICROARCH.md: HBM3E ~8 TB/s
    assert bench.MFMA_BF16_PEAK_TF == 2500.0  # dense bf16 (the 5 PF headline includes 2:1 sparsity)
