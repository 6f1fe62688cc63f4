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


def _run_bench(extra_args, extra_env):
    import json
    import subprocess

    env = dict(os.environ, AXW_BENCH_REHEARSAL="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if k not in extra_env:
            env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True,
                       env=env, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, [json.loads(l) for l in lines], r.stderr


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` (the driver's command shape) must run TWO ranks and gather both shards: the
    rehearsal mode swaps the engine for a stand-in and RCCL for gloo, everything else is the code the GPU run uses."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3"], {})
    assert rc == 0, err
    assert len(lines) == 1, "stdout must carry exactly ONE JSON line"
    j = lines[0]
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 6 and j["config"]["parallelism"] == "dp2"
    assert j["gathered_rows"] == 6 and j["scaling"] == "weak"
    assert j["value"] is None and "REHEARSAL" in j["data"]  # a rehearsal can never be mistaken for a measurement


def test_bench_default_batch_is_64_per_gpu_beyond_one_gpu():
    import bench

    a = bench.parse_args(["--gpus", "8"])
    assert a.batch == 0  # resolved per rank: 1 at N=1 (configs[1]), 64/GPU at N>1 (configs[4]: 8 x 64 = 512)


def test_bench_rejects_a_rank_count_that_differs_from_gpus():
    rc, lines, err = _run_bench(["--gpus", "4"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not lines and "--gpus 4" in err


def test_bench_gpus_8_rehearsal_is_configs4():
    """BASELINE configs[4] — 8 ranks x 64 clips = 512 — as far as it can be run without eight GPUs: the rank spawn, the
    default of 64 clips per GPU, the gather of all 512 id rows, the single-GPU reference leg behind the barrier and the ONE
    JSON line (gloo, stand-in engine)."""
    rc, lines, err = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1"], {"OMP_NUM_THREADS": "1"})
    assert rc == 0, err
    assert len(lines) == 1
    j = lines[0]
    assert j["n_gpus"] == 8 and j["config"]["batch_per_gpu"] == 64 and j["config"]["global_batch"] == 512
    assert j["gathered_rows"] == 512 and j["config"]["parallelism"] == "dp8" and j["scaling"] == "weak"
    assert j["single_gpu_same_batch"]["steps"] == 2 and "weak_scaling_efficiency" not in j  # the same-run one-GPU point; no efficiency claim
    assert j["value"] is None and "REHEARSAL" in j["data"]


def test_bench_exits_nonzero_when_a_rank_dies_mid_step():
    """World 2, rank 1 raises inside its second step while rank 0 sits in the gather: the launcher must come back with a
    non-zero code (no hang at the barrier) and must not print a result line."""
    import time

    t0 = time.time()
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "0", "--batch", "2"],
                                {"AXW_BENCH_FAIL_RANK": "1", "AXW_BENCH_TIMEOUT_S": "120"})
    assert rc != 0 and not lines, (rc, lines)
    assert "injected failure" in err
    assert time.time() - t0 < 200
