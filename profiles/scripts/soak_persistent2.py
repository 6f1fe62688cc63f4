"""Soak of the two-clip persistent launch: N random pairs of synthetic clips (different lengths, different per-clip id budgets)
through ONE two-clip launch per pair, against the same clips decoded alone through the one-clip launch — the two launches run
the same arithmetic in the same order, so every id must be EQUAL (about 9 M in-launch hand-offs per pair).

    python profiles/scripts/soak_persistent2.py [n_pairs] [model]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import numpy as np  # noqa: E402

import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
model = sys.argv[2] if len(sys.argv) > 2 else "small"
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, model, f"{model}.safetensors")):
    modelgen.write_model_dir(mdir, model, modelgen.DIMS[model], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
e2 = wa.Whisper(model, mdir, "zh", device=0, max_batch=2)
e1 = wa.Whisper(model, mdir, "zh", device=0, max_batch=1)
assert e2.L.AX_WHISPER_GetConfigInt(e2.h, b"persistent_two_clips") == 1
rng = np.random.default_rng(11)
bad = 0
t0 = time.time()
t_pair = t_single = 0.0
for i in range(n):
    clips = [modelgen.synth_clip(300 + 2 * i + k, int(rng.integers(4000, 480000))) for k in range(2)]
    budgets = [int(rng.integers(1, 445)) if rng.random() < 0.7 else 0 for _ in range(2)]  # 0: the whole context
    mels = np.stack([e2.compute_mel(c) for c in clips])
    a = time.time()
    e2.encode_mel(mels)
    got = e2.decode_greedy(2, max_new=0, max_new_clip=budgets)
    b = time.time()
    want = []
    for k in range(2):
        e1.encode_mel(mels[k])
        want.append(e1.decode_greedy(1, max_new=budgets[k])[0])
    c = time.time()
    t_pair += b - a
    t_single += c - b
    if got != want:
        bad += 1
        for k in range(2):
            if got[k] != want[k]:
                j = next((j for j, (x, y) in enumerate(zip(got[k], want[k])) if x != y), min(len(got[k]), len(want[k])))
                print(f"pair {i} clip {k} (budget {budgets[k]}): ids differ at {j}: {got[k][j:j+4]} vs {want[k][j:j+4]} (lengths {len(got[k])}, {len(want[k])})", flush=True)
    if i % 10 == 9:
        print(f"{i + 1} pairs, {bad} mismatching, two-clip launches {t_pair:.1f} s, one-clip launches {t_single:.1f} s", flush=True)
g = lambda e, k: e.L.AX_WHISPER_GetConfigInt(e.h, k)
assert g(e2, b"persistent_giveups") == 0 and g(e1, b"persistent_giveups") == 0, "a persistent launch gave up during the soak"
print(f"done: {n} pairs, {bad} mismatching; wall {time.time() - t0:.1f} s (two-clip launches {t_pair:.1f} s, one-clip launches {t_single:.1f} s)")
e2.close()
e1.close()
sys.exit(1 if bad else 0)
