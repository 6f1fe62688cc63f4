"""Soak of the multi-clip persistent launch: N random groups of 2 or 3 synthetic clips (different lengths, different per-clip id
budgets) through ONE launch per group, against the same clips decoded alone through the one-clip launch (from the same encoder output) — the two launches run
the same arithmetic in the same order, so every id must be EQUAL (about 9 M in-launch hand-offs per pair).

    python profiles/scripts/soak_persistent2.py [n_groups] [model] [clips per group: 2 | 3]
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
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, model, f"{model}.safetensors")):
    modelgen.write_model_dir(mdir, model, modelgen.DIMS[model], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
e2 = wa.Whisper(model, mdir, "zh", device=0, max_batch=G)
assert e2.L.AX_WHISPER_GetConfigInt(e2.h, b"persistent_max_clips") >= G
rng = np.random.default_rng(11)
bad = 0
t0 = time.time()
t_pair = t_single = 0.0
for i in range(n):
    clips = [modelgen.synth_clip(300 + G * i + k, int(rng.integers(4000, 480000))) for k in range(G)]
    budgets = [int(rng.integers(1, 445)) if rng.random() < 0.7 else 0 for _ in range(G)]  # 0: the whole context
    mels = np.stack([e2.compute_mel(c) for c in clips])
    a = time.time()
    e2.encode_mel(mels)
    got = e2.decode_greedy(G, max_new=0, max_new_clip=budgets)
    b = time.time()
    want = []
    for k in range(G):  # the same clip alone, from the SAME encoder pass shape (a batch of G with clip k first: the encoder's one-clip
        e2.encode_mel(np.stack([mels[(k + j) % G] for j in range(G)]))  # passes split K and round differently — that would test ties)
        want.append(e2.decode_greedy(1, max_new=budgets[k])[0])
    c = time.time()
    t_pair += b - a
    t_single += c - b
    if got != want:
        bad += 1
        for k in range(G):
            if got[k] != want[k]:
                j = next((j for j, (x, y) in enumerate(zip(got[k], want[k])) if x != y), min(len(got[k]), len(want[k])))
                print(f"group {i} clip {k} (budget {budgets[k]}): ids differ at {j}: {got[k][j:j+4]} vs {want[k][j:j+4]} (lengths {len(got[k])}, {len(want[k])})", flush=True)
    if i % 10 == 9:
        print(f"{i + 1} groups of {G}, {bad} mismatching, multi-clip launches {t_pair:.1f} s, one-clip launches {t_single:.1f} s", flush=True)
g = lambda e, k: e.L.AX_WHISPER_GetConfigInt(e.h, k)
assert g(e2, b"persistent_giveups") == 0, "a persistent launch gave up during the soak"
print(f"done: {n} groups of {G}, {bad} mismatching; wall {time.time() - t0:.1f} s (multi-clip launches {t_pair:.1f} s, one-clip launches {t_single:.1f} s)")
e2.close()
sys.exit(1 if bad else 0)
