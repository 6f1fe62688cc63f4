"""One-off soak of the persistent batch-1 decoder: N synthetic clips of different lengths through the persistent launch and
through the launch-per-phase path of a second handle; every id must agree (about 4.5 M in-launch hand-offs per clip).

    python profiles/scripts/soak_persistent.py [n_clips] [model]
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
model = sys.argv[2] if len(sys.argv) > 2 else "small"
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, model, f"{model}.safetensors")):
    modelgen.write_model_dir(mdir, model, modelgen.DIMS[model], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
ep = wa.Whisper(model, mdir, "zh", device=0, max_batch=1)
os.environ["AX_WHISPER_DECODE"] = "graph"
eg = wa.Whisper(model, mdir, "zh", device=0, max_batch=1)
assert ep.L.AX_WHISPER_GetConfigInt(ep.h, b"persistent_decode") == 1 and eg.L.AX_WHISPER_GetConfigInt(eg.h, b"persistent_decode") == 0
rng = np.random.default_rng(7)
bad = 0
t0 = time.time()
tp = tg = 0.0
for i in range(n):
    ns = int(rng.integers(4000, 480000))
    clip = modelgen.synth_clip(100 + i, ns)
    a = time.time()
    ids_p = ep.run_tokens(clip)
    b = time.time()
    ids_g = eg.run_tokens(clip)
    c = time.time()
    tp += b - a
    tg += c - b
    if ids_p != ids_g:
        bad += 1
        k = next((j for j, (x, y) in enumerate(zip(ids_p, ids_g)) if x != y), min(len(ids_p), len(ids_g)))
        print(f"clip {i} ({ns} samples): ids differ at {k}: {ids_p[k:k+4]} vs {ids_g[k:k+4]} (lengths {len(ids_p)}, {len(ids_g)})", flush=True)
    if i % 10 == 9:
        print(f"{i + 1} clips, {bad} mismatching, persistent {tp:.1f} s, graph {tg:.1f} s", flush=True)
assert ep.L.AX_WHISPER_GetConfigInt(ep.h, b"persistent_decode") == 1, "the persistent launch gave up during the soak"
print(f"done: {n} clips, {bad} mismatching; wall {time.time() - t0:.1f} s")
ep.close()
eg.close()
sys.exit(1 if bad else 0)
