"""One-off soak of the batched decode sequences: B synthetic clips of different lengths, full-context greedy decode
(444 ids each), through the clip-block sequence (LayerNorm prologue / residual epilogue / fused cross-attention queries /
register-resident vocabulary projection), through the older split-K sequence, and clip by clip through the batch-1
persistent launch. Reports how many id sequences agree and where the first difference sits.

    python profiles/scripts/soak_batched.py [B] [model]
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = sys.argv[2] if len(sys.argv) > 2 else "small"
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, model, f"{model}.safetensors")):
    modelgen.write_model_dir(mdir, model, modelgen.DIMS[model], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
rng = np.random.default_rng(11)
clips = [modelgen.synth_clip(300 + i, int(rng.integers(16000, 480000)) if i % 3 else 480000) for i in range(B)]


def run(env):
    for k in ("AX_WHISPER_BATCHED_LN",):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = wa.Whisper(model, mdir, "zh", device=0, max_batch=B)
    t0 = time.time()
    ids = e.run_tokens_batch(clips)
    dt = time.time() - t0
    e.close()
    return ids, dt


new, t_new = run({})
old, t_old = run({"AX_WHISPER_BATCHED_LN": "0"})
for k in ("AX_WHISPER_BATCHED_LN",):
    os.environ.pop(k, None)
e1 = wa.Whisper(model, mdir, "zh", device=0, max_batch=1)
single = [e1.run_tokens(c) for c in clips]
e1.close()


def compare(a, b, what):
    same = 0
    for i, (x, y) in enumerate(zip(a, b)):
        if x == y:
            same += 1
        else:
            k = next((j for j, (p, q) in enumerate(zip(x, y)) if p != q), min(len(x), len(y)))
            print(f"  {what}: clip {i} first differs at id {k} of {len(x)}/{len(y)}")
    print(f"{what}: {same}/{len(a)} id sequences identical ({sum(len(x) for x in a)} ids)")
    return same


print(f"clip-block sequence {t_new:.2f} s, split-K sequence {t_old:.2f} s for {B} clips")
n1 = compare(new, old, "clip-block vs split-K sequence")
n2 = compare(new, single, "clip-block (batched) vs persistent batch-1")
sys.exit(0 if n1 == B and n2 == B else 1)
