"""Top-2 logit margin at the first id where the batched soak (soak_batched.py) differed from the batch-1 path: clip 49, id 281."""
import os, sys
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import numpy as np, modelgen, whisper_axera_amd as wa
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, "small", "small.safetensors")):
    modelgen.write_model_dir(mdir, "small", modelgen.DIMS["small"], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
rng = np.random.default_rng(11)
clips = [modelgen.synth_clip(300 + i, int(rng.integers(16000, 480000)) if i % 3 else 480000) for i in range(64)]
e = wa.Whisper("small", mdir, "zh", device=0, max_batch=1)
ids = e.run_tokens(clips[49])
e.encode_mel(e.compute_mel(clips[49]))
lg, am = e.decode_forced(1, np.array([ids[:300]], dtype=np.int32))
row = lg[0][281]
top = np.sort(row)[-3:]
print("persistent id[281] =", ids[281], "argmax of forced logits:", int(row.argmax()), "top-3 logits:", top, "margin:", top[-1] - top[-2])
e.close()
