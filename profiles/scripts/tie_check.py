"""Top-2 logit margin at the first id where the batched soak (soak_batched.py) differed from the batch-1 path.
    python profiles/scripts/tie_check.py [clip [id]]     (round 1: clip 49, id 281; round 2: clip 41, id 293)"""
import os, sys
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import numpy as np, modelgen, whisper_axera_amd as wa
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, "small", "small.safetensors")):
    modelgen.write_model_dir(mdir, "small", modelgen.DIMS["small"], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
rng = np.random.default_rng(11)
clips = [modelgen.synth_clip(300 + i, int(rng.integers(16000, 480000)) if i % 3 else 480000) for i in range(64)]
CLIP = int(sys.argv[1]) if len(sys.argv) > 1 else 49
ID = int(sys.argv[2]) if len(sys.argv) > 2 else 281
e = wa.Whisper("small", mdir, "zh", device=0, max_batch=1)
ids = e.run_tokens(clips[CLIP])
e.encode_mel(e.compute_mel(clips[CLIP]))
lg, am = e.decode_forced(1, np.array([ids[:ID + 19]], dtype=np.int32))
row = lg[0][ID]
top = np.sort(row)[-3:]
print(f"clip {CLIP}: persistent id[{ID}] =", ids[ID], "argmax of forced logits:", int(row.argmax()), "top-3 logits:", top, "margin:", top[-1] - top[-2])
e.close()
