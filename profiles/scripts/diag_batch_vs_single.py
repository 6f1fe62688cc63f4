"""Diagnostic: B clips through the batched decode sequence vs clip by clip through the 1-clip path, teacher-forced with
the 1-clip path's ids: per-step logit error and the top-2 margin wherever the argmax differs.
usage: diag_batch_vs_single.py <model> <dtype BF16|F16> <B> <n_new> [seed]"""
import os
import sys
import tempfile

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

model, dtype, B, n_new = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dims = modelgen.DIMS[model]
td = tempfile.mkdtemp()
w = modelgen.synth_weights(dims, seed, bf16=(dtype != "F16"))
if dtype == "F16":
    w = {k: v.astype(np.float16).astype(np.float32) for k, v in w.items()}
modelgen.write_model_dir(td, model, dims, weights=w, dtype=dtype, tiktoken_path=os.path.join(R, "tests", "golden", "multilingual.tiktoken"))
e = wa.Whisper(model, td, "zh", device=0, max_batch=B)
clips = [modelgen.synth_clip(i, 480000) for i in range(B)]
mels = np.stack([e.compute_mel(c) for c in clips])
single = []
for b in range(B):
    e.encode_mel(mels[b])
    single.append(e.decode_greedy(1, max_new=n_new)[0])
forced = np.array(single, dtype=np.int32)
e.encode_mel(mels)
lg_b, am_b = e.decode_forced(B, forced)
got = e.decode_greedy(B, max_new=n_new)
tot = 0
for b in range(B):
    e.encode_mel(mels[b])
    lg_1, am_1 = e.decode_forced(1, forced[b:b + 1])
    err = np.abs(lg_b[b] - lg_1[0]).max(axis=1)
    steps = [s for s in range(n_new) if am_b[b, s] != single[b][s]]
    tot += len(steps)
    srt = np.sort(lg_1[0], axis=1)
    print(f"clip {b}: max err {err.max():.3e} at step {int(err.argmax())}, median {np.median(err):.3e}; logit std {lg_1[0].std():.3f}; "
          f"median top-2 margin {np.median(srt[:, -1] - srt[:, -2]):.3e}; argmax differs at {steps}; "
          f"margins there {[float(srt[s, -1] - srt[s, -2]) for s in steps]}; err there {[float(err[s]) for s in steps]}; "
          f"greedy equal: {got[b] == single[b]}")
print("total differing steps", tot, "of", B * n_new)
e.close()
