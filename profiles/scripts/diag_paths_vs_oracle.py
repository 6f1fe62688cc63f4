"""Diagnostic: teacher-forced logits of one clip through every decode path of the engine against the CPU oracle
(16-bit storage policy of the engine's build and pure fp32).
usage: diag_paths_vs_oracle.py <model> <dtype BF16|F16> <n_new> [seed] [B for the batched path]"""
import os
import sys
import tempfile

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "whisper.axera_amd", "tools"), os.path.join(R, "oracle")):
    sys.path.insert(0, p)
import modelgen  # noqa: E402
import oracle  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

model, dtype, n_new = sys.argv[1], sys.argv[2], int(sys.argv[3])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dims = modelgen.DIMS[model]
td = tempfile.mkdtemp()
w = modelgen.synth_weights(dims, seed, bf16=(dtype != "F16"))
if dtype == "F16":
    w = {k: v.astype(np.float16).astype(np.float32) for k, v in w.items()}
modelgen.write_model_dir(td, model, dims, weights=w, dtype=dtype, tiktoken_path=os.path.join(R, "tests", "golden", "multilingual.tiktoken"))
cfg = modelgen.make_config(model, dims)
clip = modelgen.synth_clip(0, 480000)
mel, _, _ = oracle.log_mel(clip, dims["n_mels"])
res = {}
for name, pol in (("policy16", 2 if dtype == "F16" else True), ("fp32", False)):
    o = oracle.Oracle(cfg, w, bf16_policy=pol, threads=64)
    ck, cv = o.encoder(mel)
    ids, lg = o.greedy(ck, cv, "zh", max_new=n_new, want_logits=True)
    res[name] = (ids, lg, ck, cv)
    print(name, "oracle ids", ids[:12], "logit std", float(lg.std()))
ids = res["policy16"][0]
forced = np.array([ids], dtype=np.int32)


def cmp(tag, lg):
    for name in ("policy16", "fp32"):
        ref = res[name][1]
        err = np.abs(lg[: len(ref)] - ref).max(axis=1)
        print(f"  {tag} vs {name} oracle: max {err.max():.3e} median {np.median(err):.3e} first steps {[round(float(x), 5) for x in err[:4]]}")


for mode in ("persistent", "graph"):
    if mode == "graph":
        os.environ["AX_WHISPER_DECODE"] = "graph"
    e = wa.Whisper(model, td, "zh", device=0, max_batch=B)
    e.encode_mel(mel)
    k, v = e.get_cross_kv(0)
    print(mode, "cross K/V vs policy16 oracle: max", float(np.abs(k - res["policy16"][2]).max()), float(np.abs(v - res["policy16"][3]).max()))
    lg1, _ = e.decode_forced(1, forced)
    cmp(f"{mode} 1-clip", lg1[0])
    if mode == "persistent":
        for nb in (3, B):
            e.encode_mel(np.stack([mel] * nb))
            lgb, _ = e.decode_forced(nb, np.repeat(forced, nb, axis=0))
            cmp(f"batched {nb} clips, slot 0", lgb[0])
            cmp(f"batched {nb} clips, slot {nb - 1}", lgb[nb - 1])
            print("   batched vs 1-clip:", float(np.abs(lgb[0] - lg1[0]).max()))
    e.close()
