"""Decoder step of Whisper-small at batch N (default 64), step t = 224, as graph replays: whole step, the linear layers
alone and the attention launches alone; minimum and median of 5 x 200 replays each (for profiles/scripts/ab_lib.sh).
usage: ab_step.py [clips 64] [model small] [fp16]"""
import os
import statistics
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = sys.argv[2] if len(sys.argv) > 2 else "small"
fp16 = len(sys.argv) > 3 and sys.argv[3] == "fp16"
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models") + ("_f16" if fp16 else "")
if not os.path.exists(os.path.join(mdir, model, model + ".safetensors")):
    modelgen.write_model_dir(mdir, model, seed=0, dtype="F16" if fp16 else "BF16")
e = wa.Whisper(model, mdir, "zh", device=0, max_batch=B)
e.bench("decode_step", B, 224, 50)
out = []
for what in ("decode_step", "decode_gemv", "decode_attn"):
    t = [e.bench(what, B, 224, 200) / 200 for _ in range(5)]
    out.append("%s min %.4f med %.4f ms" % (what, min(t), statistics.median(t)))
print("batch", B, model, " | ".join(out))
e.close()
