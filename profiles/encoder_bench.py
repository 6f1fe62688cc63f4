"""Encoder-only timing at several batch sizes (TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, "small", "small.safetensors")):
    modelgen.write_model_dir(mdir, "small", seed=0)
for B in [int(x) for x in (sys.argv[1:] or ["1", "16", "64"])]:
    e = wa.Whisper("small", mdir, "zh", device=0, max_batch=B)
    ms = e.bench("encoder", B, 0, 5) / 5
    print("B", B, "encoder ms %.3f" % ms, "TFLOP/s %.1f" % (386.63e9 * B / (ms * 1e-3) / 1e12))
    e.close()
