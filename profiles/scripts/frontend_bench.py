import os, sys
R = "/root/repo"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import modelgen
import whisper_axera_amd as wa
mdir = "/tmp/axw_bench_models"
if not os.path.exists(os.path.join(mdir, "small", "small.safetensors")):
    modelgen.write_model_dir(mdir, "small", seed=0)
for B in (1, 16, 64):
    e = wa.Whisper("small", mdir, "zh", device=0, max_batch=B)
    e.bench("frontend", B, 0, 5)
    ms = e.bench("frontend", B, 0, 20) / 20
    print("B", B, "frontend ms %.4f" % ms, "per clip us %.1f" % (ms * 1e3 / B))
    e.close()
