import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import whisper_axera_amd as wa
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
eng = wa.Whisper("small", mdir, "zh", device=0, max_batch=64)
ms = [eng.bench("decode_step", 64, 224, 100) / 100 for _ in range(3)]
print("pad", os.environ.get("AX_WHISPER_PAD_STREAMS"), "step ms", [round(m,4) for m in ms], flush=True)
eng.close()
