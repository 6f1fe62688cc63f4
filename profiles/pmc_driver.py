"""Tiny workload for rocprofv3 --pmc passes: a few replays of one captured decode step (and one encoder pass).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/fetch -- python3 profiles/pmc_driver.py 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/write -- python3 profiles/pmc_driver.py 1
    python3 profiles/pmc_summarize.py <out> small_b1      # -> profiles/r01_pmc_traffic.json
(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md, rocprofv3 PMC slots)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = sys.argv[2] if len(sys.argv) > 2 else "small"          # round 4: pmc_driver.py 16 turbo fp16 / 256 small
fp16 = len(sys.argv) > 3 and sys.argv[3] == "fp16"
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models") + ("_f16" if fp16 else "")
if not os.path.exists(os.path.join(mdir, model, model + ".safetensors")):
    modelgen.write_model_dir(mdir, model, seed=0, dtype="F16" if fp16 else "BF16")
e = wa.Whisper(model, mdir, "zh", device=0, max_batch=B)
e.bench("encoder", B, 0, 1)
e.bench("decode_step", B, 224, 4)
if B == 1:  # the persistent batch-1 decode launch: one whole utterance (448 steps) = one launch
    import numpy as np

    e.run_tokens(modelgen.synth_clip(0, 480000))
if B in (2, 3):  # the multi-clip persistent launch: two or three whole utterances = one launch
    e.run_tokens_batch([modelgen.synth_clip(i, 480000) for i in range(B)])
e.close()
