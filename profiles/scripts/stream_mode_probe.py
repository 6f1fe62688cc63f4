"""Does the slot stream's rate (bench.py stream64: 328-341 or 365-371 clips/s, drawn per process start) go with which of the
engine's streams share a hardware queue?  One process = one line: the concurrency mask of Engine::bench("queue_probe")
(bit 0 main-admission, 1 main-branch0, 2 main-copies, 3 admission-branch0, 4 admission-copies, 5 branch0-copies) and the
stream64 rate of the same engine.     python profiles/scripts/stream_mode_probe.py [n_slots] [n_clips]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import numpy as np  # noqa: E402

import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 384
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, "small", "small.safetensors")):
    modelgen.write_model_dir(mdir, "small", modelgen.DIMS["small"], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
distinct = [modelgen.synth_clip(i, 480000) for i in range(min(n_slots, 64))]
clips = [distinct[i % len(distinct)] for i in range(n_clips)]
budgets = [int(x) for x in np.random.Generator(np.random.PCG64(20260105)).integers(60, 151, size=n_clips)]
eng = wa.Whisper("small", mdir, "zh", device=0, max_batch=n_slots)
mask = int(eng.bench("queue_probe", 1, 0, 1))
eng.run_stream(clips[: 2 * n_slots], n_slots, max_new=budgets[: 2 * n_slots])
rates = []
for _ in range(2):
    t0 = time.perf_counter()
    got, calls = eng.run_stream(clips, n_slots, max_new=budgets)
    rates.append(n_clips / (time.perf_counter() - t0))
mask2 = int(eng.bench("queue_probe", 1, 0, 1))
print(f"queues side by side: mask {mask:06b} (after the runs {mask2:06b}); stream{n_slots}: {rates[0]:.1f} {rates[1]:.1f} clips/s", flush=True)
eng.close()
