"""Admission policies of the slot stream on the realistic-length workload (bench.py stream64): Whisper-small, N clips of 30 s through
64 refilled slots, per-clip budgets U{60..150}. usage: stream_policy_sweep.py [clips 384]   (SWEEP_WARM=k: k 64-clip decodes first)"""
import os, sys, time
R = "/root/repo"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import numpy as np
import torch  # noqa: F401  (first: libax_whisper.so then binds to the HIP runtime torch ships, as in bench.py — the system runtime measures ~10 % slower on this loop)
import modelgen
import whisper_axera_amd as wa
import bench
n_slots, n_clips = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 384
mdir = "/tmp/axw_bench_models"
if not os.path.exists(os.path.join(mdir, "small", "small.safetensors")):
    modelgen.write_model_dir(mdir, "small", seed=0)
distinct = [modelgen.synth_clip(i, 480000) for i in range(64)]
clips = [distinct[i % 64] for i in range(n_clips)]
budgets = bench.realistic_budgets(n_clips, seed=20260105)
e = wa.Whisper("small", mdir, "zh", device=0, max_batch=n_slots)
pol = [("admit as they come, 8 steps per call", dict()), ("4 steps per call", dict(steps_per_call=4)), ("2 steps per call", dict(steps_per_call=2)),
       ("wait for 4 free slots", dict(min_admit=4)), ("wait for 8 free slots", dict(min_admit=8)), ("wait for 16 free slots", dict(min_admit=16))]
if os.environ.get("SWEEP_WARM"):  # the state bench.py's stream64 leg runs in: seconds of solid GPU work right before it
    t0 = time.perf_counter()
    for _ in range(int(os.environ["SWEEP_WARM"])):
        e.run_tokens_batch(distinct, max_new=0)
    print(f"warm-up: {time.perf_counter() - t0:.2f} s of 64-clip full-context decodes", flush=True)
for rnd in range(2):
    for name, kw in pol:
        e.run_stream(clips[:128], n_slots, max_new=budgets[:128], **kw)
        t0 = time.perf_counter()
        got, calls = e.run_stream(clips, n_slots, max_new=budgets, **kw)
        dt = time.perf_counter() - t0
        assert [len(g) for g in got] == budgets
        print(f"{name:40s} {n_clips / dt:7.1f} clips/s  wall {dt:.3f} s  step calls {calls} x {kw.get('steps_per_call', 8)}", flush=True)
e.close()
