"""Mixed-length serving soak: N clips whose greedy runs end after 60..150 ids (budgets stand in for eot — synthetic
weights never emit it; real utterances of 5-30 s end in that range) through
  (a) refilled slots (AX_WHISPER_Stream*: a finished slot takes the next clip at once), and
  (b) micro-batches of the same size (AX_WHISPER_RunPCMBatchTokens-style: a batch returns with its slowest clip),
both with host PCM, front-end and encoder included. Prints clips/s of both and checks the ids agree.
usage: soak_stream.py [model small] [n_clips 384] [slots 16,32,64]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "small"
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 384
slot_list = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16,32,64").split(",")]
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, model, model + ".safetensors")):
    modelgen.write_model_dir(mdir, model, seed=0)
base = [modelgen.synth_clip(i, 480000) for i in range(16)]
clips = [base[i % 16] for i in range(n_clips)]
rng = np.random.default_rng(7)
budgets = [int(x) for x in rng.integers(60, 151, n_clips)]
for n_slots in slot_list:
    e = wa.Whisper(model, mdir, "zh", device=0, max_batch=n_slots)
    e.run_stream(clips[:n_slots], n_slots, max_new=4)  # warm: graph capture, encoder workspaces
    t0 = time.perf_counter()
    got, calls = e.run_stream(clips, n_slots, max_new=budgets, steps_per_call=int(os.environ.get("AXW_STEPS_PER_CALL", "8")),
                              min_admit=int(os.environ.get("AXW_MIN_ADMIT", "1")))
    t_stream = time.perf_counter() - t0
    # micro-batches: front-end + encoder + ragged decode per group of n_slots (the existing batched entry points)
    e.run_tokens_batch(clips[:n_slots], max_new=4)
    # ragged micro-batches: every clip of a batch leaves at its own budget (no K/V streamed for it afterwards), the batch
    # returns with its slowest clip; mels precomputed (the front-end is 0.04 ms per clip at batch), encoder + decode timed
    mel16 = [e.compute_mel(c) for c in base]
    t0 = time.perf_counter()
    want = []
    for g0 in range(0, n_clips, n_slots):
        k = min(n_slots, n_clips - g0)
        e.encode_mel(np.stack([mel16[(g0 + i) % 16] for i in range(k)]))
        want += e.decode_greedy(k, max_new=150, max_new_clip=budgets[g0:g0 + k])
    t_batch_staged = time.perf_counter() - t0
    # the same micro-batches through the one-call entry point (uniform budget = the group's longest: what a batch costs)
    t0 = time.perf_counter()
    for g0 in range(0, n_clips, n_slots):
        grp = clips[g0:g0 + n_slots]
        e.run_tokens_batch(grp, max_new=max(budgets[g0:g0 + len(grp)]))
    t_batch = time.perf_counter() - t0
    same = sum(a == b for a, b in zip(got, want))
    print(f"{model} {n_clips} clips, ids 60-150 (mean {np.mean(budgets):.0f}), {n_slots} slots: refilled slots {n_clips / t_stream:.1f} clips/s "
          f"({calls} step calls) | ragged micro-batches {n_clips / t_batch_staged:.1f} clips/s (encoder + decode) | micro-batches run to "
          f"their longest clip {n_clips / t_batch:.1f} clips/s | ids identical {same}/{n_clips}", flush=True)
    e.close()
