"""Randomised stress of the slot stream (AX_WHISPER_Stream*) for a fixed wall-clock time: streams of random slot counts are
opened and closed on one handle; clips of random lengths (0.2-40 s, silence, a clip longer than the front-end's staging
row) arrive in random group sizes with random id budgets (now and then the whole context), the step call takes random
step counts, slots are collected late or at once, a stream is sometimes closed with clips still in flight, and the
batched entry points are used between streams. Every collected id sequence must be a prefix-exact match of what the
same clip gives through the ragged batch path (same kernels), and the engine must stay usable throughout.
usage: stress_stream.py [model micro] [seconds 120] [max_slots 96] [seed 1]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "micro"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
max_slots = int(sys.argv[3]) if len(sys.argv) > 3 else 96
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, model, model + ".safetensors")):
    modelgen.write_model_dir(mdir, model, seed=0)
rng = np.random.default_rng(seed)

# a pool of distinct clips; the reference ids of a clip are its greedy run to the end of the context through the batch path
lens = [3200, 16000, 77777, 123457, 200000, 300000, 480000, 480000, 521000, 640000]
pool = [modelgen.synth_clip(900 + i, lens[i % len(lens)]) for i in range(20)]
pool.append(np.zeros(160000, np.float32))  # silence
e = wa.Whisper(model, mdir, "zh", device=0, max_batch=max_slots)
n_ctx_ids = len(e.run_tokens(pool[0], max_new=0))
ref = []
for g0 in range(0, len(pool), 7):
    ref += e.run_tokens_batch(pool[g0:g0 + 7], max_new=0)
assert all(len(r) == n_ctx_ids for r in ref)

t_end = time.time() + seconds
streams = clips_done = mismatches = aborted = 0
differing = {}
while time.time() < t_end:
    n_slots = int(rng.integers(1, max_slots + 1))
    n_clips = int(rng.integers(1, 3 * n_slots + 2))
    which = rng.integers(0, len(pool), n_clips)
    budgets = [0 if rng.random() < 0.03 else int(rng.integers(1, 60)) for _ in range(n_clips)]
    abort_after = int(rng.integers(0, n_clips)) if rng.random() < 0.1 else -1
    e.stream_open(n_slots)
    owner, free, held = {}, list(range(n_slots)), []
    nxt = got = 0
    while nxt < n_clips or owner or held:
        k = min(len(free), n_clips - nxt, int(rng.integers(0, n_slots + 1)))
        if k == 1:
            e.stream_admit(free[0], pool[which[nxt]], budgets[nxt])
        elif k > 1:
            e.stream_admit_batch(free[:k], [pool[i] for i in which[nxt:nxt + k]], budgets[nxt:nxt + k])
        for _ in range(k):
            owner[free.pop(0)] = nxt
            nxt += 1
        held += [sl for sl in e.stream_step(int(rng.integers(1, 17))) if sl in owner and sl not in held]
        # collect now, or leave finished slots sitting for a while
        for sl in list(held):
            if rng.random() < 0.7:
                ids = e.stream_collect(sl)
                c = owner.pop(sl)
                want = ref[which[c]][:budgets[c]] if budgets[c] else ref[which[c]]
                if ids != want:
                    mismatches += 1
                    differing.setdefault(int(which[c]), (ids, want))
                    print(f"MISMATCH stream {streams} slots {n_slots} clip {c} (pool {which[c]}, budget {budgets[c]}): {ids[:6]} .. vs {want[:6]} ..", flush=True)
                held.remove(sl)
                free.append(sl)
                got += 1
        if abort_after >= 0 and got >= abort_after:
            aborted += 1
            break
    e.stream_close()
    clips_done += got
    streams += 1
    if rng.random() < 0.3:  # the batched entry points between two streams
        i = int(rng.integers(0, len(pool)))
        b = int(rng.integers(1, 20))
        assert e.run_tokens(pool[i], max_new=b) is not None
        grp = [pool[int(j)] for j in rng.integers(0, len(pool), int(rng.integers(2, 9)))]
        assert len(e.run_tokens_batch(grp, max_new=b)) == len(grp)
# a difference against the batch path (other cross-attention split, other summation order) must be a numerical tie: the
# two candidates' logits at the first differing position, teacher-forced through the one-clip path
real = 0
for i, (ids, want) in differing.items():
    pos = next(k for k in range(min(len(ids), len(want))) if ids[k] != want[k])
    e.encode_mel(e.compute_mel(pool[i]))
    lg, _ = e.decode_forced(1, np.asarray(want[:max(pos, 1)], np.int32))
    margin = abs(float(lg[0, pos, ids[pos]]) - float(lg[0, pos, want[pos]]))
    top = float(np.max(lg[0, pos]))
    print(f"pool clip {i}: first difference at id {pos}: {ids[pos]} vs {want[pos]}, logits {lg[0, pos, ids[pos]]:.6f} / {lg[0, pos, want[pos]]:.6f} "
          f"(top {top:.6f}), margin {margin:.2e}", flush=True)
    real += margin > 2e-3
print(f"{model}: {streams} streams ({aborted} closed with clips in flight), {clips_done} clips collected, {mismatches} mismatches against the batch path, "
      f"{real} of them not a tie; persistent give-ups {e.L.AX_WHISPER_GetConfigInt(e.h, b'persistent_giveups')}", flush=True)
e.close()
sys.exit(1 if real else 0)
