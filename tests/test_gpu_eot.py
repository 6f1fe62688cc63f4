"""GPU: the greedy loop ENDS ON eot (Whisper.cpp:219-222 `while (idx != WHISPER_EOT && offset < n_text_ctx)`), on every
decode path, at a different step for every clip of a batch — not on an id budget standing in for it.

tests/eot_case.py shapes a seeded model until the oracle's greedy run emits eot after 1..53 ids depending on the audio
(every step of every run keeps a margin that numerical differences cannot cross, so ids must be EQUAL: no tie rule here).
Checked per path: the number of ids, every id, that eot itself is not among them (Whisper.cpp:220 pushes the token before
the step that produces eot), and — slot stream — that the slot is reported finished, handed back and refilled.

  persistent launches (1 clip; 2 and 3 clips = one multi-clip launch)   decode_persistent.hip, decode_persistent2.hip
  launch-per-phase path / GEMV family (AX_WHISPER_DECODE=graph)   advance_kernel, decode_gemv.hip
  clip-block sequence, 3 / 6 / 20 / 64 clips (1 and 2 graph branches, 1-4 clip blocks)
  split-K sequence (AX_WHISPER_BATCHED_LN=0; the d_model > 1024 path) at 6 clips, and at d = 1280 in fp16
  slot stream: 40 clips through 4 and 7 slots
  host-pointer entry points (RunPCMBatchTokens, RunPCM text)
"""
import numpy as np
import pytest
import torch  # noqa: F401

from eot_case import EotCase

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def micro_eot(oracle_mod, tmp_path_factory):
    case = EotCase("micro", 11)
    case.root = case.write(tmp_path_factory.mktemp("models_eot_micro"))
    case.sel = case.select(64)
    stops = [len(x) for x in case.sel[2]]
    print("eot model (micro): gain", case.g, "bias", round(case.beta0, 3), "stops of 64 clips:", stops)
    assert len(set(stops[:16])) >= 8 and max(stops) < case.budget  # every clip ends on eot, at many different steps
    return case


def _check(case, got, want, what):
    assert len(got) == len(want), what
    for b, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w), (what, "clip", b, "stopped after", len(g), "ids, the oracle after", len(w))
        assert g == w, (what, "clip", b)
        assert case.eot not in g


@pytest.mark.parametrize("mode,batch", [("persistent", 1), ("persistent", 2), ("persistent", 3), ("graph", 1), ("graph", 2), ("cblock", 3), ("cblock", 6),
                                        ("cblock", 20), ("cblock", 64), ("splitk", 6)])
def test_loop_ends_on_eot(built_lib, micro_eot, monkeypatch, mode, batch):
    case = micro_eot
    if mode == "graph":
        monkeypatch.setenv("AX_WHISPER_DECODE", "graph")
    if mode == "splitk":
        monkeypatch.setenv("AX_WHISPER_BATCHED_LN", "0")
    if mode == "cblock" and batch == 3:
        monkeypatch.setenv("AX_WHISPER_PERSIST2", "2")   # three clips through the clip-block sequence (default: one three-clip launch)
    _, clips, want = case.sel
    e = built_lib.Whisper("micro", case.root, "zh", device=0, max_batch=batch)
    try:
        g = lambda k: e.L.AX_WHISPER_GetConfigInt(e.h, k.encode())
        assert g("persistent_decode") == (0 if mode == "graph" else 1)
        assert g("batched_ln") == (0 if mode == "splitk" else 1)
        if mode == "persistent" or batch <= 2:  # one group after the other through the same slot(s), so a second run starts from a used state
            for b0 in range(0, 9 - batch, batch):
                mels = np.stack([e.compute_mel(c) for c in clips[b0:b0 + batch]])
                e.encode_mel(mels)
                _check(case, e.decode_greedy(batch, max_new=case.budget), want[b0:b0 + batch], f"{mode} clips {b0}..")
                if mode == "persistent":
                    assert g("persistent_giveups") == 0
        else:
            mels = np.stack([e.compute_mel(c) for c in clips[:batch]])
            e.encode_mel(mels)
            _check(case, e.decode_greedy(batch, max_new=case.budget), want[:batch], f"{mode} {batch} clips")
            # the whole context as the budget (max_new 0): eot still ends every clip; and a budget BELOW a clip's stop wins over eot
            _check(case, e.decode_greedy(batch), want[:batch], f"{mode} {batch} clips, no budget")
            cut = [max(1, len(w) - 3) if b % 2 else case.budget for b, w in enumerate(want[:batch])]
            got = e.decode_greedy(batch, max_new=case.budget, max_new_clip=cut)
            _check(case, got, [w[:c] for w, c in zip(want[:batch], cut)], f"{mode} {batch} clips, budgets below the stop")
        # host-pointer entry point: front-end + encoder + loop in one call
        n = min(batch, 4)
        _check(case, e.run_tokens_batch(clips[:n], max_new=case.budget), want[:n], f"{mode} RunPCMBatchTokens")
    finally:
        e.close()


def test_text_entry_points_end_on_eot(built_lib, micro_eot):
    """AX_WHISPER_RunPCM / RunPCMBatch (ax_whisper_api.cpp:141-163): the text is the detokenised ids up to, not including, eot."""
    case = micro_eot
    _, clips, want = case.sel
    e = built_lib.Whisper("micro", case.root, "zh", device=0, max_batch=4)
    try:
        texts = e.run_batch(clips[:4])
        for b in range(4):
            assert texts[b] == e.transcript(want[b])   # ids -> text exactly as RunPCM* returns it (zh post-pass included)
            assert e.run(clips[b]) == texts[b]
    finally:
        e.close()


@pytest.mark.parametrize("n_slots", [4, 7])
def test_slot_stream_frees_a_slot_on_eot(built_lib, micro_eot, n_slots):
    """AX_WHISPER_Stream*: a slot whose clip emitted eot is reported, collected and refilled while the others decode on."""
    case = micro_eot
    _, clips, want = case.sel
    n_clips = 40
    e = built_lib.Whisper("micro", case.root, "zh", device=0, max_batch=8)
    try:
        got, calls = e.run_stream(clips[:n_clips], n_slots, max_new=0, steps_per_call=4)
        _check(case, got, want[:n_clips], f"stream {n_slots} slots")
        # had the slots run to the context's end instead, 40 clips through n_slots would need 40 / n_slots * 444 / 4 step calls
        total_steps = sum(len(w) + 4 for w in want[:n_clips])
        print(f"{n_clips} clips through {n_slots} slots in {calls} calls of 4 steps; sum of steps {total_steps}")
        assert calls < 2 * total_steps / (4 * n_slots) + 3 * n_clips
        # slot by slot: admit, step until reported, collect, and the SAME slot takes the next clip at once
        e.stream_open(3)
        try:
            for i in (5, 9, 2):
                e.stream_admit(1, clips[i], 0)
                fin = []
                for _ in range(200):
                    fin = e.stream_step(2)
                    if fin:
                        break
                assert fin == [1]
                assert e.stream_collect(1) == want[i]
        finally:
            e.stream_close()
    finally:
        e.close()


@pytest.mark.parametrize("model_type,dtype,batches,shape", [
    ("w512", "BF16", (1, 5), {}), ("w1280", "F16", (1, 5, 18), {}),
    ("small", "BF16", (1, 6), dict(row0_scale=2.0, cross_scale=4.0, ramp=8.0))])   # BASELINE configs[1]/[2] dims: 12 layers add up
def test_loop_ends_on_eot_other_widths(built_lib, oracle_mod, tmp_path, model_type, dtype, batches, shape):
    """The persistent launch is one template instantiation per d_model and d_model > 1024 runs the split-K sequence: eot at
    d = 512 (bf16), d = 1280 (fp16 build) and at the full Whisper-small size, one clip and batches."""
    case = EotCase(model_type, 5, dtype=dtype, n_cal=6, budget=40, **shape)
    root = case.write(tmp_path / "m")
    _, clips, want = case.select(max(batches), min_margin=0.02)
    stops = [len(x) for x in want]
    print(f"eot model ({model_type} {dtype}): gain {case.g} bias {case.beta0:.3f} stops {stops}")
    assert len(set(stops)) >= 3
    e = built_lib.Whisper(model_type, root, "zh", device=0, max_batch=max(batches))
    try:
        mels = np.stack([e.compute_mel(c) for c in clips])
        for B in batches:
            if B == 1:
                for b in range(3):
                    e.encode_mel(mels[b])
                    _check(case, e.decode_greedy(1, max_new=case.budget), want[b:b + 1], f"{model_type} 1 clip")
                assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_giveups") == 0
            else:
                e.encode_mel(mels[:B])
                _check(case, e.decode_greedy(B, max_new=case.budget), want[:B], f"{model_type} {B} clips")
    finally:
        e.close()
