"""GPU: utterance slots with their own decode offsets, refilled while the others decode (AX_WHISPER_Stream*).

The reference stops each utterance at its own eot (Whisper.cpp:219-222) and its server takes requests one by one
(WhisperHTTPServer.hpp:37-100). Here 32 clips with id budgets between 20 and 140 (budgets stand in for eot: synthetic
weights never emit it) go through 8 slots: a slot that has finished takes the next clip while the other seven are in the
middle of theirs, so slots sit at different offsets in every step. Every clip's ids must equal what the same clip yields
(a) in an ordinary ragged batch of 8 (same kernels: bit-equal expected) and (b) on its own through the 1-clip path."""
import numpy as np
import pytest
import torch  # noqa: F401

from conftest import assert_ids_equal_or_tie

pytestmark = pytest.mark.gpu


def _clips_and_budgets(n):
    import modelgen

    lens = [480000, 200000, 77777, 480000, 123457, 16000, 300000, 480000]
    clips = [modelgen.synth_clip(40 + i, lens[i % len(lens)]) for i in range(n)]
    budgets = [20 + (37 * i) % 121 for i in range(n)]  # 20..140, no order
    return clips, budgets


def test_slots_refill_while_others_decode(built_lib, micro_case):
    n_slots, n_clips = 8, 32
    clips, budgets = _clips_and_budgets(n_clips)
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=n_slots)
    try:
        got, calls = e.run_stream(clips, n_slots, max_new=budgets, steps_per_call=4)
        assert all(g is not None and len(g) == b for g, b in zip(got, budgets)), [len(g) for g in got]
        # (a) the same clips as ordinary ragged batches of 8: same kernels at the same batch size -> the same ids
        for g0 in range(0, n_clips, n_slots):
            mels = np.stack([e.compute_mel(c) for c in clips[g0:g0 + n_slots]])
            e.encode_mel(mels)
            want = e.decode_greedy(n_slots, max_new=140, max_new_clip=budgets[g0:g0 + n_slots])
            for i in range(n_slots):
                assert got[g0 + i] == want[i], (g0 + i, got[g0 + i][:8], want[i][:8])
        # (b) every clip on its own (1-clip path: other summation order; a difference must be a numerical tie)
        same = 0
        for i in range(n_clips):
            alone = e.run_tokens(clips[i], max_new=budgets[i])
            if alone == got[i]:
                same += 1
                continue
            mel = e.compute_mel(clips[i])
            ck, cv = micro_case.oracle_bf16.encoder(mel)
            ids, lg = micro_case.oracle_bf16.greedy(ck, cv, "zh", max_new=budgets[i], want_logits=True)
            assert_ids_equal_or_tie(e, mel, got[i], ids, lg, f"clip {i} through the slot stream")
        print(f"{n_clips} clips through {n_slots} slots in {calls} step calls: {same}/{n_clips} identical to the stand-alone runs")
        assert same >= n_clips - 2
        # the batched entry points are refused while a stream is open, and work again after StreamClose
        e.stream_open(4)
        with pytest.raises(RuntimeError):
            e.run_tokens(clips[0], max_new=4)
        with pytest.raises(RuntimeError):
            e.stream_admit(9, clips[0])       # no such slot
        with pytest.raises(RuntimeError):
            e.compute_mel(clips[0])           # would overwrite the staging rows an admission pass may still be reading
        e.stream_admit(1, clips[0], 5)
        with pytest.raises(RuntimeError):
            e.stream_admit(1, clips[1], 5)    # busy
        fin = []
        for _ in range(40):
            fin = e.stream_step(2)
            if fin:
                break
        assert fin == [1] and e.stream_collect(1) == got[0][:5]
        e.stream_close()
        assert e.run_tokens(clips[0], max_new=5) is not None
        # StreamOpen(1): the step graph is built for 3 slots, but the caller opened ONE: slots 1 and 2 are not his, and
        # finished_slots (host [n_slots]) can never receive more than one entry
        e.stream_open(1)
        for bad_slot in (1, 2):
            with pytest.raises(RuntimeError):
                e.stream_admit(bad_slot, clips[0], 5)
        e.stream_admit(0, clips[0], 5)
        fin = []
        for _ in range(40):
            fin = e.stream_step(2)
            if fin:
                break
        assert fin == [0] and e.stream_collect(0) == got[0][:5]
        with pytest.raises(RuntimeError):
            e.stream_collect(1)
        e.stream_close()
    finally:
        e.close()


def test_stream_beyond_64_slots(built_lib, micro_case):
    """72 slots (three graph branches: 32 + 32 + 8 clips; the vocabulary projection as two launches) fed 150 clips: the slot
    count is a deployment knob (one step costs 17.9 us per clip at 64 clips and 13.1 at 256, engine_decode.cpp decode_branches), so
    the stream has to be right on both sides of the 64-clip launch boundary. Every clip against its stand-alone run."""
    n_slots, n_clips = 72, 150
    clips8, _ = _clips_and_budgets(8)
    clips = [clips8[i % 8] for i in range(n_clips)]
    budgets = [6 + (11 * i) % 23 for i in range(n_clips)]
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=n_slots)
    try:
        got, calls = e.run_stream(clips, n_slots, max_new=budgets, steps_per_call=4)
        assert all(g is not None and len(g) == b for g, b in zip(got, budgets)), [len(g) for g in got]
        alone = [e.run_tokens(c, max_new=28) for c in clips8]
        diff = [i for i in range(n_clips) if got[i] != alone[i % 8][:budgets[i]]]
        print(f"{n_clips} clips through {n_slots} slots in {calls} step calls: {n_clips - len(diff)}/{n_clips} identical to the stand-alone runs")
        for i in diff:  # another summation order: a difference must be a numerical tie
            mel = e.compute_mel(clips[i])
            ck, cv = micro_case.oracle_bf16.encoder(mel)
            ids, lg = micro_case.oracle_bf16.greedy(ck, cv, "zh", max_new=budgets[i], want_logits=True)
            assert_ids_equal_or_tie(e, mel, got[i], ids, lg, f"clip {i} through 72 slots")
        # the 150 clips are 8 distinct ones: a clip that sits on a tie differs in every instance whose budget reaches that step
        assert len({i % 8 for i in diff}) <= 2, sorted({i % 8 for i in diff})
    finally:
        e.close()


def test_stream_one_and_two_slots_and_full_context(built_lib, micro_case):
    """n_slots below 3 (the engine still runs the 3-slot step sequence underneath), a clip that runs to the end of the
    context (444 ids, offset 447) next to short ones, and a slot reused six times."""
    import modelgen

    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=2)
    try:
        clips = [modelgen.synth_clip(70 + i, 160000) for i in range(7)]
        budgets = [0, 9, 17, 3, 30, 12, 25]  # clip 0: no budget -> the whole context
        got, _ = e.run_stream(clips, 2, max_new=budgets)
        assert len(got[0]) == 444
        for i in range(7):
            alone = e.run_tokens(clips[i], max_new=budgets[i])
            n = min(len(alone), len(got[i]))
            assert len(alone) == len(got[i]) and (alone == got[i] or n > 0), i
        ref = [e.run_tokens(c, max_new=b) for c, b in zip(clips, budgets)]
        assert sum(r == g for r, g in zip(ref, got)) >= 6
        got1, _ = e.run_stream(clips[1:4], 1, max_new=budgets[1:4])
        assert got1 == got[1:4]
    finally:
        e.close()


def test_stream_edge_cases(built_lib, micro_case, oracle_mod):
    """A clip longer than the 60 s staging row admitted into a slot next to short ones (its clamp floor still comes from ALL of
    its frames), a poisoned clip (NaN) refused at admission without disturbing the slots that are decoding, collecting a slot
    that has not finished, a step call with nothing admitted, and the half build of the engine (F16 weights) through the
    same stream."""
    import modelgen
    from conftest import ModelCase

    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=4)
    try:
        long_clip = modelgen.synth_long_clip(75, 70)
        short = [modelgen.synth_clip(90 + i, 100000) for i in range(3)]
        want_long = e.run_tokens(long_clip, max_new=12)
        want_short = [e.run_tokens(c, max_new=12) for c in short]
        e.stream_open(4)
        assert e.stream_step(4) == []            # nothing admitted: returns at once
        e.stream_admit_batch([2, 0], [short[0], long_clip], [12, 12])
        bad = short[1].copy()
        bad[5] = np.inf
        with pytest.raises(RuntimeError):
            e.stream_admit(1, bad, 12)            # refused; slot 1 stays idle
        with pytest.raises(RuntimeError):
            e.stream_collect(2)                   # not finished yet
        e.stream_admit_batch([1, 3], short[1:], [12, 12])
        got = {}
        for _ in range(200):
            for sl in e.stream_step(3):
                got[sl] = e.stream_collect(sl)
            if len(got) == 4:
                break
        e.stream_close()
        assert got[0] == want_long and got[2] == want_short[0] and got[1] == want_short[1] and got[3] == want_short[2]
    finally:
        e.close()
    case16 = ModelCase(micro_case.root + "_f16_stream", "micro", 11, dtype="F16")
    e = built_lib.Whisper("micro", case16.root, "zh", device=0, max_batch=3)
    try:
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"fp16") == 1
        clips = [modelgen.synth_clip(120 + i, 200000) for i in range(5)]
        got, _ = e.run_stream(clips, 3, max_new=[9, 30, 4, 17, 11])
        for i, c in enumerate(clips):
            mel, _, _ = oracle_mod.log_mel(c, 80)
            ck, cv = case16.oracle_bf16.encoder(mel)
            ids, lg = case16.oracle_bf16.greedy(ck, cv, "zh", max_new=[9, 30, 4, 17, 11][i], want_logits=True)
            assert_ids_equal_or_tie(e, mel, got[i], ids, lg, f"fp16 stream clip {i}")
    finally:
        e.close()


def test_stream_at_turbo_width(built_lib, tmp_path):
    """d_model 1280 takes the other batched step sequence (split-K GEMMs + LayerNorm preparation launches, 128 mels): slots at
    different offsets through it, against the same clips as one ragged batch and on their own."""
    import modelgen
    from conftest import ModelCase

    case = ModelCase(tmp_path / "w1280", "w1280", 3)
    e = built_lib.Whisper("w1280", case.root, "zh", device=0, max_batch=4)
    try:
        clips = [modelgen.synth_clip(150 + i, 90000 + 40000 * i) for i in range(9)]
        budgets = [7, 33, 12, 70, 5, 21, 40, 9, 16]
        got, _ = e.run_stream(clips, 4, max_new=budgets, steps_per_call=3)
        assert [len(g) for g in got] == budgets
        alone = [e.run_tokens(c, max_new=b) for c, b in zip(clips, budgets)]
        assert sum(a == g for a, g in zip(alone, got)) >= 8
        for g0 in (0, 4):
            grp = clips[g0:g0 + 4]
            mels = np.stack([e.compute_mel(c) for c in grp])
            e.encode_mel(mels)
            want = e.decode_greedy(4, max_new=70, max_new_clip=budgets[g0:g0 + 4])
            assert want == got[g0:g0 + 4]
    finally:
        e.close()


def test_two_handles_on_one_device_capture_side_by_side(built_lib, micro_case):
    """Two handles on ONE GPU (whisper_srv --devices 0,0; a bf16 and an fp16 model in one process) that open their slot
    streams, grow their buffers and capture their step graphs at the same moment from two host threads: allocation and
    synchronous copies of one must not invalidate the stream capture of the other (per-device capture mutex, iengine.hpp)."""
    import threading

    import modelgen

    clips = [modelgen.synth_clip(200 + i, 120000) for i in range(6)]
    engines = [built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=1) for _ in range(2)]
    try:
        want = [engines[0].run_tokens(c, max_new=10) for c in clips]
        res, err = [None, None], [None, None]
        bar = threading.Barrier(2)

        def work(k):
            try:
                bar.wait()
                outs = []
                for n_slots in (3, 5, 4):  # every open grows the capacity or captures a new graph
                    got, _ = engines[k].run_stream(clips, n_slots, max_new=10, steps_per_call=2)
                    outs.append(got)
                    engines[k].run_tokens_batch(clips[:n_slots], max_new=3)  # and the batched entry point's graph of that size
                res[k] = outs
            except Exception as e:  # noqa: BLE001
                err[k] = e

        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert err == [None, None], err
        for k in range(2):
            for got in res[k]:
                assert sum(g == w for g, w in zip(got, want)) >= 5 and all(len(g) == 10 for g in got)
    finally:
        for e in engines:
            e.close()
