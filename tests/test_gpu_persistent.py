"""GPU: the persistent batch-1 decode launch (decode_persistent.hip) against the launch-per-phase path and the oracle.

Both paths implement the same arithmetic (bf16 weights x fp32 activations, fp32 LayerNorm / softmax, bf16 K/V), in a
different summation order. Bars: greedy ids identical between the two paths and equal to the bf16-policy oracle's
(or a numerical tie, as in test_gpu_parity.py); teacher-forced logits of the two paths within 2e-3 abs of each other
and within 2e-2 abs of the bf16-policy oracle."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (imported before libax_whisper.so so both share torch's HIP runtime in this process)

from conftest import ModelCase, load_demo_pcm

pytestmark = pytest.mark.gpu


def _engine(wa, case, mode):
    old = os.environ.get("AX_WHISPER_DECODE")
    if mode:
        os.environ["AX_WHISPER_DECODE"] = mode
    else:
        os.environ.pop("AX_WHISPER_DECODE", None)
    try:
        e = wa.Whisper(case.model_type, case.root, "zh", device=0, max_batch=1)
    finally:
        if old is None:
            os.environ.pop("AX_WHISPER_DECODE", None)
        else:
            os.environ["AX_WHISPER_DECODE"] = old
    return e


@pytest.mark.parametrize("model_type,seed", [("micro", 21), ("mini", 22), ("tiny", 23)])
def test_persistent_equals_graph_path_and_oracle(built_lib, oracle_mod, tmp_path, model_type, seed):
    import modelgen

    case = ModelCase(tmp_path, model_type, seed)
    ep, eg = _engine(built_lib, case, None), _engine(built_lib, case, "graph")
    try:
        assert ep.L.AX_WHISPER_GetConfigInt(ep.h, b"persistent_decode") == 1
        assert eg.L.AX_WHISPER_GetConfigInt(eg.h, b"persistent_decode") == 0
        for pcm, max_new in ((load_demo_pcm(), 40), (modelgen.synth_clip(3, 160000), 0)):
            mel, _, _ = oracle_mod.log_mel(pcm, 80)
            ck, cv = case.oracle_bf16.encoder(mel)
            ref_ids, ref_lg = case.oracle_bf16.greedy(ck, cv, "zh", max_new=max_new or 24, want_logits=True)
            if max_new == 0:  # full context through both engine paths (444 ids with synthetic weights)
                ids_p, ids_g = ep.run_tokens(pcm), eg.run_tokens(pcm)
                assert ids_p == ids_g and len(ids_p) > 400  # oracle parity of full-context runs: test_gpu_fp32_ids.py
                continue
            ids_p, ids_g = ep.run_tokens(pcm, max_new=max_new), eg.run_tokens(pcm, max_new=max_new)
            assert ids_p == ids_g
            ep.encode_mel(mel[None])
            eg.encode_mel(mel[None])
            lp, ap = ep.decode_forced(1, np.array([ref_ids]))
            lgp, ag = eg.decode_forced(1, np.array([ref_ids]))
            n = len(ref_ids) + 1
            e_paths = float(np.abs(lp[0, :n] - lgp[0, :n]).max())
            e_ref = float(np.abs(lp[0, :n] - ref_lg).max())
            print(model_type, "logits: persistent vs graph", e_paths, "persistent vs oracle", e_ref)
            assert e_paths < 5e-4 and e_ref < 4e-3  # measured 4.3e-5 .. 5.6e-5 and 3.1e-4 .. 7.4e-4 (a margin for other compiler / driver versions)
            assert np.array_equal(ap, ag)
            top2 = np.sort(ref_lg, axis=1)[:, -2:]
            margin = float((top2[:, 1] - top2[:, 0]).min())
            assert ids_p == ref_ids or margin < 2 * e_ref
    finally:
        ep.close()
        eg.close()


def test_persistent_repeated_runs_are_deterministic(built_lib, micro_case):
    """State (granule tags, LDS caches, error word) is re-initialised per launch: back-to-back utterances of different
    lengths through one handle give the same ids as fresh handles."""
    import modelgen

    clips = [modelgen.synth_clip(i, n) for i, n in ((0, 48000), (1, 480000), (2, 16000))]
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=1)
    try:
        first = [e.run_tokens(c, max_new=30) for c in clips]
        again = [e.run_tokens(c, max_new=30) for c in reversed(clips)][::-1]
        assert first == again
    finally:
        e.close()


def test_persistent_gives_up_and_falls_back(built_lib, micro_case, monkeypatch):
    """Every spin in the persistent launch is bounded: with a workgroup that never publishes (test hook) the launch
    drains within seconds, the handle switches to the launch-per-phase decoder and still returns the right ids."""
    import time

    import modelgen

    clip = modelgen.synth_clip(4, 64000)
    ref = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=1)
    try:
        want = ref.run_tokens(clip, max_new=20)
    finally:
        ref.close()
    monkeypatch.setenv("AX_WHISPER_PERSIST_FAULT", "1")
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=1)
    try:
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_decode") == 1
        t0 = time.time()
        got = e.run_tokens(clip, max_new=20)
        dt = time.time() - t0
        assert got == want
        g = lambda k: e.L.AX_WHISPER_GetConfigInt(e.h, k)
        assert g(b"persistent_decode") == 0 and g(b"persistent_giveups") == 1
        assert dt < 2.0, dt  # the spin limit is ~50 ms of wall clock, then one fallback decode
        # the fast path is re-armed after 8 requests on the fallback path (it used to be lost for good)
        monkeypatch.delenv("AX_WHISPER_PERSIST_FAULT")
        for i in range(8):
            assert e.run_tokens(clip, max_new=20) == want
            assert g(b"persistent_decode") == (0 if i < 7 else 1)
        assert e.run_tokens(clip, max_new=20) == want  # through the persistent launch again
        assert g(b"persistent_decode") == 1 and g(b"persistent_giveups") == 1
        # a second give-up backs off longer (32 requests)
        monkeypatch.setenv("AX_WHISPER_PERSIST_FAULT", "1")
        assert e.run_tokens(clip, max_new=20) == want
        assert g(b"persistent_giveups") == 2 and g(b"persistent_decode") == 0
        monkeypatch.delenv("AX_WHISPER_PERSIST_FAULT")
        assert e.run_tokens(clip, max_new=20) == want
    finally:
        e.close()



def _need_clips(e, n):
    """The multi-clip launches need grid - L*H >= 3*n*H workgroups without a head: all 256 CUs of an MI355X for Whisper-small.
    On a smaller part or a CU-masked queue the engine decodes such groups another way: skip, do not fail."""
    have = e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_max_clips")
    if have < n:
        e.close()
        pytest.skip(f"this device runs {have} clip(s) per persistent launch, the test needs {n}")


def test_two_clips_run_one_persistent_launch(built_lib, micro_case):
    """Two clips per call: ONE two-clip persistent launch (decode_persistent2.hip; round 4 — it was one launch per clip);
    ids equal the one-clip runs, per-clip budgets are honoured, and every clip reads ITS slot's cross K/V (seeded
    weights barely listen to ordinary audio, so the second clip's features are a constant far outside the normal
    range — the one input found to change the ids)."""
    import modelgen
    from make_model_goldens_inputs import demo_mel

    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=2)
    try:
        _need_clips(e, 2)
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_two_clips") == 1
        mels = [demo_mel(80), np.full((80, 3000), 5.0, dtype=np.float32)]
        single = []
        for m in mels:
            e.encode_mel(m)
            single.append(e.decode_greedy(1, max_new=20)[0])
        assert single[0] != single[1]
        for order in ((0, 1), (1, 0)):
            e.encode_mel(np.stack([mels[i] for i in order]))
            assert e.decode_greedy(2, max_new=20) == [single[i] for i in order]
        e.encode_mel(np.stack(mels))
        assert e.decode_greedy(2, max_new=20, max_new_clip=[5, 13]) == [single[0][:5], single[1][:13]]
        assert e.decode_greedy(2, max_new=20, max_new_clip=[13, 5]) == [single[0][:13], single[1][:5]]
        assert e.decode_greedy(2, max_new=7, max_new_clip=[0, 400]) == [single[0][:7], single[1][:7]]
        clips = [load_demo_pcm(), modelgen.synth_clip(9, 123456)]
        assert e.run_tokens_batch(clips, max_new=12) == [e.run_tokens(c, max_new=12) for c in clips]
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_giveups") == 0
    finally:
        e.close()


def test_two_clip_launch_gives_up_and_falls_back(built_lib, micro_case, monkeypatch):
    """A workgroup of the two-clip launch that never publishes (test hook): every other workgroup times out and drains, the
    call falls back to the launch-per-phase path and returns the same ids."""
    import modelgen

    clips = [load_demo_pcm(), modelgen.synth_clip(9, 123456)]
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=2)
    try:
        want = e.run_tokens_batch(clips, max_new=16)
        monkeypatch.setenv("AX_WHISPER_PERSIST_FAULT", "1")
        assert e.run_tokens_batch(clips, max_new=16) == want
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_giveups") == 1
        monkeypatch.delenv("AX_WHISPER_PERSIST_FAULT")
        assert e.run_tokens_batch(clips, max_new=16) == want
    finally:
        e.close()


@pytest.mark.parametrize("model_type,seed,max_new", [("mini", 31, 40), ("tiny", 32, 60), ("w512", 33, 40), ("small", 34, 100)])
def test_two_clip_launch_equals_one_clip_launches(built_lib, tmp_path, monkeypatch, model_type, seed, max_new):
    """The two-clip launch runs the one-clip launch's arithmetic per clip (same rows, same summation order): ids of a
    pair EQUAL the ids of its clips decoded alone — every template instantiation (d_model 256 / 384 / 512 / 768), long
    enough for clip 1's self-attention cache (global memory) to cross a 64-key block, and again with the pair swapped.
    AX_WHISPER_PERSIST2=0 keeps the round-3 behaviour (one launch per clip)."""
    import modelgen

    case = ModelCase(tmp_path, model_type, seed)
    clips = [load_demo_pcm(), modelgen.synth_clip(seed, 200000), modelgen.synth_clip(seed + 1, 90000)]
    e = built_lib.Whisper(model_type, case.root, "zh", device=0, max_batch=2)
    try:
        _need_clips(e, 2)
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_two_clips") == 1
        mels = np.stack([e.compute_mel(c) for c in clips])
        single = []
        for m in mels:
            e.encode_mel(m)
            single.append(e.decode_greedy(1, max_new=max_new)[0])
        for pair in ((0, 1), (1, 2), (2, 0), (1, 1)):
            e.encode_mel(np.stack([mels[i] for i in pair]))
            for _ in range(2):   # a second run starts from a used self-attention cache
                assert e.decode_greedy(2, max_new=max_new) == [single[i] for i in pair], pair
        cut = [max_new // 3, max_new - 1]
        assert e.decode_greedy(2, max_new=max_new, max_new_clip=cut) == [single[1][:cut[0]], single[1][:cut[1]]]
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_giveups") == 0
    finally:
        e.close()
    monkeypatch.setenv("AX_WHISPER_PERSIST2", "0")
    e = built_lib.Whisper(model_type, case.root, "zh", device=0, max_batch=2)
    try:
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_two_clips") == 0
        e.encode_mel(mels[:2])
        assert e.decode_greedy(2, max_new=max_new) == single[:2]
    finally:
        e.close()


@pytest.mark.parametrize("model_type,seed,max_new,dtype", [("micro", 41, 30, "BF16"), ("tiny", 42, 70, "BF16"), ("small", 43, 100, "BF16"),
                                                        ("tiny", 44, 70, "F16"), ("small", 45, 80, "F16")])
def test_three_clip_launch_equals_one_clip_launches(built_lib, tmp_path, monkeypatch, model_type, seed, max_new, dtype):
    """Three clips per call: ONE three-clip persistent launch (3 x 36 one-clip cross-attention units per layer on the 112
    workgroups without a head; the caches of clips 1 and 2 in global memory, their blocks run by the poller waves one clip
    after the other). ids of a triple EQUAL the ids of its clips decoded alone, in every order, with per-clip budgets;
    AX_WHISPER_PERSIST2=2 sends three clips through the clip-block sequence as before."""
    import itertools

    import modelgen

    case = ModelCase(tmp_path, model_type, seed, dtype=dtype)   # both builds of the kernel: bfloat16 and IEEE half
    clips = [load_demo_pcm(), modelgen.synth_clip(seed, 200000), modelgen.synth_clip(seed + 1, 90000)]
    e = built_lib.Whisper(model_type, case.root, "zh", device=0, max_batch=3)
    try:
        _need_clips(e, 3)
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_max_clips") == 3
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"fp16") == (1 if dtype == "F16" else 0)
        mels = np.stack([e.compute_mel(c) for c in clips])
        single = []
        for m in mels:
            e.encode_mel(m)
            single.append(e.decode_greedy(1, max_new=max_new)[0])
        for order in list(itertools.permutations(range(3)))[:4] + [(1, 1, 2)]:
            e.encode_mel(np.stack([mels[i] for i in order]))
            for _ in range(2):
                assert e.decode_greedy(3, max_new=max_new) == [single[i] for i in order], order
        cut = [max_new // 3, max_new - 1, 2]
        e.encode_mel(mels)
        assert e.decode_greedy(3, max_new=max_new, max_new_clip=cut) == [single[i][:cut[i]] for i in range(3)]
        assert e.run_tokens_batch(clips, max_new=12) == [s[:12] for s in single]
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_giveups") == 0
    finally:
        e.close()
    monkeypatch.setenv("AX_WHISPER_PERSIST2", "2")
    e = built_lib.Whisper(model_type, case.root, "zh", device=0, max_batch=3)
    try:
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_max_clips") == 2
        e.encode_mel(mels)
        got = e.decode_greedy(3, max_new=max_new)   # clip-block sequence: same ids up to numerical ties
        assert [len(g) for g in got] == [len(s) for s in single]
    finally:
        e.close()


@pytest.mark.parametrize("model_type,seed,kind", [("micro", 11, "benign"), ("mini", 42, "realistic"), ("small", 45, "realistic")])
def test_query_fold_equals_the_unfolded_launch(built_lib, oracle_mod, tmp_path, monkeypatch, model_type, seed, kind):
    """Round 5: the persistent launches fold the cross-attention query through the output projection and the LayerNorm into
    the QKV / mlp.0 rows (decode_persistent.hip). The folded and the unfolded launch (AX_WHISPER_QFOLD=0) evaluate the same
    arithmetic in another association: teacher-forced logits within 2e-5 (benign) / 4e-4 (trained-model statistics) of the
    logit scale, greedy ids equal (or a tie at
    the first difference, measured), for one, two and three clips — on benign and on trained-model-like weights."""
    import modelgen

    case = ModelCase(tmp_path, model_type, seed, kind=kind)
    clips = [load_demo_pcm(), modelgen.synth_clip(seed, 200000), modelgen.synth_clip(seed + 1, 90000)]
    n_new = 40 if model_type != "small" else 24
    res = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("AX_WHISPER_QFOLD", fold)
        e = built_lib.Whisper(model_type, case.root, "zh", device=0, max_batch=3)
        try:
            assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_qfold") == int(fold)
            mels = np.stack([e.compute_mel(c) for c in clips])
            e.encode_mel(mels[0])
            ids1 = e.decode_greedy(1, max_new=n_new)[0]
            forced = np.array([ids1], dtype=np.int32) if fold == "1" else np.array([res["1"][0]], dtype=np.int32)
            e.encode_mel(mels[0])
            logits, _ = e.decode_forced(1, forced)
            groups = []
            for B in (2, 3):
                if B > e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_max_clips"):
                    continue
                e.encode_mel(mels[:B])
                groups.append(e.decode_greedy(B, max_new=n_new))
            assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_giveups") == 0
            # every clip of the largest group teacher-forced along the FOLDED launch's ids of that clip (one-clip path): both
            # arithmetics on one context, so a differing group id can be judged against that clip's own margins and error
            ref_groups = groups if fold == "1" else res["1"][2]
            clip_logits = []
            if ref_groups:
                for b, ids_b in enumerate(ref_groups[-1]):
                    e.encode_mel(mels[b])
                    lg_b, _ = e.decode_forced(1, np.array([ids_b], dtype=np.int32))
                    clip_logits.append(lg_b[0])
            res[fold] = (ids1, logits[0], groups, clip_logits)
        finally:
            e.close()
    (ids_f, lg_f, gr_f, cl_f), (ids_u, lg_u, gr_u, cl_u) = res["1"], res["0"]
    scale = float(np.abs(lg_u).max())
    err = float(np.abs(lg_f - lg_u).max())
    print(f"{model_type} ({kind}): folded vs unfolded logits differ by {err:.3e} at |logit| <= {scale:.1f}")
    # benign weights: fp32 association noise. Trained-model statistics: a last-bit fp32 difference flips the 16-bit rounding of a
    # stored self-attention K/V entry here and there (2^-9 relative), which the outlier channels amplify — the same size as the
    # difference between any two decode paths of the battery (tests/test_gpu_realistic.py: 6.5e-2 / 8.2e-2 vs the oracle at
    # these dims); measured 3.3e-2 at |logit| <= 189 (small), 3.4e-4 at 328 (mini)
    rel = 2e-5 if kind == "benign" else 4e-4
    assert err < rel * max(scale, 1.0) + 2e-5, (err, scale)

    def same_or_tie(a, b, lg, e_max, ctx=None):
        if a == b:
            return True
        i = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), None)
        if ctx is not None:  # lg was forced along ctx: where `a` left ctx first is the step the margins can speak for
            j = next((j for j in range(min(len(a), len(ctx))) if a[j] != ctx[j]), None)
            if j is not None and (i is None or j < i):
                i = j
        if i is None or i >= len(lg):  # one is a strict prefix of the other: never a tie
            return False
        srt = np.sort(lg[i])
        return srt[-1] - srt[-2] < 2 * e_max + 1e-4

    assert same_or_tie(ids_f, ids_u, lg_u, err)
    for gf, gu in zip(gr_f, gr_u):
        # group launches: the same association difference. EVERY clip equal, or a measured tie at its first difference: that clip's
        # own unfolded margins (forced along the folded ids, so the context up to the difference is common) against the error
        # between the two arithmetics measured on that clip
        for b in range(len(gf)):
            err_b = float(np.abs(cl_f[b] - cl_u[b]).max())
            assert err_b < rel * max(scale, 1.0) + 2e-5, (b, err_b, scale)
            assert same_or_tie(gf[b], gu[b], cl_u[b], err_b, ctx=gr_f[-1][b]), (len(gf), b, gf[b], gu[b], err_b)
