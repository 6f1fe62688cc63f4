"""GPU: the parity battery under TRAINED-MODEL ACTIVATION STATISTICS (round 5).

Every other model test runs N(0, 0.02) weights: logit std 0.2-0.5, no outlier channels, near-uniform attention. Those
weights let a 4e-2 logit defect of one decode path pass for two rounds (DESIGN "Round 3 (c)"). The reference's real
checkpoints do not exist here, so `modelgen.realistic_weights` builds seeded models whose activations look like a trained
Whisper's (tests/golden/probe_realistic_stats.py prints them; the CPU oracle is pinned under the same statistics against
transformers: tests/test_oracle_model.py real_*):

    residual outlier channels |x| 150-500 | LayerNorm gains 0.02-30 | attention heads from flat (1/1500 per key) to
    saturated (one key ~1.0) | FFN hidden values ~5000 | logit std ~10, |logit| up to ~130 | vocabulary rows with exact
    and near duplicates: top-2 margins from 0 (exact tie: the LOWER id must win, Whisper.cpp:42-45) over 1e-3 to ~10

Checked on every decode path, in the bf16 and the fp16 build:
  * cross K/V, teacher-forced logits and argmax against the oracle that narrows at the engine's 16-bit storage points
    (tight) AND against the PURE-fp32 oracle (the independent bar: a wrong storage point would show here);
  * greedy ids against both oracles (a difference only where the oracle's own top-2 margin is below twice the logit error
    measured at that step, export_onnx.py:103-150 fp32 softmax / -60000 fill, Whisper.cpp:42-45);
  * AX_WHISPER_ScanStored16: no NaN / Inf in any 16-bit tensor the engine stores, and the FFN peak really is there.
Tolerances are RELATIVE to the logit scale (std ~10): stated per (model, dtype) in TOL, ~5x what was measured on MI355X.
"""
import numpy as np
import pytest
import torch  # noqa: F401

from conftest import ModelCase, assert_ids_equal_or_tie

pytestmark = pytest.mark.gpu

# (model, dtype) -> (logits vs policy oracle, logits vs fp32 oracle, cross K/V vs policy oracle, vs fp32 oracle), abs
TOL = {   # measured on MI355X (profiles/r05_realistic_battery.txt): the bound is ~5x the worst path of that model
    ("micro", "BF16"): (4e-2, 0.2, 0.13, 0.13),     # 7.4e-3 | 4.4e-2 | 6.25e-2 = one bf16 ulp at |x| 8..16 | 4.8e-2
    ("micro", "F16"): (3e-3, 3e-2, 1.6e-2, 1.6e-2),   # 5.4e-4 | 6.6e-3 | 7.8e-3 = one half ulp at 8..16 | 5.9e-3
    ("mini", "BF16"): (5e-2, 0.2, 0.13, 0.13),       # 9.6e-3 | 3.7e-2
    ("mini", "F16"): (6e-3, 3e-2, 1.6e-2, 1.6e-2),    # 1.3e-3 | 6.3e-3
    ("w512", "BF16"): (1.2e-2, 0.12, 0.13, 0.13),    # 2.2e-3 | 2.4e-2
    ("tiny", "BF16"): (3.5e-2, 0.15, 0.13, 0.13),    # 6.9e-3 | 3.0e-2 (Whisper-tiny dims: logit std 7.8, |logit| up to 321)
    ("tiny", "F16"): (3.5e-3, 1.7e-2, 1.6e-2, 1.6e-2),  # 6.3e-4 | 3.4e-3
    ("w1280", "F16"): (2e-3, 1.3e-2, 1.6e-2, 1.6e-2),  # 3.3e-4 | 2.6e-3
    ("small", "BF16"): (0.4, 0.75, 0.13, 0.13),      # 8.2e-2 | 0.15 (logit std 13.4, |logit| up to 160: 0.6 % / 1.1 % of the std)
    ("small", "F16"): (4e-2, 0.1, 1.6e-2, 1.6e-2),    # 7.6e-3 | 1.9e-2
    ("turbo", "F16"): (6e-3, 2.5e-2, 1.6e-2, 1.6e-2),  # 1.1e-3 | 4.8e-3 (logit std 13.2, |logit| up to 709)
}


def _clips(n):
    from eot_case import eot_clips

    return eot_clips(n)


class Refs:
    """Oracle results per clip, computed once per (case, clip): mel, cross K/V, greedy ids + logits under both oracles, and
    teacher-forced logits over a MIXED context (the policy oracle's own ids, then seeded random ids: contexts a greedy
    run of an untrained model never visits, so argmax and margins move from step to step)."""

    def __init__(self, case, oracle_mod, n_new, n_rand):
        self.case, self.o, self.n_new, self.n_rand, self.c = case, oracle_mod, n_new, n_rand, {}

    def get(self, i, pcm):
        if i in self.c:
            return self.c[i]
        case = self.case
        mel, _, _ = self.o.log_mel(pcm, case.dims["n_mels"])
        r = {"mel": mel}
        for tag, orc in (("pol", case.oracle_bf16), ("f32", case.oracle_fp32)):
            ck, cv = orc.encoder(mel)
            ids, lg = orc.greedy(ck, cv, "zh", max_new=self.n_new, want_logits=True)
            r[tag] = dict(ck=ck, cv=cv, ids=ids, lg=lg)
        rng = np.random.Generator(np.random.PCG64(900 + i))
        forced = list(r["pol"]["ids"][: self.n_new // 2]) + [int(x) for x in rng.integers(0, 50257, self.n_rand)]
        r["forced"] = forced
        for tag, orc in (("pol", case.oracle_bf16), ("f32", case.oracle_fp32)):
            _, r[tag]["flg"] = orc.greedy(r[tag]["ck"], r[tag]["cv"], "zh", max_new=len(forced), forced=forced, want_logits=True)
        self.c[i] = r
        return r


def _margins(lg):
    srt = np.sort(lg, axis=1)
    return srt[:, -1] - srt[:, -2]


def _check_forced(e, refs, clips, batch, tol, what, report):
    """Teacher-forced logits + per-step argmax of `batch` clips (slots 0..batch-1 already encoded) against both oracles."""
    rs = [refs.get(b % len(clips), clips[b % len(clips)]) for b in range(batch)]
    n = min(len(r["forced"]) for r in rs)
    forced = np.array([r["forced"][:n] for r in rs], dtype=np.int32)
    logits, am = e.decode_forced(batch, forced)
    worst_p = worst_f = 0.0
    ties = exact = 0
    for b, r in enumerate(rs):
        for tag in ("pol", "f32"):
            lg = r[tag]["flg"][: n + 1]
            err = np.abs(logits[b] - lg).max(axis=1)
            if tag == "pol":
                worst_p = max(worst_p, float(err.max()))
            else:
                worst_f = max(worst_f, float(err.max()))
            mg = _margins(lg)
            for s in range(n + 1):
                want = int(lg[s].argmax())  # numpy: first maximum = the lower id of an exact tie (Whisper.cpp:42-45)
                if am[b, s] != want:
                    assert mg[s] < 2 * err[s] + 1e-4, (what, tag, "clip", b, "step", s, "margin", float(mg[s]), "err", float(err[s]))
                    ties += tag == "pol"
                elif mg[s] == 0.0 and tag == "pol":
                    exact += 1
    report.append(f"{what}: forced logits err vs policy oracle {worst_p:.3e}, vs fp32 oracle {worst_f:.3e}; argmax ties accepted {ties}, exact-tie steps won by the lower id {exact}")
    assert worst_p < tol[0], (what, worst_p)
    assert worst_f < tol[1], (what, worst_f)
    return worst_p, worst_f


def _check_greedy(e, refs, clips, got, what, report, batch_mels=None):
    agree = total = 0
    for b, g in enumerate(got):
        r = refs.get(b % len(clips), clips[b % len(clips)])
        for tag in ("pol", "f32"):
            ids, lg = r[tag]["ids"], r[tag]["lg"]
            k = assert_ids_equal_or_tie(e, r["mel"], g, ids, lg, f"{what} clip {b} vs {tag}", batch_mels=batch_mels, slot=b)
            if tag == "pol":
                agree += k
                total += len(ids)
    report.append(f"{what}: greedy ids equal to the policy oracle up to the first accepted tie: {agree}/{total}")


def _scan_ok(e, batch, dtype, report, what, want_ffn_peak=None):
    sc = e.scan_stored16(batch)
    bad = {k: v for k, v in sc.items() if v[0]}
    assert not bad, (what, bad)
    lim = 65504.0 if dtype == "F16" else 3.0e38
    assert all(v[1] <= lim for v in sc.values())
    if want_ffn_peak:
        assert sc["enc.ffn_hidden"][1] > want_ffn_peak, sc["enc.ffn_hidden"]
    report.append(f"{what}: stored 16-bit tensors finite; max|x| " + ", ".join(f"{k} {v[1]:.4g}" for k, v in sc.items() if v[1] > 0))


def _logit_stats(refs, clips, report, what):
    r = refs.get(0, clips[0])
    lg = r["f32"]["flg"]
    mg = _margins(lg)
    report.append(f"{what}: fp32 oracle logits std {lg.std():.2f}, max |logit| {np.abs(lg).max():.1f}; top-2 margins over {len(mg)} forced steps: "
                  f"min {mg.min():.2e}, median {np.median(mg):.2e}, max {mg.max():.2e}; exact ties {int((mg == 0).sum())}")
    assert lg.std() > 5.0


PATHS = [("persistent", 1), ("graph", 1), ("graph", 2), ("persistent", 2), ("persistent", 3), ("gemv", 3), ("cblock", 3), ("cblock", 7),
         ("cblock", 20), ("splitk", 6), ("stream", 5)]


@pytest.fixture(scope="module")
def report():
    lines = []
    yield lines
    print("\n==== realistic-statistics battery, measured ====")
    for ln in lines:
        print(ln)


@pytest.fixture(scope="module", params=[("micro", "BF16", 41), ("micro", "F16", 41), ("mini", "BF16", 42), ("mini", "F16", 42)],
                ids=lambda p: f"{p[0]}-{p[1]}")
def real_case(request, tmp_path_factory, oracle_mod):
    mt, dtype, seed = request.param
    case = ModelCase(tmp_path_factory.mktemp(f"real_{mt}_{dtype}"), mt, seed, dtype=dtype, kind="realistic")
    case.refs = Refs(case, oracle_mod, n_new=16, n_rand=10)
    case.clips = _clips(4)
    return case


@pytest.mark.parametrize("mode,batch", PATHS)
def test_realistic_statistics_every_decode_path(built_lib, real_case, monkeypatch, report, mode, batch):
    case = real_case
    tol = TOL[(case.model_type, case.dtype)]
    what = f"{case.model_type} {case.dtype} {mode} B={batch}"
    if mode == "graph":
        monkeypatch.setenv("AX_WHISPER_DECODE", "graph")
    if mode == "gemv":
        monkeypatch.setenv("AX_WHISPER_GEMV_MAX", "4")
    if mode == "splitk":
        monkeypatch.setenv("AX_WHISPER_BATCHED_LN", "0")
    if mode == "cblock" and batch == 3:
        monkeypatch.setenv("AX_WHISPER_PERSIST2", "2")
    clips, refs = case.clips, case.refs
    e = built_lib.Whisper(case.model_type, case.root, "zh", device=0, max_batch=max(batch, 3))
    try:
        g = lambda k: e.L.AX_WHISPER_GetConfigInt(e.h, k.encode())
        assert g("fp16") == (1 if case.dtype == "F16" else 0)
        if mode == "persistent" and batch > g("persistent_max_clips"):
            pytest.skip("this device / shape has no multi-clip persistent launch")
        if mode == "stream":
            got, _ = e.run_stream([clips[b % len(clips)] for b in range(9)], batch, max_new=16, steps_per_call=3)
            _check_greedy(e, refs, clips, got, what, report)
            return
        mels = np.stack([refs.get(b % len(clips), clips[b % len(clips)])["mel"] for b in range(batch)])
        e.encode_mel(mels)
        if mode in ("persistent", "graph") and batch == 1:
            r = refs.get(0, clips[0])
            k, v = e.get_cross_kv(0)
            ekp, evp = float(np.abs(k - r["pol"]["ck"]).max()), float(np.abs(v - r["pol"]["cv"]).max())
            ekf, evf = float(np.abs(k - r["f32"]["ck"]).max()), float(np.abs(v - r["f32"]["cv"]).max())
            report.append(f"{what}: cross K/V err vs policy oracle {ekp:.3e}/{evp:.3e}, vs fp32 oracle {ekf:.3e}/{evf:.3e} (max |K| {np.abs(r['f32']['ck']).max():.1f})")
            assert max(ekp, evp) < tol[2] and max(ekf, evf) < tol[3]
            _logit_stats(refs, clips, report, what)
        got = e.decode_greedy(batch, max_new=16)
        if mode == "persistent":
            assert g("persistent_giveups") == 0
        _check_greedy(e, refs, clips, got, what, report, batch_mels=None if batch <= 2 else mels)
        _scan_ok(e, batch, case.dtype, report, what, want_ffn_peak=1000.0)
        if not (mode == "persistent" and batch > 1):  # the multi-clip launch has no teacher-forced form: ids only
            e.encode_mel(mels)
            _check_forced(e, refs, clips, batch, tol, what, report)
    finally:
        e.close()


@pytest.mark.parametrize("model_type,dtype,seed,batches", [("w512", "BF16", 46, (1, 5)), ("w1280", "F16", 47, (1, 5, 18)),
                                                           ("tiny", "BF16", 48, (1, 2, 3, 5)), ("tiny", "F16", 48, (1, 3, 5))])
def test_realistic_statistics_other_widths(built_lib, oracle_mod, tmp_path, report, model_type, dtype, seed, batches):
    """d_model 512 (its own persistent-launch instantiation), 1280 (the split-K sequence of d_model > 1024, fp16) and Whisper-tiny's
    dims (BASELINE configs[0]: d 384, 4 + 4 layers — the one-, two- and three-clip persistent launches of that width)."""
    case = ModelCase(tmp_path, model_type, seed, dtype=dtype, kind="realistic")
    refs = Refs(case, oracle_mod, n_new=12, n_rand=8)
    clips = _clips(3)
    tol = TOL[(model_type, dtype)]
    e = built_lib.Whisper(model_type, case.root, "zh", device=0, max_batch=max(batches))
    try:
        for B in batches:
            what = f"{model_type} {dtype} B={B}"
            mels = np.stack([refs.get(b % 3, clips[b % 3])["mel"] for b in range(B)])
            e.encode_mel(mels)
            got = e.decode_greedy(B, max_new=12)
            _check_greedy(e, refs, clips, got, what, report, batch_mels=None if B <= 2 else mels)
            _scan_ok(e, B, dtype, report, what)
            e.encode_mel(mels)
            _check_forced(e, refs, clips, B, tol, what, report)
        _logit_stats(refs, clips, report, f"{model_type} {dtype}")
    finally:
        e.close()


@pytest.mark.parametrize("dtype", ["BF16", "F16"])
def test_realistic_statistics_whisper_small_dims(built_lib, oracle_mod, tmp_path, report, dtype):
    """BASELINE configs[1]/[2] dims (d 768, 12 + 12 layers): one clip through the persistent launch, 2 and 3 clips through the
    multi-clip launch, 5 through the clip-block sequence; logits and ids vs both oracles."""
    case = ModelCase(tmp_path, "small", 45, dtype=dtype, kind="realistic")
    refs = Refs(case, oracle_mod, n_new=10, n_rand=6)
    clips = _clips(2)
    tol = TOL[("small", dtype)]
    e = built_lib.Whisper("small", case.root, "zh", device=0, max_batch=40 if dtype == "BF16" else 5)
    try:
        for B in (1, 2, 3, 5, 40):   # 40: three graph branches of whole clip blocks (16 + 16 + 8), two distinct clips repeated
            what = f"small {dtype} B={B}"
            if B == 40 and dtype == "F16":
                continue   # (one build is enough for the branch plumbing: the kernels are the 5-clip run's)
            if B in (2, 3) and B > e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_max_clips"):
                continue
            mels = np.stack([refs.get(b % 2, clips[b % 2])["mel"] for b in range(B)])
            e.encode_mel(mels)
            if B == 1:
                r = refs.get(0, clips[0])
                k, v = e.get_cross_kv(0)
                ekp, ekf = float(np.abs(k - r["pol"]["ck"]).max()), float(np.abs(k - r["f32"]["ck"]).max())
                evp, evf = float(np.abs(v - r["pol"]["cv"]).max()), float(np.abs(v - r["f32"]["cv"]).max())
                report.append(f"{what}: cross K/V err vs policy oracle {ekp:.3e}/{evp:.3e}, vs fp32 oracle {ekf:.3e}/{evf:.3e}")
                assert max(ekp, evp) < tol[2] and max(ekf, evf) < tol[3]
            got = e.decode_greedy(B, max_new=10)
            _check_greedy(e, refs, clips, got, what, report, batch_mels=None if B <= 3 else mels)
            _scan_ok(e, B, dtype, report, what)
            if B in (1, 5, 40):
                e.encode_mel(mels)
                _check_forced(e, refs, clips, B, tol, what, report)
        _logit_stats(refs, clips, report, f"small {dtype}")
    finally:
        e.close()


def test_realistic_statistics_turbo_dims_fp16(built_lib, oracle_mod, tmp_path, report):
    """BASELINE configs[3] dims (large-v3-turbo: d 1280, 32 encoder / 4 decoder layers, 128 mels) in the fp16 build: the
    one-clip persistent launch at d = 1280 and the d_model > 1024 sequence at 4 clips."""
    case = ModelCase(tmp_path, "turbo", 48, dtype="F16", kind="realistic")
    refs = Refs(case, oracle_mod, n_new=8, n_rand=4)
    clips = _clips(1)
    tol = TOL[("turbo", "F16")]
    e = built_lib.Whisper("turbo", case.root, "zh", device=0, max_batch=4)
    try:
        for B in (1, 4):
            what = f"turbo F16 B={B}"
            mels = np.stack([refs.get(0, clips[0])["mel"]] * B)
            e.encode_mel(mels)
            got = e.decode_greedy(B, max_new=8)
            _check_greedy(e, refs, clips, got, what, report, batch_mels=None if B <= 2 else mels)
            _scan_ok(e, B, "F16", report, what)
            e.encode_mel(mels)
            _check_forced(e, refs, clips, B, tol, what, report)
        _logit_stats(refs, clips, report, "turbo F16")
    finally:
        e.close()


def test_natural_eot_under_realistic_statistics(built_lib, oracle_mod, tmp_path, report):
    """The loop ends on eot (Whisper.cpp:219-222) with logits of std ~10 around it: tests/eot_case.py's stop signal on top of
    realistic weights. Stops must be the oracle's; ids equal or a measured tie (the vocabulary holds near-duplicate rows)."""
    from eot_case import EotCase

    case = EotCase("micro", 41, base="realistic", gains=(1, 2, 4, 8, 16, 32, 64), n_cal=8, ramp=64.0, budget=48)
    root = case.write(tmp_path / "m")
    _, clips, want = case.select(8, min_margin=0.05)
    stops = [len(x) for x in want]
    report.append(f"natural eot, realistic micro: gain {case.g} stops {stops}")
    assert len(set(stops)) >= 2 and max(stops) < case.budget
    for mode, B in (("persistent", 1), ("persistent", 3), ("cblock", 8)):
        e = built_lib.Whisper("micro", root, "zh", device=0, max_batch=8)
        try:
            mels = np.stack([e.compute_mel(c) for c in clips[:B]])
            e.encode_mel(mels)
            got = e.decode_greedy(B, max_new=case.budget)
            n_eq = 0
            for b in range(B):
                assert case.eot not in got[b]
                if got[b] != want[b]:  # only as a measured tie at the first diverging step (near-duplicate rows)
                    mel, _, _ = oracle_mod.log_mel(clips[b], 80)
                    ck, cv = case.oracle.encoder(mel)
                    ids, lg = case.oracle.greedy(ck, cv, "zh", max_new=case.budget, want_logits=True)
                    assert ids == want[b]
                    assert_ids_equal_or_tie(e, mel, got[b], ids, lg, f"natural eot {mode} clip {b}", batch_mels=None if B <= 2 else mels, slot=b)
                n_eq += got[b] == want[b]
            report.append(f"natural eot {mode} B={B}: {n_eq}/{B} clips id-for-id equal to the oracle, all stops within the tie rule")
        finally:
            e.close()


@pytest.mark.parametrize("dtype", ["BF16", "F16"])
def test_realistic_statistics_full_context(built_lib, oracle_mod, tmp_path, report, dtype):
    """The whole 448-step context under the realistic statistics (self-attention over all 7 key blocks with saturated heads,
    positions up to 447 of the outlier-carrying positional embedding): teacher-forced with the policy oracle's own 444 ids, every
    logits row and every argmax of the one-clip launch and of the clip-block sequence against both oracles."""
    case = ModelCase(tmp_path, "micro", 49, dtype=dtype, kind="realistic")
    clips = _clips(2)
    tol = TOL[("micro", dtype)]
    o_pol, o_f32 = case.oracle_bf16, case.oracle_fp32
    refs = []
    for c in clips:
        mel, _, _ = oracle_mod.log_mel(c, 80)
        ck, cv = o_pol.encoder(mel)
        ids, _ = o_pol.greedy(ck, cv, "zh", max_new=444, want_logits=True)
        forced = ids if len(ids) >= 444 else ids + [(13 * i + 7) % 50257 for i in range(444 - len(ids))]
        _, lg_p = o_pol.greedy(ck, cv, "zh", max_new=444, forced=forced, want_logits=True)
        ckf, cvf = o_f32.encoder(mel)
        _, lg_f = o_f32.greedy(ckf, cvf, "zh", max_new=444, forced=forced, want_logits=True)
        refs.append((mel, forced, lg_p, lg_f))
    for B in (1, 4):
        e = built_lib.Whisper("micro", case.root, "zh", device=0, max_batch=B)
        try:
            mels = np.stack([refs[b % 2][0] for b in range(B)])
            e.encode_mel(mels)
            logits, am = e.decode_forced(B, np.array([refs[b % 2][1] for b in range(B)], dtype=np.int32))
            wp = wf = 0.0
            ties = 0
            for b in range(B):
                _, _, lg_p, lg_f = refs[b % 2]
                for lg, which in ((lg_p, 0), (lg_f, 1)):
                    err = np.abs(logits[b] - lg).max(axis=1)
                    if which == 0:
                        wp = max(wp, float(err.max()))
                    else:
                        wf = max(wf, float(err.max()))
                    mg = _margins(lg)
                    want = lg.argmax(axis=1)
                    for s_ in np.nonzero(am[b] != want)[0]:
                        assert mg[s_] < 2 * err[s_] + 1e-4, (B, b, int(s_), float(mg[s_]), float(err[s_]))
                        ties += which == 0
            report.append(f"micro {dtype} full context B={B}: 445 logits rows, err vs policy oracle {wp:.3e}, vs fp32 oracle {wf:.3e}; ties accepted {ties}")
            assert wp < 2 * tol[0] and wf < 2 * tol[1], (wp, wf)   # 444 cached rows deep: twice the short-run bound
            _scan_ok(e, B, dtype, report, f"micro {dtype} full context B={B}")
        finally:
            e.close()
