"""GPU: parity of the HIP path (through the C ABI) against the CPU oracle and the committed goldens.

Tolerances (written here, justified in DESIGN.md §Numerics):
  front-end        2e-4 abs on the normalised log-mel (fp32 both sides; same bar the oracle meets vs the reference)
  cross K/V        vs bf16-policy oracle: 2e-2 abs (values up to ~2; ONE bf16 ulp at 2.0 is 1.56e-2 and that is what is
                   measured: a value on a rounding boundary lands on either side), mean abs 1.5e-3 (measured 5.3e-4)
                   vs fp32 oracle:        2e-2 abs (measured 8.7e-3 = half an ulp of storage + the arithmetic)
  logits           vs bf16-policy oracle: 1e-3 abs (measured 1.4e-4 .. 1.8e-4); vs fp32 oracle 5e-3 abs (measured 1.1e-3);
                   logit std ~0.23
  token ids        equal to the bf16-policy oracle's, except where the oracle's own top-2 margin is below 2x the
                   measured logit error at that step + 1e-4 (a numerical tie)
(round 4: every bound cut to at most ~5x what the test prints on MI355X — a 1e-2 logit defect cannot pass anywhere.)
"""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (imported before libax_whisper.so so both share torch's HIP runtime in this process)

from conftest import GOLDEN, ModelCase, assert_ids_equal_or_tie, load_demo_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(built_lib, micro_case):
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=4)
    yield e
    e.close()


# ------------------------------------------------------------------ front-end
@pytest.mark.parametrize("n_mels", [80])
def test_frontend_demo_wav_vs_reference_golden(engine, n_mels):
    g = np.load(os.path.join(GOLDEN, f"frontend_demo_{n_mels}.npz"))
    mel = engine.compute_mel(load_demo_pcm())
    nf = int(g["n_frames"])
    err = np.abs(mel[:, :nf] - g["mel_real"]).max()
    print("frontend demo err", err)
    assert err < 2e-4
    assert np.all(mel[:, nf:] == 0.0)


@pytest.mark.parametrize("clip,n", [(0, 480000), (3, 7777), (5, 16000), (2, 123457), (4, 401), (6, 480160), (1, 521000)])
def test_frontend_seeded_clips_vs_oracle(engine, oracle_mod, clip, n):
    """ragged lengths, a clip shorter than one FFT, exactly one frame over 30 s, and > 30 s (truncated to 3000 frames
    while the clamp floor still comes from ALL frames, Whisper.cpp:158-172)."""
    import modelgen

    x = modelgen.synth_clip(clip, n)
    ref, nf, _ = oracle_mod.log_mel(x, 80)
    mel = engine.compute_mel(x)
    err = np.abs(mel - ref).max()
    print("frontend", n, "err", err)
    assert err < 2e-4
    assert np.all(mel[:, min(nf, 3000):] == 0.0)


def test_frontend_long_clip_vs_reference_golden(engine, oracle_mod):
    """75 s input, loudest second at 70 s: beyond the 30 s window and beyond the engine's 60 s staging row. The clamp
    floor of the 3000 kept frames comes from the maximum over ALL 7501 frames (Whisper.cpp:158-172); golden from the
    reference's own librosa.h build (tests/golden/make_frontend_goldens.py)."""
    import modelgen

    g = np.load(os.path.join(GOLDEN, "frontend_long75s_loud70.npz"))
    x = modelgen.synth_long_clip(75, 70)
    mel = engine.compute_mel(x)
    err = np.abs(mel[:, g["idx"]] - g["mel_sub"]).max()
    ref, _, mmax = oracle_mod.log_mel(x, 80)
    err_all = np.abs(mel - ref).max()
    print("frontend 75 s: vs reference golden", err, "vs oracle (all frames)", err_all)
    assert err < 2e-4 and err_all < 2e-4
    # the same clip through the transcription entry point (staging rows + packed tails) next to a 10-minute clip in one
    # batch: both get their own floor
    y = np.concatenate([modelgen.synth_clip(9, 9 * 60 * 16000) * np.float32(0.02), modelgen.synth_clip(9, 16000)])
    for clip in (x, y):
        ref, _, _ = oracle_mod.log_mel(clip, 80)
        assert np.abs(engine.compute_mel(clip) - ref).max() < 2e-4
    ids_pair = engine.run_tokens_batch([x, y, x[:480000]], max_new=6)
    assert ids_pair[0] == engine.run_tokens(x, max_new=6) and ids_pair[1] == engine.run_tokens(y, max_new=6)


def test_frontend_silence_and_dc(engine, oracle_mod):
    for x in (np.zeros(16000, np.float32), np.full(32000, 0.25, np.float32)):
        ref, _, _ = oracle_mod.log_mel(x, 80)
        assert np.abs(engine.compute_mel(x) - ref).max() < 2e-4


# ------------------------------------------------------------------ encoder
def _mels():
    from make_model_goldens_inputs import demo_mel, synth_mel

    return [demo_mel(80), synth_mel(5, 80, 3000), synth_mel(6, 80, 1777)]


def test_encoder_cross_kv_vs_oracle(engine, micro_case):
    mels = _mels()
    engine.encode_mel(np.stack(mels))
    for slot, mel in enumerate(mels):
        k, v = engine.get_cross_kv(slot)
        kb, vb = micro_case.oracle_bf16.encoder(mel)
        kf, vf = micro_case.oracle_fp32.encoder(mel)
        eb = max(np.abs(k - kb).max(), np.abs(v - vb).max())
        ef = max(np.abs(k - kf).max(), np.abs(v - vf).max())
        mb = max(np.abs(k - kb).mean(), np.abs(v - vb).mean())
        print(f"slot {slot}: vs bf16-policy max {eb:.3e} mean {mb:.3e}; vs fp32 max {ef:.3e}; scale {np.abs(kf).max():.2f}")
        assert eb < 2e-2 and mb < 1.5e-3 and ef < 2e-2


def test_encoder_vs_transformers_golden(engine):
    """The committed HF-generated golden directly against the GPU (micro_demo uses this module's weights seed 11)."""
    from make_model_goldens_inputs import golden_mel

    g = np.load(os.path.join(GOLDEN, "model_micro_demo.npz"))
    assert int(g["seed"]) == 11
    engine.encode_mel(golden_mel("micro_demo"))
    k, v = engine.get_cross_kv(0)
    ek, ev = np.abs(k[:, ::53, ::7] - g["cross_k_sub"]).max(), np.abs(v[:, ::53, ::7] - g["cross_v_sub"]).max()
    print("cross K/V vs the transformers golden (fp32)", ek, ev)
    assert ek < 2e-2 and ev < 2e-2
    assert abs(np.abs(k).astype(np.float64).sum() / float(g["cross_k_abs"]) - 1) < 5e-3


def test_encoder_attention_rescale_threshold(built_lib, micro_case, monkeypatch):
    """encoder_attn.hip raises the running softmax maximum only when a tile's row maximum exceeds it by more than a
    threshold (2^8 by default). The branch is data dependent, so: threshold 0 (rescale on every increase, the classic
    online softmax), 0.5 (branch taken on some tiles, skipped on others) and the default must agree to rounding, on
    inputs whose late frames are much louder than the early ones (the maximum keeps rising along the keys) and on the
    reverse; every variant is also held against the bf16-policy oracle."""
    from make_model_goldens_inputs import synth_mel

    base = synth_mel(9, 80, 3000)
    ramp = np.linspace(-1.5, 1.5, 3000, dtype=np.float32)[None, :]
    mels = [base, np.clip(base * 0.2 + ramp, -1.5, 2.0).astype(np.float32), np.clip(base * 0.2 - ramp, -1.5, 2.0).astype(np.float32)]
    outs = {}
    for thr in ("0", "0.5", None):
        if thr is None:
            monkeypatch.delenv("AX_WHISPER_ENC_RESCALE_THR", raising=False)
        else:
            monkeypatch.setenv("AX_WHISPER_ENC_RESCALE_THR", thr)
        e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=3)
        try:
            e.encode_mel(np.stack(mels))
            outs[thr] = [e.get_cross_kv(i) for i in range(3)]
        finally:
            e.close()
    for i, mel in enumerate(mels):
        kb, vb = micro_case.oracle_bf16.encoder(mel)
        for thr, o in outs.items():
            k, v = o[i]
            err = max(np.abs(k - kb).max(), np.abs(v - vb).max())
            print(f"mel {i} threshold {thr}: vs bf16-policy oracle {err:.3e}")
            assert err < 2e-2
        for thr in ("0.5", None):
            dk = max(np.abs(outs[thr][i][0] - outs["0"][i][0]).max(), np.abs(outs[thr][i][1] - outs["0"][i][1]).max())
            print(f"mel {i} threshold {thr} vs 0: {dk:.3e}")
            assert dk < 2e-2  # h16 roundings of P at a different scale, a few ulps of the outputs


# ------------------------------------------------------------------ decoder
def test_decoder_teacher_forced_logits(engine, micro_case):
    mels = _mels()
    B = len(mels)
    engine.encode_mel(np.stack(mels))
    forced, ref_logits = [], []
    for mel in mels:
        ck, cv = micro_case.oracle_bf16.encoder(mel)
        ids, lg = micro_case.oracle_bf16.greedy(ck, cv, "zh", max_new=24, want_logits=True)
        ids = (ids + [0] * 24)[:24]
        _, lg = micro_case.oracle_bf16.greedy(ck, cv, "zh", max_new=24, forced=ids, want_logits=True)
        forced.append(ids)
        ref_logits.append(lg)
    logits, am = engine.decode_forced(B, np.array(forced))
    for b in range(B):
        ref = ref_logits[b]
        err = np.abs(logits[b] - ref).max(axis=1)
        print(f"clip {b}: logits err max {err.max():.3e}, logit std {ref.std():.3f}")
        assert err.max() < 1e-3
        srt = np.sort(ref, axis=1)
        margin = srt[:, -1] - srt[:, -2]
        for s in range(ref.shape[0]):
            assert am[b, s] == int(ref[s].argmax()) or margin[s] < 2 * err[s] + 1e-4, (b, s, margin[s], err[s])


def test_decoder_given_oracle_cross_kv_isolated(engine, micro_case):
    """fp32 oracle end to end as the loosest bar: logits within 5e-3 abs (measured 1.1e-3)."""
    mel = _mels()[0]
    engine.encode_mel(mel)
    ck, cv = micro_case.oracle_fp32.encoder(mel)
    ids, lg = micro_case.oracle_fp32.greedy(ck, cv, "zh", max_new=12, want_logits=True)
    logits, _ = engine.decode_forced(1, np.array([ids]))
    err = np.abs(logits[0, : len(lg)] - lg).max()
    print("vs fp32 oracle logits err", err)
    assert err < 5e-3


def test_greedy_ids_match_oracle(engine, micro_case):
    mels = _mels()
    engine.encode_mel(np.stack(mels))
    got = engine.decode_greedy(len(mels), max_new=16)
    for b, mel in enumerate(mels):
        ck, cv = micro_case.oracle_bf16.encoder(mel)
        ids, lg = micro_case.oracle_bf16.greedy(ck, cv, "zh", max_new=16, want_logits=True)
        assert len(got[b]) == len(ids) == 16
        assert_ids_equal_or_tie(engine, mel, got[b], ids, lg, f"clip {b}")  # only a numerical tie may differ


def test_batch_rows_are_independent(engine):
    """A clip decodes to the same ids alone (B=1) and inside a batch (B=3), at any slot."""
    mels = _mels()
    engine.encode_mel(np.stack(mels))
    together = engine.decode_greedy(3, max_new=12)
    for b, mel in enumerate(mels):
        engine.encode_mel(mel)
        assert engine.decode_greedy(1, max_new=12)[0] == together[b]


def test_full_context_run_stops_at_444_ids(engine, micro_case):
    """Whisper.cpp:219-222: without eot the loop ends when offset reaches n_text_ctx -> 444 ids."""
    ids = engine.run_tokens(load_demo_pcm())
    eot = micro_case.cfg["eot"]
    assert len(ids) <= 444 and eot not in ids
    if len(ids) < 444:  # stopped on eot: the oracle must agree it is (nearly) the argmax there
        pytest.skip("synthetic weights emitted eot")
    assert len(ids) == 444


# ------------------------------------------------------------------ end to end through the legacy ABI
def test_run_pcm_and_run_file_agree(engine, tmp_path):
    pcm = load_demo_pcm()
    ids = engine.run_tokens(pcm, max_new=10)
    text_pcm = engine.run(pcm)
    text_file = engine.run(os.path.join(GOLDEN, "demo.wav"))
    assert text_pcm == text_file
    full = engine.run_tokens(pcm)
    assert full[:10] == ids
    assert engine.detokenize(full).decode("utf-8", errors="replace") == text_pcm


def test_end_to_end_ids_vs_oracle(engine, micro_case):
    pcm = load_demo_pcm()
    got = engine.run_tokens(pcm, max_new=12)
    mel, _, _ = __import__("oracle").log_mel(pcm, 80)
    ck, cv = micro_case.oracle_bf16.encoder(mel)
    ids, lg = micro_case.oracle_bf16.greedy(ck, cv, "zh", max_new=12, want_logits=True)
    assert_ids_equal_or_tie(engine, mel, got, ids, lg)


def test_ragged_batch_through_host_pointers(engine):
    import modelgen

    clips = [load_demo_pcm(), modelgen.synth_clip(1, 480000), modelgen.synth_clip(2, 30000)]
    together = engine.run_tokens_batch(clips, max_new=8)
    for b, c in enumerate(clips):
        assert engine.run_tokens(c, max_new=8) == together[b]
    texts = engine.run_batch(clips)
    assert len(texts) == 3 and all(isinstance(t, str) for t in texts)


def test_device_resident_input(engine):
    import torch

    import modelgen

    clips = np.stack([modelgen.synth_clip(i, 480000) for i in range(2)])
    d = torch.from_numpy(clips).cuda()
    torch.cuda.synchronize()
    got = engine.run_device_tokens(d.data_ptr(), 480000, [480000, 480000], max_new=6)
    assert got == engine.run_tokens_batch(list(clips), max_new=6)


def test_error_behaviour(engine, tmp_path):
    import ctypes as C

    out = C.c_void_p(123)
    L = engine.L
    assert L.AX_WHISPER_RunFile(engine.h, str(tmp_path / "missing.wav").encode(), C.byref(out)) == -1
    assert out.value is None  # *result = nullptr before the failure (ax_whisper_api.cpp:98)
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"not a wav file at all")
    assert L.AX_WHISPER_RunFile(engine.h, str(bad).encode(), C.byref(out)) == -1
    assert L.AX_WHISPER_RunPCM(engine.h, None, 10, C.byref(out)) == -1


def test_language_fallback_and_selection(built_lib, micro_case):
    """Whisper.cpp:241-251: unknown language -> zh; known language -> its token in the SOT sequence."""
    e = built_lib.Whisper("micro", micro_case.root, "xx", device=0)
    assert e.sot_seq == [50258, 50260, 50359, 50363]
    e.close()
    e = built_lib.Whisper("micro", micro_case.root, "en", device=0)
    assert e.sot_seq == [50258, 50259, 50359, 50363]
    e.close()


def test_detokenizer_bytes(engine):
    import base64

    lines = open(os.path.join(GOLDEN, "multilingual.tiktoken")).read().splitlines()
    ids = [0, 1, 255, 1000, 30000, 50255, 50256, 50257, 51864]  # the last two are specials: skipped
    want = b"".join(base64.b64decode(lines[i].split()[0]) for i in ids if i < 50257)
    assert engine.detokenize(ids) == want


def test_feature_mode_openai_front_end(built_lib, micro_case, oracle_mod, monkeypatch, tmp_path):
    """SURVEY A.1 column 3 behind a switch (config key "feature_mode" / env AX_WHISPER_FEATURE_MODE): the front-end of the
    fp32 ONNX lineage (model_convert/generate_data.py:162-176) — 30 s zero padding before the STFT, last frame
    dropped, clamp floor instead of zeros behind the clip. Default stays the C++ runtime's pipeline."""
    import json
    import shutil

    import modelgen

    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
    assert e.L.AX_WHISPER_GetConfigInt(e.h, b"feature_mode_openai") == 0
    e.close()
    monkeypatch.setenv("AX_WHISPER_FEATURE_MODE", "openai")
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=3)
    try:
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"feature_mode_openai") == 1
        clips = [load_demo_pcm(), modelgen.synth_clip(1, 480000), modelgen.synth_clip(2, 500321), modelgen.synth_clip(3, 401)]
        for pcm in clips:
            want, _ = oracle_mod.log_mel_openai(pcm, 80)
            got = e.compute_mel(pcm)
            err = float(np.abs(got - want).max())
            print("openai-mode mel err", len(pcm), err)
            assert err < 2e-4
        # end to end: ids follow the openai-mode mel (oracle fed with the same mel)
        pcm = clips[0]
        mel, _ = oracle_mod.log_mel_openai(pcm, 80)
        ck, cv = micro_case.oracle_bf16.encoder(mel)
        ids, lg = micro_case.oracle_bf16.greedy(ck, cv, "zh", max_new=10, want_logits=True)
        assert_ids_equal_or_tie(e, mel, e.run_tokens(pcm, max_new=10), ids, lg, "openai feature mode")
        assert e.run_tokens_batch(clips[:3], max_new=6)[0] == e.run_tokens(pcm, max_new=6)
    finally:
        e.close()
    monkeypatch.delenv("AX_WHISPER_FEATURE_MODE")
    # the same through the config file; and a wrong value is an Init failure, not a silent default
    root = tmp_path / "m"
    shutil.copytree(micro_case.root, root)
    cfgp = root / "micro" / "micro_config.json"
    cfg = json.load(open(cfgp))
    cfg["feature_mode"] = "openai"
    json.dump(cfg, open(cfgp, "w"))
    e = built_lib.Whisper("micro", str(root), "zh", device=0)
    assert e.L.AX_WHISPER_GetConfigInt(e.h, b"feature_mode_openai") == 1
    e.close()
    cfg["feature_mode"] = "kaldi"
    json.dump(cfg, open(cfgp, "w"))
    with pytest.raises(RuntimeError, match="feature_mode"):
        built_lib.Whisper("micro", str(root), "zh", device=0)
