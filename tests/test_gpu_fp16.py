"""GPU: the engine's IEEE-half build (csrc/common.hpp AXW_F16=1: fp16 weights, MFMA operands, K/V caches and activation
pairs; fp32 accumulation / LayerNorm / softmax / residual stream) — BASELINE configs[3] names "Whisper-turbo fp16".
A model whose weights file is F16 selects it (AX_WHISPER_DTYPE overrides). Parity bar as for the bf16 build: logits vs
the oracle narrowing at the same storage points in half (policy 2) within 2e-2 abs (measured ~1e-3: half carries 3 more
mantissa bits than bfloat16), greedy ids identical or a measured numerical tie, on all three decode paths."""
import os

import numpy as np
import pytest
import torch  # noqa: F401

from conftest import ModelCase, assert_ids_equal_or_tie, load_demo_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def micro16(tmp_path_factory, oracle_mod):
    return ModelCase(tmp_path_factory.mktemp("models_micro_f16"), "micro", 31, dtype="F16")


def _mels(n):
    from make_model_goldens_inputs import demo_mel, synth_mel

    return [demo_mel(80)] + [synth_mel(300 + i, 80, 3000 if i % 2 else 1200 + 50 * i) for i in range(1, n)]


def test_f16_weights_select_the_half_build(built_lib, micro16, micro_case, monkeypatch):
    e = built_lib.Whisper("micro", micro16.root, "zh", device=0)
    assert e.L.AX_WHISPER_GetConfigInt(e.h, b"fp16") == 1
    e.close()
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
    assert e.L.AX_WHISPER_GetConfigInt(e.h, b"fp16") == 0
    e.close()
    monkeypatch.setenv("AX_WHISPER_DTYPE", "fp16")  # bf16-representable weights are half-representable here (|w| < 1, 8 bits)
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
    assert e.L.AX_WHISPER_GetConfigInt(e.h, b"fp16") == 1
    ids = e.run_tokens(load_demo_pcm(), max_new=8)
    assert len(ids) == 8
    e.close()
    monkeypatch.setenv("AX_WHISPER_DTYPE", "int8")
    with pytest.raises(RuntimeError, match="AX_WHISPER_DTYPE"):
        built_lib.Whisper("micro", micro_case.root, "zh", device=0)


def test_fp16_encoder_cross_kv_vs_oracle(built_lib, micro16):
    e = built_lib.Whisper("micro", micro16.root, "zh", device=0)
    try:
        mel = _mels(1)[0]
        e.encode_mel(mel)
        k, v = e.get_cross_kv(0)
        kb, vb = micro16.oracle_bf16.encoder(mel)
        kf, vf = micro16.oracle_fp32.encoder(mel)
        ek, ev = np.abs(k - kb).max(), np.abs(v - vb).max()
        print(f"fp16 cross K/V vs half-policy oracle {ek:.2e} {ev:.2e}; vs fp32 oracle {np.abs(k - kf).max():.2e}")
        # one half ulp at |x| ~ 2-4 is 2e-3 (bfloat16: 1.6e-2)
        assert ek < 2e-3 and ev < 2e-3  # one half ulp at 1..2 is 9.8e-4, and that is what is measured
        assert np.abs(k - kf).max() < 3e-3  # measured 9.5e-4
    finally:
        e.close()


# 3 clips: clip-block GEMMs with the cross-attention split over several workgroups; "gemv": the fp32-FMA family (the path
# of one or two clips outside the persistent launch), forced for 2 and 3 clips
@pytest.mark.parametrize("batch,mode", [(1, "persistent"), (1, "graph"), (2, "graph"), (3, "gemv"), (3, "mfma"), (7, "mfma"), (20, "mfma")])
def test_fp16_decode_paths_vs_oracle(built_lib, micro16, monkeypatch, batch, mode):
    if mode == "graph":
        monkeypatch.setenv("AX_WHISPER_DECODE", "graph")
    if mode == "gemv":
        monkeypatch.setenv("AX_WHISPER_GEMV_MAX", "4")
    e = built_lib.Whisper("micro", micro16.root, "zh", device=0, max_batch=batch)
    try:
        mels = _mels(min(batch, 4))
        allm = np.stack([mels[b % len(mels)] for b in range(batch)])
        e.encode_mel(allm)
        got = e.decode_greedy(batch, max_new=14)
        n = 14
        refs = {}
        for i, mel in enumerate(mels):
            ck, cv = micro16.oracle_bf16.encoder(mel)
            refs[i] = micro16.oracle_bf16.greedy(ck, cv, "zh", max_new=n, want_logits=True)
        forced = np.array([refs[b % len(mels)][0] for b in range(batch)], dtype=np.int32)
        logits, am = e.decode_forced(batch, forced)
        worst = 0.0
        for b in range(batch):
            ids, lg = refs[b % len(mels)]
            err = np.abs(logits[b] - lg).max(axis=1)
            worst = max(worst, float(err.max()))
            srt = np.sort(lg, axis=1)
            for s in range(n + 1):
                assert am[b, s] == int(lg[s].argmax()) or srt[s, -1] - srt[s, -2] < 2 * err[s] + 1e-4, (b, s)
            if got[b] != ids:
                i = next(i for i in range(n) if got[b][i] != ids[i])
                assert srt[i, -1] - srt[i, -2] < 2 * err[i] + 1e-4, (b, i, got[b], ids)
        print(f"fp16 {mode} B={batch}: logits err vs half-policy oracle {worst:.2e}")
        assert worst < 2e-4  # measured 2.1e-5 .. 3.9e-5
    finally:
        e.close()


def test_fp16_end_to_end_vs_fp32_oracle(built_lib, oracle_mod, tmp_path):
    """tiny dims on demo.wav (BASELINE configs[0] shape) through the half build against the PURE-fp32 oracle."""
    case = ModelCase(tmp_path, "tiny", 33, dtype="F16")
    e = built_lib.Whisper("tiny", case.root, "zh", device=0)
    try:
        pcm = load_demo_pcm()
        got = e.run_tokens(pcm, max_new=60)
        mel, _, _ = oracle_mod.log_mel(pcm, 80)
        ck, cv = case.oracle_fp32.encoder(mel)
        ids, lg = case.oracle_fp32.greedy(ck, cv, "zh", max_new=60, want_logits=True)
        agree = assert_ids_equal_or_tie(e, mel, got, ids, lg, "tiny fp16 vs fp32 oracle")
        e.encode_mel(e.compute_mel(pcm))
        logits, _ = e.decode_forced(1, np.array([ids], dtype=np.int32))
        err = float(np.abs(logits[0] - lg).max())
        print(f"tiny fp16 / demo.wav: {agree}/{len(ids)} ids equal to the fp32 oracle, logits err {err:.2e}")
        assert err < 1.5e-3  # measured 2.8e-4 (fp16 build vs the PURE fp32 oracle)
    finally:
        e.close()
