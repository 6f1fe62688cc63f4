"""GPU: parity at the FULL Whisper-small size (BASELINE configs[1]/[2] dims: d=768, 12+12 layers, 51865 vocab) on
seeded synthetic weights: end-to-end ids and teacher-forced logits vs the CPU oracle (bf16 policy), single clip and
inside a 6-clip batch (the MFMA decode path). The oracle side costs a few CPU-seconds per clip."""
import numpy as np
import pytest
import torch  # noqa: F401

from conftest import ModelCase, load_demo_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_case(tmp_path_factory, oracle_mod):
    return ModelCase(tmp_path_factory.mktemp("models_small"), "small", 0)


@pytest.fixture(scope="module")
def engine(built_lib, small_case):
    e = built_lib.Whisper("small", small_case.root, "zh", device=0, max_batch=6)
    yield e
    e.close()


def _check(ids_gpu, ids_ref, logits_ref, logits_gpu):
    err = np.abs(logits_gpu - logits_ref).max(axis=1)
    assert err.max() < 6e-3, err.max()  # measured 1.6e-3 .. 1.8e-3 at d = 768 (12 layers of bf16 storage points), logit std 0.5
    if ids_gpu != ids_ref:
        i = next(i for i in range(len(ids_ref)) if ids_ref[i] != ids_gpu[i])
        srt = np.sort(logits_ref[i])
        assert srt[-1] - srt[-2] < 2 * err[i] + 1e-4, (i, ids_ref, ids_gpu)
    return float(err.max())


def test_small_single_clip_end_to_end(engine, small_case, oracle_mod):
    pcm = load_demo_pcm()
    n = 20
    mel, _, _ = oracle_mod.log_mel(pcm, 80)
    ck, cv = small_case.oracle_bf16.encoder(mel)
    ids, lg = small_case.oracle_bf16.greedy(ck, cv, "zh", max_new=n, want_logits=True)
    got = engine.run_tokens(pcm, max_new=n)
    engine.encode_mel(engine.compute_mel(pcm))
    logits, _ = engine.decode_forced(1, np.array([ids]))
    k, v = engine.get_cross_kv(0)
    # stored bf16 on both sides: a value may fall on either side of a rounding boundary -> two bf16 ulps of its magnitude
    dk, dv = np.abs(k - ck), np.abs(v - cv)
    print("small B=1 cross K/V max diff", dk.max(), dv.max(), "mean", dk.mean(), dv.mean(), "scale", np.abs(ck).max())
    assert (dk <= 2.0 ** -7 * np.maximum(np.abs(ck), 1.0)).all() and (dv <= 2.0 ** -7 * np.maximum(np.abs(cv), 1.0)).all()
    assert dk.mean() < 2e-3 and dv.mean() < 2e-3
    print("small B=1 logits err", _check(got, ids, lg, logits[0]))


def test_small_batched_clips(engine, small_case, oracle_mod):
    import modelgen

    clips = [load_demo_pcm()] + [modelgen.synth_clip(i, 480000 if i != 2 else 200000) for i in range(1, 6)]
    n = 12
    got = engine.run_tokens_batch(clips, max_new=n)
    mels = np.stack([engine.compute_mel(c) for c in clips])
    engine.encode_mel(mels)
    forced = np.zeros((6, n), dtype=np.int32)
    refs = {}
    for b in (1, 2):  # a full 30 s synthetic clip and a ragged one
        ck, cv = small_case.oracle_bf16.encoder(oracle_mod.log_mel(clips[b], 80)[0])
        ids, lg = small_case.oracle_bf16.greedy(ck, cv, "zh", max_new=n, want_logits=True)
        forced[b] = ids
        refs[b] = (ids, lg)
    for b in range(6):
        if b not in refs:
            forced[b] = forced[1]
    logits, _ = engine.decode_forced(6, forced)
    for b, (ids, lg) in refs.items():
        print("small B=6 clip", b, "logits err", _check(got[b], ids, lg, logits[b]))


def test_small_encoder_tile_shapes_agree(built_lib, small_case):
    """22 clips in one encoder pass run the 256x256-tile GEMM (enough tiles to fill the chip), one clip the 128x128
    tile: cross K/V of the same clip must agree within a bf16 ulp of its magnitude (the two kernels differ only in
    fp32 summation order and in where a value is narrowed to bf16)."""
    from make_model_goldens_inputs import demo_mel, synth_mel

    B = 22
    mels = [demo_mel(80), synth_mel(7, 80, 3000), synth_mel(8, 80, 1700)]
    e = built_lib.Whisper("small", small_case.root, "zh", device=0, max_batch=B)
    try:
        e.encode_mel(np.stack([mels[b % 3] for b in range(B)]))
        batched = {b: e.get_cross_kv(b) for b in (0, 1, 2, 20, 21)}
        for b, (kb, vb) in batched.items():
            e.encode_mel(mels[b % 3])
            k1, v1 = e.get_cross_kv(0)
            dk, dv = np.abs(kb - k1), np.abs(vb - v1)
            print(f"slot {b}: cross K/V batch-22 vs batch-1 max diff {dk.max():.3e} {dv.max():.3e}, mean {dk.mean():.2e}")
            tol = 2.0 ** -7 * np.maximum(np.abs(k1), 1.0)  # two bf16 ulps
            assert (dk <= tol).all() and (dv <= 2.0 ** -7 * np.maximum(np.abs(v1), 1.0)).all()
            assert dk.mean() < 2e-3 and dv.mean() < 2e-3
    finally:
        e.close()


def test_small_batched_decode_sequences_agree(built_lib, small_case, monkeypatch):
    """The clip-block decode sequence (LayerNorm as GEMM prologue, residual add as epilogue) and the older one
    (separate LayerNorm/bf16-pair launch, split-K partials) give the same teacher-forced logits and argmax ids."""
    from make_model_goldens_inputs import demo_mel, synth_mel

    B, n = 18, 8
    mels = np.stack([demo_mel(80) if b % 2 == 0 else synth_mel(30 + b, 80, 3000 if b % 3 else 1500) for b in range(B)])
    forced = np.tile(np.arange(100, 100 + n, dtype=np.int32), (B, 1))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AX_WHISPER_BATCHED_LN", mode)
        e = built_lib.Whisper("small", small_case.root, "zh", device=0, max_batch=B)
        try:
            assert e.L.AX_WHISPER_GetConfigInt(e.h, b"batched_ln") == int(mode)
            e.encode_mel(mels)
            out[mode] = e.decode_forced(B, forced)
        finally:
            e.close()
    lg1, am1 = out["1"]
    lg0, am0 = out["0"]
    err = np.abs(lg1 - lg0).max()
    print("clip-block vs split-K sequence: logits diff", err)
    assert err < 2e-3
    srt = np.sort(lg0, axis=2)
    margin = srt[:, :, -1] - srt[:, :, -2]
    assert ((am1 == am0) | (margin < 4e-3)).all()
