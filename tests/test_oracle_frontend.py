"""CPU: the oracle's front-end against the goldens captured from the reference's own C++ front-end
(tests/golden/make_frontend_goldens.py; librosa.h + Whisper::preprocess), tolerance 2e-4 abs
(the reference computes in fp32 with kissfft; SURVEY §8c measured 7.5e-5..1.3e-4 for a float64 restatement)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_demo_pcm

TOL = 2e-4


@pytest.mark.parametrize("n_mels", [80, 128])
def test_demo_wav_matches_reference_frontend(oracle_mod, n_mels):
    g = np.load(os.path.join(GOLDEN, f"frontend_demo_{n_mels}.npz"))
    pcm = load_demo_pcm()
    assert len(pcm) == int(g["n_samples"]) == 67263
    mel, nf, mmax = oracle_mod.log_mel(pcm, n_mels)
    assert nf == int(g["n_frames"]) == 421
    assert abs(mmax - float(g["mmax"])) < 1e-5
    assert np.abs(mel[:, :nf] - g["mel_real"]).max() < TOL
    assert np.all(mel[:, nf:] == 0.0)  # feature-space zero padding (Whisper.cpp:172)


def test_survey_appendix_c_values(oracle_mod):
    """Spot values recorded in SURVEY.md Appendix C for demo.wav, 80 mels."""
    mel, nf, mmax = oracle_mod.log_mel(load_demo_pcm(), 80)
    assert abs(mmax - (-0.453550041)) < 1e-5
    np.testing.assert_allclose(mel[0, :3], [0.266238868, 0.254720032, 0.128041804], atol=TOL)
    assert abs(mel[0, 420] - 0.314599454) < TOL and mel[0, 421] == 0.0
    assert abs(mel[40, 100] - (-0.106739402)) < TOL
    assert abs(mel.min() - (-1.11338758)) < TOL


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filterbank_exact(oracle_mod, n_mels):
    g = np.load(os.path.join(GOLDEN, f"melfilter_{n_mels}.npz"))
    np.testing.assert_allclose(oracle_mod.mel_filterbank(n_mels), g["fb"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("name,clip,n", [("synth0_30s", 0, 480000), ("synth3_7777", 3, 7777), ("synth5_1s", 5, 16000)])
def test_seeded_clips(oracle_mod, name, clip, n):
    import modelgen

    g = np.load(os.path.join(GOLDEN, f"frontend_{name}.npz"))
    x = modelgen.synth_clip(clip, n)
    mel, nf, mmax = oracle_mod.log_mel(x, 80)
    assert nf == int(g["n_frames"]) == 1 + n // 160
    assert abs(mmax - float(g["mmax"])) < 1e-5
    assert np.abs(mel[:, g["idx"]] - g["mel_sub"]).max() < TOL
    assert np.all(mel[:, min(nf, 3000):] == 0.0)


def test_long_clip_floor_comes_from_all_frames(oracle_mod):
    """75 s, loudest second at 70 s (Whisper.cpp:158-172: the maximum is taken over ALL frames of the input, then 3000
    are kept): golden from the reference's own front-end; cutting the input anywhere before 70 s changes the floor."""
    import modelgen

    g = np.load(os.path.join(GOLDEN, "frontend_long75s_loud70.npz"))
    x = modelgen.synth_long_clip(75, 70)
    mel, nf, mmax = oracle_mod.log_mel(x, 80)
    assert nf == int(g["n_frames"]) == 7501 and len(x) == int(g["n_samples"])
    assert abs(mmax - float(g["mmax"])) < 1e-5
    assert np.abs(mel[:, g["idx"]] - g["mel_sub"]).max() < TOL
    cut, _, mmax_cut = oracle_mod.log_mel(x[:960000], 80)
    assert mmax_cut < mmax - 1.0 and np.abs(cut - mel).max() > 0.1  # the cap this round removed WAS a result difference


def test_against_live_reference_build(oracle_mod):
    """Where oracle/_ref was built (the container with /root/reference) compare live, full frames."""
    if oracle_mod.ref_lib() is None:
        pytest.skip("oracle/_ref not built here")
    import modelgen

    for i, n in [(1, 480000), (2, 123457), (4, 401)]:
        x = modelgen.synth_clip(i, n)
        a, nfa, ma = oracle_mod.log_mel(x, 80)
        b, nfb, mb = oracle_mod.log_mel(x, 80, use_ref=True)
        assert nfa == nfb and abs(ma - mb) < 1e-5
        assert np.abs(a - b).max() < TOL


def test_openai_feature_mode_matches_torch_stft(oracle_mod, demo_pcm):
    """feature_mode "openai" (SURVEY A.1 column 3, the fp32 ONNX lineage's front-end, model_convert/generate_data.py:
    162-176): the oracle's restatement against an independent evaluation of the same published recipe with torch.stft
    (openai-whisper's own module is not importable here): pad to 30 s, STFT(400, 160, Hann, center/reflect), drop the
    last frame, |X|^2, Slaney mel, log10/clamp/scale."""
    import modelgen
    import torch

    for pcm, n_mels in ((demo_pcm, 80), (modelgen.synth_clip(2, 480000), 128), (modelgen.synth_clip(3, 500123), 80)):
        got, mmax = oracle_mod.log_mel_openai(pcm, n_mels)
        x = torch.zeros(480000, dtype=torch.float32)
        n = min(len(pcm), 480000)
        x[:n] = torch.from_numpy(pcm[:n])
        st = torch.stft(x, 400, 160, window=torch.hann_window(400), return_complex=True)
        mag = st[..., :-1].abs() ** 2
        assert mag.shape == (201, 3000)
        # Slaney filterbank: the C++ front-end's (float32 arithmetic) agrees with the float64 one to ~1e-7 relative
        fb = torch.from_numpy(oracle_mod.mel_filterbank(n_mels))
        mel = fb @ mag
        lg = torch.clamp(mel, min=1e-10).log10()
        lg = torch.maximum(lg, lg.max() - 8.0)
        want = ((lg + 4.0) / 4.0).numpy()
        assert abs(float(lg.max()) - mmax) < 1e-4
        err = np.abs(got - want).max()
        assert err < 2e-4, err
    # versus the C++ pipeline on a short clip: real frames agree except next to the clip's end (reflection of the real
    # end vs zeros) — and the padded region is the clamp floor, not 0
    a, nfr, _ = oracle_mod.log_mel(demo_pcm, 80)
    b, mmax = oracle_mod.log_mel_openai(demo_pcm, 80)
    assert np.abs(a[:, : nfr - 3] - b[:, : nfr - 3]).max() < 2e-4
    assert np.all(a[:, nfr:] == 0.0) and np.allclose(b[:, nfr + 2:], (mmax - 8.0 + 4.0) / 4.0, atol=1e-6)
