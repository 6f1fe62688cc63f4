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
