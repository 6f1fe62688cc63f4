"""Generate front-end goldens from the REFERENCE's own C++ front-end (oracle/_ref).

Run only in the container that has /root/reference (``make -C oracle ref`` first):
    python tests/golden/make_frontend_goldens.py
Outputs (committed): tests/golden/frontend_*.npz — inputs are demo.wav (a data file the
reference ships) and seeded synthetic clips (regenerated from the seed at test time);
expected outputs are what librosa.h + Whisper::preprocess produce on them.
Only the real frames are stored (the rest of the 3000-frame window is exactly 0.0).
"""
import os, sys, wave
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import oracle, modelgen  # noqa: E402


def load_wav(path):
    w = wave.open(path)
    return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / np.float32(32768.0)


def main():
    assert oracle.ref_lib() is not None, "build oracle/_ref first (make -C oracle ref)"
    pcm = load_wav(os.path.join(HERE, "demo.wav"))
    for nm in (80, 128):
        mel, nf, mmax = oracle.log_mel(pcm, nm, use_ref=True)
        assert np.all(mel[:, nf:] == 0)
        np.savez_compressed(os.path.join(HERE, f"frontend_demo_{nm}.npz"), mel_real=mel[:, :nf].astype(np.float32),
                            n_frames=nf, mmax=np.float32(mmax), n_samples=len(pcm))
        fb = oracle.mel_filterbank(nm, use_ref=True)
        np.savez_compressed(os.path.join(HERE, f"melfilter_{nm}.npz"), fb=fb)
    # seeded clips: a 30 s clip (3001 frames -> truncated), a short ragged one, a 1-frame-over one
    # ... and a 75 s clip whose loudest second is at 70 s: the maximum over ALL frames sets the clamp floor of the 3000
    # kept ones (Whisper.cpp:158-172)
    cases = {"synth0_30s": modelgen.synth_clip(0), "synth3_7777": modelgen.synth_clip(3, 7777),
             "synth5_1s": modelgen.synth_clip(5, 16000), "long75s_loud70": modelgen.synth_long_clip(75, 70)}
    for name, x in cases.items():
        mel, nf, mmax = oracle.log_mel(x, 80, use_ref=True)
        keep = min(nf, 3000)
        # subsample frames to keep the fixture small: every 7th frame + the last 4 kept ones
        idx = np.unique(np.concatenate([np.arange(0, keep, 7), np.arange(max(keep - 4, 0), keep)]))
        np.savez_compressed(os.path.join(HERE, f"frontend_{name}.npz"), mel_sub=mel[:, idx], idx=idx,
                            n_frames=nf, mmax=np.float32(mmax), n_samples=len(x),
                            tail_zero=bool(np.all(mel[:, keep:] == 0)))
    print("ok")


if __name__ == "__main__":
    main()
