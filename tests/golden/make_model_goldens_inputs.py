"""Inputs of the model goldens, importable without torch/transformers (used by the tests)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def synth_mel(seed, n_mels, n_real):
    """A mel-like input: random values in the front-end's range, zeros past n_real."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mel = np.zeros((n_mels, 3000), dtype=np.float32)
    mel[:, :n_real] = np.clip(rng.standard_normal((n_mels, n_real)).astype(np.float32) * 0.4, -1.0, 1.5)
    return mel


def demo_mel(n_mels=80):
    g = np.load(os.path.join(HERE, f"frontend_demo_{n_mels}.npz"))
    mel = np.zeros((n_mels, 3000), dtype=np.float32)
    mel[:, : int(g["n_frames"])] = g["mel_real"]
    return mel


def golden_mel(name):
    return {"micro_demo": lambda: demo_mel(80), "micro_synth": lambda: synth_mel(5, 80, 3000),
            "mini_synth": lambda: synth_mel(6, 80, 1777), "tiny_demo": lambda: demo_mel(80),
            "small_demo": lambda: demo_mel(80), "miniturbo_synth": lambda: synth_mel(9, 128, 2500),
            "real_micro_demo": lambda: demo_mel(80), "real_mini_synth": lambda: synth_mel(6, 80, 1777),
            "real_miniturbo_synth": lambda: synth_mel(9, 128, 2500), "real_tiny_demo": lambda: demo_mel(80),
            "real_small_demo": lambda: demo_mel(80)}[name]()
