import os
import sys
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "whisper.axera_amd", "tools"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_demo_pcm():
    w = wave.open(os.path.join(GOLDEN, "demo.wav"))
    return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / np.float32(32768.0)


@pytest.fixture(scope="session")
def demo_pcm():
    return load_demo_pcm()


@pytest.fixture(scope="session")
def built_lib():
    """libax_whisper.so, built once per session (hipcc cross-compiles without a GPU)."""
    import whisper_axera_amd as wa

    # always: make is incremental (compiler-generated header dependencies), so an edited source can never be tested
    # against a stale binary. On a box without hipcc the prebuilt library that travelled with the tree is used.
    import shutil

    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists(wa.LIB_PATH):
        wa.build()
    return wa


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build(ref=False)
    return oracle


class ModelCase:
    """A seeded synthetic model on disk + the matching oracle objects."""

    def __init__(self, tmpdir, model_type, seed, dtype="BF16", kind="benign"):
        """dtype "BF16": weights representable in bfloat16, engine in its bf16 build; "F16": weights representable in
        IEEE half, stored as F16, engine in its fp16 build. Either way both sides hold identical weight values and
        `oracle_bf16` is the oracle that narrows at the engine's 16-bit storage points (in that engine's type).
        kind "realistic": modelgen.realistic_weights (trained-model activation statistics) instead of N(0, 0.02)."""
        import modelgen
        import oracle

        self.model_type, self.seed, self.root, self.dtype, self.kind = model_type, seed, str(tmpdir), dtype, kind
        self.dims = modelgen.DIMS[model_type]
        if kind == "realistic":
            self.weights = modelgen.realistic_weights(self.dims, seed, dtype=dtype)
        elif dtype == "F16":
            w = modelgen.synth_weights(self.dims, seed, bf16=False)
            self.weights = {k: v.astype(np.float16).astype(np.float32) for k, v in w.items()}
        else:
            self.weights = modelgen.synth_weights(self.dims, seed)
        self.cfg = modelgen.make_config(model_type, self.dims)
        modelgen.write_model_dir(self.root, model_type, self.dims, weights=self.weights, dtype=dtype,
                                 tiktoken_path=os.path.join(GOLDEN, "multilingual.tiktoken"))
        self.oracle_fp32 = oracle.Oracle(self.cfg, self.weights, bf16_policy=False)
        self.oracle_bf16 = oracle.Oracle(self.cfg, self.weights, bf16_policy=2 if dtype == "F16" else True)


@pytest.fixture(scope="session")
def micro_case(tmp_path_factory, oracle_mod):
    return ModelCase(tmp_path_factory.mktemp("models_micro"), "micro", 11)


def assert_ids_equal_or_tie(engine, mel, got, ids, lg, what="", batch_mels=None, slot=0):
    """Greedy ids must equal the oracle's; the FIRST divergence is accepted only as a numerical tie: the oracle's own
    top-2 margin at that step must be below twice the logit error measured at that very step (the engine teacher-forced
    with the oracle's ids, so both sides see the same context) + 1e-4 — never a fixed margin. batch_mels: measure the
    error through the BATCHED decode path (all clips encoded, this clip in `slot`) instead of the 1-clip path.
    Returns the number of ids in agreement."""
    n = min(len(ids), len(got))
    if list(got[:n]) == list(ids[:n]):
        assert len(got) == len(ids), (what, len(got), len(ids))
        return n
    i = next(i for i in range(n) if ids[i] != got[i])
    if batch_mels is None:
        engine.encode_mel(mel)
        logits, _ = engine.decode_forced(1, np.array([list(ids[:i])], dtype=np.int32).reshape(1, i))
    else:
        # logits row i needs the ids before step i only; beyond 8 clips the error is measured in a 4-clip window around the
        # slot (still the batched decode sequence) so that a 64-clip, 444-id run does not ask for gigabytes of logits
        lo = 0
        if len(batch_mels) > 8:
            lo = min(max(slot - 1, 0), len(batch_mels) - 4)
            batch_mels = batch_mels[lo:lo + 4]
        engine.encode_mel(batch_mels)
        logits, _ = engine.decode_forced(len(batch_mels), np.array([list(ids[:i])] * len(batch_mels), dtype=np.int32).reshape(len(batch_mels), i))
        slot -= lo
    err = float(np.abs(logits[slot, i] - lg[i]).max())
    srt = np.sort(lg[i])
    margin = float(srt[-1] - srt[-2])
    assert margin < 2 * err + 1e-4, (what, "step", i, "margin", margin, "logit err", err, list(ids), list(got))
    return i
