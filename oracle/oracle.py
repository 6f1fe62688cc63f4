"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module. The product package never does. See whisper_oracle.h for what is restated and
how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

fp = C.POINTER(C.c_float)


class OrcPolicy(C.Structure):
    _fields_ = [("bf16_policy", C.c_int), ("literal_mask", C.c_int)]


class OrcBlock(C.Structure):
    _fields_ = [(n, fp) for n in (
        "attn_ln_w", "attn_ln_b", "q_w", "q_b", "k_w", "v_w", "v_b", "o_w", "o_b",
        "cross_ln_w", "cross_ln_b", "cq_w", "cq_b", "ck_w", "cv_w", "cv_b", "co_w", "co_b",
        "mlp_ln_w", "mlp_ln_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class OrcModel(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")] + [
        (n, fp) for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "enc_pos", "ln_post_w",
                          "ln_post_b", "tok_emb", "dec_pos", "dec_ln_w", "dec_ln_b")] + [
        ("enc", C.POINTER(OrcBlock)), ("dec", C.POINTER(OrcBlock))]


def build(ref: bool = True) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    targets = ["all"] + (["ref"] if ref and os.path.isdir("/root/reference/cpp/src/librosa") else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.orc_log_mel.restype = C.c_int
        L.orc_log_mel.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.orc_mel_filterbank.argtypes = [C.c_int, fp]
        L.orc_log_mel_openai.restype = None
        L.orc_log_mel_openai.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.orc_sinusoids.argtypes = [C.c_int, C.c_int, fp]
        L.orc_encoder.argtypes = [C.POINTER(OrcModel), C.POINTER(OrcPolicy), fp, fp, fp]
        L.orc_decoder_step.argtypes = [C.POINTER(OrcModel), C.POINTER(OrcPolicy), C.c_int, C.c_int,
                                       fp, fp, fp, fp, fp]
        L.orc_greedy.restype = C.c_int
        L.orc_greedy.argtypes = [C.POINTER(OrcModel), C.POINTER(OrcPolicy), fp, fp,
                                 C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                 C.POINTER(C.c_int), fp]
        L.orc_transcribe.restype = C.c_int
        L.orc_transcribe.argtypes = [C.POINTER(OrcModel), C.POINTER(OrcPolicy), fp, C.c_int,
                                     C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_argmax.restype = C.c_int
        L.orc_argmax.argtypes = [fp, C.c_int]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_clear_cache.argtypes = []
        # libgomp with hundreds of threads makes the many tiny parallel regions of the decoder crawl
        L.orc_set_threads(min(os.cpu_count() or 1, 16))
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own front-end (oracle/_ref), or None where it has not been built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_frontend.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_log_mel.restype = C.c_int
        L.ref_log_mel.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.ref_mel_filterbank.argtypes = [C.c_int, fp]
        _REF = L
    return _REF


def _p(a: np.ndarray):
    return a.ctypes.data_as(fp)


def log_mel(pcm: np.ndarray, n_mels: int = 80, use_ref: bool = False):
    """-> (mel [n_mels, 3000] f32, n_frames, mmax)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.float32)
    out = np.empty((n_mels, 3000), dtype=np.float32)
    mmax = C.c_float()
    L = ref_lib() if use_ref else lib()
    fn = L.ref_log_mel if use_ref else L.orc_log_mel
    n = fn(_p(pcm), len(pcm), n_mels, _p(out), C.byref(mmax))
    return out, n, mmax.value


def log_mel_openai(pcm: np.ndarray, n_mels: int = 80):
    """feature_mode "openai" (generate_data.py:162-176 lineage) -> (mel [n_mels, 3000] f32, mmax)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.float32)
    out = np.empty((n_mels, 3000), dtype=np.float32)
    mmax = C.c_float()
    lib().orc_log_mel_openai(_p(pcm), len(pcm), n_mels, _p(out), C.byref(mmax))
    return out, mmax.value


def mel_filterbank(n_mels: int, use_ref: bool = False) -> np.ndarray:
    out = np.empty((n_mels, 201), dtype=np.float32)
    (ref_lib().ref_mel_filterbank if use_ref else lib().orc_mel_filterbank)(n_mels, _p(out))
    return out


class Oracle:
    """CPU restatement bound to one set of weights (openai-whisper state_dict names, fp32)."""

    def __init__(self, config: dict, weights: dict, bf16_policy=False, threads: int = 0, literal_mask: bool = False):
        """bf16_policy: False/0 = pure fp32, True/1 = the engine's bfloat16 storage points, 2 or "fp16" = the same
        storage points in IEEE half (the engine's fp16 build)."""
        self.cfg = config
        self.w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
        # literal_mask: decoder self-attention as the exported graph states it (448-row cache, -60000 fill, separate
        # current-token column, 449-wide softmax: export_onnx.py:124-137) instead of the equivalent causal form
        self.policy = OrcPolicy(2 if bf16_policy in (2, "fp16") else (1 if bf16_policy else 0), 1 if literal_mask else 0)
        self.L = lib()
        if threads:
            self.L.orc_set_threads(threads)
        m = OrcModel()
        for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                  "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer"):
            setattr(m, k, int(config[k]))
        g = lambda name: _p(self.w[name])
        m.conv1_w, m.conv1_b = g("encoder.conv1.weight"), g("encoder.conv1.bias")
        m.conv2_w, m.conv2_b = g("encoder.conv2.weight"), g("encoder.conv2.bias")
        if "encoder.positional_embedding" not in self.w:
            pe = np.empty((m.n_audio_ctx, m.n_audio_state), dtype=np.float32)
            self.L.orc_sinusoids(m.n_audio_ctx, m.n_audio_state, _p(pe))
            self.w["encoder.positional_embedding"] = pe
        m.enc_pos = g("encoder.positional_embedding")
        m.ln_post_w, m.ln_post_b = g("encoder.ln_post.weight"), g("encoder.ln_post.bias")
        m.tok_emb, m.dec_pos = g("decoder.token_embedding.weight"), g("decoder.positional_embedding")
        m.dec_ln_w, m.dec_ln_b = g("decoder.ln.weight"), g("decoder.ln.bias")

        def blk(prefix, cross):
            b = OrcBlock()
            b.attn_ln_w, b.attn_ln_b = g(prefix + ".attn_ln.weight"), g(prefix + ".attn_ln.bias")
            b.q_w, b.q_b = g(prefix + ".attn.query.weight"), g(prefix + ".attn.query.bias")
            b.k_w = g(prefix + ".attn.key.weight")
            b.v_w, b.v_b = g(prefix + ".attn.value.weight"), g(prefix + ".attn.value.bias")
            b.o_w, b.o_b = g(prefix + ".attn.out.weight"), g(prefix + ".attn.out.bias")
            if cross:
                b.cross_ln_w, b.cross_ln_b = g(prefix + ".cross_attn_ln.weight"), g(prefix + ".cross_attn_ln.bias")
                b.cq_w, b.cq_b = g(prefix + ".cross_attn.query.weight"), g(prefix + ".cross_attn.query.bias")
                b.ck_w = g(prefix + ".cross_attn.key.weight")
                b.cv_w, b.cv_b = g(prefix + ".cross_attn.value.weight"), g(prefix + ".cross_attn.value.bias")
                b.co_w, b.co_b = g(prefix + ".cross_attn.out.weight"), g(prefix + ".cross_attn.out.bias")
            b.mlp_ln_w, b.mlp_ln_b = g(prefix + ".mlp_ln.weight"), g(prefix + ".mlp_ln.bias")
            b.fc1_w, b.fc1_b = g(prefix + ".mlp.0.weight"), g(prefix + ".mlp.0.bias")
            b.fc2_w, b.fc2_b = g(prefix + ".mlp.2.weight"), g(prefix + ".mlp.2.bias")
            return b

        self._enc = (OrcBlock * m.n_audio_layer)(*[blk(f"encoder.blocks.{i}", False) for i in range(m.n_audio_layer)])
        self._dec = (OrcBlock * m.n_text_layer)(*[blk(f"decoder.blocks.{i}", True) for i in range(m.n_text_layer)])
        m.enc, m.dec = self._enc, self._dec
        self.m = m

    def __del__(self):
        try:
            self.L.orc_clear_cache()
        except Exception:
            pass

    # -- stages -------------------------------------------------------------------------
    def sot_seq(self, language: str = "zh"):
        """[sot, lang, transcribe, no_timestamps] (Whisper.cpp:129-139, 241-251)."""
        codes = self.cfg["all_language_codes"].split(",")
        toks = [int(t) for t in self.cfg["all_language_tokens"].split(",")]
        if language not in codes:
            language = "zh"
        return [int(self.cfg["sot"]), toks[codes.index(language)], int(self.cfg["transcribe"]),
                int(self.cfg["no_timestamps"])]

    def encoder(self, mel: np.ndarray):
        m = self.m
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        ck = np.empty((m.n_text_layer, m.n_audio_ctx, m.n_text_state), dtype=np.float32)
        cv = np.empty_like(ck)
        self.L.orc_encoder(C.byref(m), C.byref(self.policy), _p(mel), _p(ck), _p(cv))
        return ck, cv

    def decoder_step(self, token, offset, ck, cv, self_k, self_v, want_logits=True):
        m = self.m
        logits = np.empty(m.n_vocab, dtype=np.float32) if want_logits else None
        self.L.orc_decoder_step(C.byref(m), C.byref(self.policy), int(token), int(offset), _p(ck), _p(cv),
                                _p(self_k), _p(self_v), _p(logits) if want_logits else None)
        return logits

    def new_self_cache(self):
        m = self.m
        return (np.zeros((m.n_text_layer, m.n_text_ctx, m.n_text_state), dtype=np.float32),
                np.zeros((m.n_text_layer, m.n_text_ctx, m.n_text_state), dtype=np.float32))

    def greedy(self, ck, cv, language="zh", max_new=444, forced=None, want_logits=False, eot=None):
        """eot: the id that ends the loop (Whisper.cpp:219); default the config's, -1 = never (calibration runs of
        tests/eot_case.py)."""
        m = self.m
        sot = (C.c_int * 4)(*self.sot_seq(language))
        out = (C.c_int * m.n_text_ctx)()
        nf = 0 if forced is None else len(forced)
        fa = (C.c_int * max(nf, 1))(*(forced if nf else [0]))
        n_steps = (nf if nf else max_new) + 1
        logits = np.zeros((n_steps, m.n_vocab), dtype=np.float32) if want_logits else None
        n = self.L.orc_greedy(C.byref(m), C.byref(self.policy), _p(ck), _p(cv), sot, int(self.cfg["eot"] if eot is None else eot),
                              int(max_new), fa if nf else None, nf, out, _p(logits) if want_logits else None)
        ids = [out[i] for i in range(n)]
        return (ids, logits[: n + 1]) if want_logits else ids

    def transcribe(self, pcm, language="zh", max_new=444):
        m = self.m
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        sot = (C.c_int * 4)(*self.sot_seq(language))
        out = (C.c_int * m.n_text_ctx)()
        n = self.L.orc_transcribe(C.byref(m), C.byref(self.policy), _p(pcm), len(pcm), sot,
                                  int(self.cfg["eot"]), int(max_new), out)
        return [out[i] for i in range(n)]
