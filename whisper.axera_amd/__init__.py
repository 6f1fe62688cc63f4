"""Python host mirror of the whisper.axera C ABI on MI355X (ctypes over libax_whisper.so).

The reference ships a Python twin of its C++ pipeline (python/whisper.py: ``Whisper(model_type,
model_path, language, task).run(audio)``); this module keeps that shape on top of the HIP
library. There is no CPU fallback: if the library or a GPU is missing, construction fails loudly.

The directory name contains a dot, so import it through the repo-root shim::

    import whisper_axera_amd as wa
    w = wa.Whisper("small", "/path/to/models", "zh")
    ids = w.run_tokens(pcm)
"""
from __future__ import annotations

import ctypes as C
import os
import time
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libax_whisper.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ax_whisper_api.h")

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
_lib = None

# name -> (restype, argtypes): every symbol include/ax_whisper_api.h declares
SYMBOLS = {
    "AX_WHISPER_Init": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p]),
    "AX_WHISPER_InitEx": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "AX_WHISPER_InitMulti": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_int]),
    "AX_WHISPER_GetDeviceCount": (C.c_int, [C.c_void_p]),
    "AX_WHISPER_VisibleDeviceCount": (C.c_int, []),
    "AX_WHISPER_Uninit": (None, [C.c_void_p]),
    "AX_WHISPER_RunFile": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "AX_WHISPER_RunPCM": (C.c_int, [C.c_void_p, fp, C.c_int, C.POINTER(C.c_void_p)]),
    "AX_WHISPER_GetConfigInt": (C.c_int, [C.c_void_p, C.c_char_p]),
    "AX_WHISPER_LastError": (C.c_char_p, [C.c_void_p]),
    "AX_WHISPER_SetStream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "AX_WHISPER_RunPCMBatchTokens": (C.c_int, [C.c_void_p, C.POINTER(fp), C.POINTER(C.c_int), C.c_int, C.c_int, ip, C.POINTER(C.c_int)]),
    "AX_WHISPER_RunPCMBatch": (C.c_int, [C.c_void_p, C.POINTER(fp), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "AX_WHISPER_RunDeviceBatchTokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, ip, C.POINTER(C.c_int)]),
    "AX_WHISPER_RunDeviceBatchTokensRagged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int), ip, C.POINTER(C.c_int)]),
    "AX_WHISPER_Detokenize": (C.c_int, [C.c_void_p, ip, C.c_int, C.POINTER(C.c_void_p)]),
    "AX_WHISPER_Transcript": (C.c_int, [C.c_void_p, ip, C.c_int, C.POINTER(C.c_void_p)]),
    "AX_WHISPER_ConvertT2S": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "AX_WHISPER_LoadAudioFile": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "AX_WHISPER_DetokenizeWithTable": (C.c_int, [C.c_char_p, ip, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "AX_WHISPER_ComputeMel": (C.c_int, [C.c_void_p, fp, C.c_int, fp]),
    "AX_WHISPER_EncodeMel": (C.c_int, [C.c_void_p, fp, C.c_int]),
    "AX_WHISPER_GetCrossKV": (C.c_int, [C.c_void_p, C.c_int, fp, fp]),
    "AX_WHISPER_ScanStored16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int64), fp, C.POINTER(C.c_int)]),
    "AX_WHISPER_DecodeForced": (C.c_int, [C.c_void_p, C.c_int, ip, C.c_int, fp, ip]),
    "AX_WHISPER_DecodeGreedy": (C.c_int, [C.c_void_p, C.c_int, C.c_int, ip, C.POINTER(C.c_int)]),
    "AX_WHISPER_DecodeGreedyRagged": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), ip, C.POINTER(C.c_int)]),
    "AX_WHISPER_StreamOpen": (C.c_int, [C.c_void_p, C.c_int]),
    "AX_WHISPER_StreamAdmit": (C.c_int, [C.c_void_p, C.c_int, fp, C.c_int, C.c_int]),
    "AX_WHISPER_StreamAdmitBatch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(fp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "AX_WHISPER_StreamStep": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "AX_WHISPER_StreamCollect": (C.c_int, [C.c_void_p, C.c_int, ip, C.POINTER(C.c_int)]),
    "AX_WHISPER_StreamClose": (C.c_int, [C.c_void_p]),
    "AX_WHISPER_GetTimings": (C.c_int, [C.c_void_p, fp]),
    "AX_WHISPER_Bench": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, fp]),
}


def build(verbose: bool = False) -> str:
    """Compile libax_whisper.so, whisper_cli and whisper_srv for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _HERE, "-j8"], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libax_whisper.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


def load_library():
    """dlopen the HIP library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (the engine has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        L._free = C.CDLL(None).free
        L._free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def convert_t2s(config_path: str, text: str) -> str:
    """The reference's zh post-pass on its own (cpp/src/Whisper.cpp:231-236): OpenCC t2s.json + its .ocd2 dictionaries."""
    L = load_library()
    out = C.c_void_p()
    if L.AX_WHISPER_ConvertT2S(os.fspath(config_path).encode(), text.encode("utf-8"), C.byref(out)) != 0:
        raise RuntimeError("AX_WHISPER_ConvertT2S failed: " + (L.AX_WHISPER_LastError(None) or b"").decode())
    s = C.string_at(out.value).decode("utf-8")
    L._free(out.value)
    return s


def load_audio_file(path: str):
    """The file decode of AX_WHISPER_RunFile on its own (host only) -> (mono f32 samples, sample rate, channels)."""
    L = load_library()
    out, n, info = C.c_void_p(), C.c_int(), (C.c_int * 2)()
    if L.AX_WHISPER_LoadAudioFile(os.fspath(path).encode(), C.byref(out), C.byref(n), info) != 0:
        raise RuntimeError("AX_WHISPER_LoadAudioFile failed: " + (L.AX_WHISPER_LastError(None) or b"").decode())
    a = np.ctypeslib.as_array(C.cast(out, fp), shape=(max(n.value, 1),))[: n.value].copy()
    L._free(out.value)
    return a, info[0], info[1]


def detokenize_with_table(tokens_path: str, ids) -> bytes:
    """ids -> bytes through a tokens file alone (host only): what Whisper.detokenize returns for them."""
    L = load_library()
    a = np.ascontiguousarray(ids, dtype=np.int32)
    out, n = C.c_void_p(), C.c_int()
    if L.AX_WHISPER_DetokenizeWithTable(os.fspath(tokens_path).encode(), a.ctypes.data_as(ip), len(a), C.byref(out), C.byref(n)) != 0:
        raise RuntimeError("AX_WHISPER_DetokenizeWithTable failed: " + (L.AX_WHISPER_LastError(None) or b"").decode())
    b = C.string_at(out.value, n.value)
    L._free(out.value)
    return b


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Whisper:
    """Mirror of the reference's ``Whisper`` (python/whisper.py:54-99, cpp/src/Whisper.hpp:28-59)."""

    def __init__(self, model_type: str, model_path: str, language: str = "zh", device: int = -1, max_batch: int = 0, devices=None):
        """devices: a list of HIP ordinals (or "all") -> one engine per device behind this handle (AX_WHISPER_InitMulti)."""
        self.L = load_library()
        if devices is not None:
            if devices == "all":
                self.h = self.L.AX_WHISPER_InitMulti(model_type.encode(), model_path.encode(), language.encode(), None, 0, max_batch)
            else:
                arr = (C.c_int * len(devices))(*[int(d) for d in devices])
                self.h = self.L.AX_WHISPER_InitMulti(model_type.encode(), model_path.encode(), language.encode(), arr, len(devices), max_batch)
        else:
            self.h = self.L.AX_WHISPER_InitEx(model_type.encode(), model_path.encode(), language.encode(), device, max_batch)
        if not self.h:
            raise RuntimeError("AX_WHISPER_Init failed: " + (self.L.AX_WHISPER_LastError(None) or b"").decode())
        g = lambda k: self.L.AX_WHISPER_GetConfigInt(self.h, k.encode())
        self.n_mels, self.n_vocab, self.n_text_ctx = g("n_mels"), g("n_vocab"), g("n_text_ctx")
        self.n_text_layer, self.n_text_state, self.eot = g("n_text_layer"), g("n_text_state"), g("eot")
        self.sot_seq = [g(f"sot_seq{i}") for i in range(4)]
        self.n_devices = self.L.AX_WHISPER_GetDeviceCount(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.AX_WHISPER_Uninit(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: " + (self.L.AX_WHISPER_LastError(self.h) or b"").decode())

    def _take(self, p):
        s = C.string_at(p).decode("utf-8", errors="replace") if p else ""
        if p:
            self.L._free(p)
        return s

    # ---- legacy entry points ------------------------------------------------------------
    def run(self, audio) -> str:
        """PCM (16 kHz mono f32) or a wav path -> text (python/whisper.py:213-271)."""
        out = C.c_void_p()
        if isinstance(audio, (str, os.PathLike)):
            self._check(self.L.AX_WHISPER_RunFile(self.h, os.fspath(audio).encode(), C.byref(out)), "AX_WHISPER_RunFile")
        else:
            a = _f32(audio)
            self._check(self.L.AX_WHISPER_RunPCM(self.h, a.ctypes.data_as(fp), len(a), C.byref(out)), "AX_WHISPER_RunPCM")
        return self._take(out.value)

    # ---- additions ------------------------------------------------------------------------
    def run_tokens_batch(self, clips, max_new: int = 0):
        clips = [_f32(c) for c in clips]
        B = len(clips)
        ptrs = (fp * B)(*[c.ctypes.data_as(fp) for c in clips])
        lens = (C.c_int * B)(*[len(c) for c in clips])
        ids = np.zeros((B, self.n_text_ctx), dtype=np.int32)
        n = (C.c_int * B)()
        self._check(self.L.AX_WHISPER_RunPCMBatchTokens(self.h, ptrs, lens, B, max_new, ids.ctypes.data_as(ip), n), "RunPCMBatchTokens")
        return [ids[b, : n[b]].tolist() for b in range(B)]

    def run_tokens(self, pcm, max_new: int = 0):
        return self.run_tokens_batch([pcm], max_new)[0]

    def run_batch(self, clips):
        clips = [_f32(c) for c in clips]
        B = len(clips)
        ptrs = (fp * B)(*[c.ctypes.data_as(fp) for c in clips])
        lens = (C.c_int * B)(*[len(c) for c in clips])
        outs = (C.c_void_p * B)()
        self._check(self.L.AX_WHISPER_RunPCMBatch(self.h, ptrs, lens, B, outs), "RunPCMBatch")
        return [self._take(outs[b]) for b in range(B)]

    def run_device_tokens(self, d_ptr: int, stride: int, n_samples, max_new: int = 0, max_new_clip=None):
        """d_ptr: device address of [B][stride] f32 PCM already resident in HBM; max_new_clip: per-clip id budgets."""
        B = len(n_samples)
        lens = (C.c_int * B)(*[int(x) for x in n_samples])
        ids = np.zeros((B, self.n_text_ctx), dtype=np.int32)
        n = (C.c_int * B)()
        if max_new_clip is not None:
            mc = (C.c_int * B)(*[int(x) for x in max_new_clip])
            self._check(self.L.AX_WHISPER_RunDeviceBatchTokensRagged(self.h, C.c_void_p(d_ptr), stride, lens, B, max_new, mc, ids.ctypes.data_as(ip), n), "RunDeviceBatchTokensRagged")
            return [ids[b, : n[b]].tolist() for b in range(B)]
        self._check(self.L.AX_WHISPER_RunDeviceBatchTokens(self.h, C.c_void_p(d_ptr), stride, lens, B, max_new, ids.ctypes.data_as(ip), n), "RunDeviceBatchTokens")
        return [ids[b, : n[b]].tolist() for b in range(B)]

    def detokenize(self, ids) -> bytes:
        a = np.ascontiguousarray(ids, dtype=np.int32)
        out = C.c_void_p()
        self._check(self.L.AX_WHISPER_Detokenize(self.h, a.ctypes.data_as(ip), len(a), C.byref(out)), "Detokenize")
        b = C.string_at(out.value) if out.value else b""
        if out.value:
            self.L._free(out.value)
        return b

    def transcript(self, ids) -> str:
        a = np.ascontiguousarray(ids, dtype=np.int32)
        out = C.c_void_p()
        self._check(self.L.AX_WHISPER_Transcript(self.h, a.ctypes.data_as(ip), len(a), C.byref(out)), "Transcript")
        return self._take(out.value)

    def set_stream(self, stream_ptr: int):
        self._check(self.L.AX_WHISPER_SetStream(self.h, C.c_void_p(stream_ptr)), "SetStream")

    def compute_mel(self, pcm) -> np.ndarray:
        a = _f32(pcm)
        out = np.empty((self.n_mels, 3000), dtype=np.float32)
        self._check(self.L.AX_WHISPER_ComputeMel(self.h, a.ctypes.data_as(fp), len(a), out.ctypes.data_as(fp)), "ComputeMel")
        return out

    def encode_mel(self, mel):
        m = _f32(mel)
        if m.ndim == 2:
            m = m[None]
        self._check(self.L.AX_WHISPER_EncodeMel(self.h, m.ctypes.data_as(fp), m.shape[0]), "EncodeMel")
        return m.shape[0]

    def get_cross_kv(self, slot: int = 0):
        shape = (self.n_text_layer, 1500, self.n_text_state)
        k, v = np.empty(shape, dtype=np.float32), np.empty(shape, dtype=np.float32)
        self._check(self.L.AX_WHISPER_GetCrossKV(self.h, slot, k.ctypes.data_as(fp), v.ctypes.data_as(fp)), "GetCrossKV")
        return k, v

    def scan_stored16(self, batch: int = 1):
        """{buffer name: (non-finite elements, max finite |x|)} over every 16-bit tensor the engine stores between kernels."""
        n_max = 32
        names = C.create_string_buffer(32 * n_max)
        bad = (C.c_int64 * n_max)()
        mx = (C.c_float * n_max)()
        n = C.c_int()
        self._check(self.L.AX_WHISPER_ScanStored16(self.h, batch, n_max, names, bad, mx, C.byref(n)), "ScanStored16")
        return {names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode(): (int(bad[i]), float(mx[i])) for i in range(n.value)}

    def decode_forced(self, batch: int, forced, want_logits: bool = True):
        f = np.ascontiguousarray(forced, dtype=np.int32).reshape(batch, -1)
        n = f.shape[1]
        logits = np.empty((batch, n + 1, self.n_vocab), dtype=np.float32) if want_logits else None
        am = np.empty((batch, n + 1), dtype=np.int32)
        self._check(self.L.AX_WHISPER_DecodeForced(self.h, batch, f.ctypes.data_as(ip), n,
                                                   logits.ctypes.data_as(fp) if want_logits else None, am.ctypes.data_as(ip)), "DecodeForced")
        return logits, am

    def decode_greedy(self, batch: int, max_new: int = 0, max_new_clip=None):
        ids = np.zeros((batch, self.n_text_ctx), dtype=np.int32)
        n = (C.c_int * batch)()
        if max_new_clip is None:
            self._check(self.L.AX_WHISPER_DecodeGreedy(self.h, batch, max_new, ids.ctypes.data_as(ip), n), "DecodeGreedy")
        else:
            mc = (C.c_int * batch)(*[int(x) for x in max_new_clip])
            self._check(self.L.AX_WHISPER_DecodeGreedyRagged(self.h, batch, max_new, mc, ids.ctypes.data_as(ip), n), "DecodeGreedyRagged")
        return [ids[b, : n[b]].tolist() for b in range(batch)]

    # ---- utterance slots refilled while the others decode (AX_WHISPER_Stream*)
    def stream_open(self, n_slots: int):
        self._check(self.L.AX_WHISPER_StreamOpen(self.h, n_slots), "StreamOpen")
        self._n_slots = n_slots

    def stream_admit(self, slot: int, pcm, max_new: int = 0):
        a = _f32(pcm)
        self._check(self.L.AX_WHISPER_StreamAdmit(self.h, slot, a.ctypes.data_as(fp), len(a), max_new), "StreamAdmit")

    def stream_admit_batch(self, slots, clips, max_new=None):
        clips = [_f32(c) for c in clips]
        n = len(clips)
        sl = (C.c_int * n)(*[int(x) for x in slots])
        ptrs = (fp * n)(*[c.ctypes.data_as(fp) for c in clips])
        lens = (C.c_int * n)(*[len(c) for c in clips])
        mn = (C.c_int * n)(*[int(x) for x in max_new]) if max_new is not None else None
        self._check(self.L.AX_WHISPER_StreamAdmitBatch(self.h, sl, ptrs, lens, mn, n), "StreamAdmitBatch")

    def stream_step(self, n_steps: int = 8):
        fin = (C.c_int * max(self._n_slots, 3))()
        n = C.c_int()
        self._check(self.L.AX_WHISPER_StreamStep(self.h, n_steps, fin, C.byref(n)), "StreamStep")
        return [fin[i] for i in range(n.value)]

    def stream_collect(self, slot: int):
        ids = np.zeros(self.n_text_ctx, dtype=np.int32)
        n = C.c_int()
        self._check(self.L.AX_WHISPER_StreamCollect(self.h, slot, ids.ctypes.data_as(ip), C.byref(n)), "StreamCollect")
        return ids[: n.value].tolist()

    def stream_close(self):
        self._check(self.L.AX_WHISPER_StreamClose(self.h), "StreamClose")

    def run_stream(self, clips, n_slots: int, max_new=0, steps_per_call: int = 8, min_admit: int = 1, stamps=None):
        """Feed `clips` through n_slots refillable slots in arrival order; max_new: one budget or one per clip.
        min_admit: free slots to wait for before an admission pass (a larger encoder batch per pass; the last clips are
        admitted as they come). stamps: a list that receives time.perf_counter() of every collected clip, in completion order.
        Returns (ids per clip, decoder-step calls made)."""
        budgets = list(max_new) if hasattr(max_new, "__len__") else [max_new] * len(clips)
        self.stream_open(n_slots)
        try:
            out = [None] * len(clips)
            owner = {}
            free = list(range(n_slots))
            nxt = calls = 0
            while nxt < len(clips) or owner:
                k = min(len(free), len(clips) - nxt)
                if k < min(min_admit, len(clips) - nxt) and owner:
                    k = 0  # wait for more slots to free up (something is still decoding)
                if k == 1:
                    self.stream_admit(free[0], clips[nxt], budgets[nxt])
                elif k > 1:  # every free slot is refilled by ONE batched front-end + encoder pass
                    self.stream_admit_batch(free[:k], clips[nxt:nxt + k], budgets[nxt:nxt + k])
                for i in range(k):
                    owner[free.pop(0)] = nxt
                    nxt += 1
                calls += 1
                for sl in self.stream_step(steps_per_call):
                    if sl in owner:
                        out[owner.pop(sl)] = self.stream_collect(sl)
                        free.append(sl)
                        if stamps is not None:
                            stamps.append(time.perf_counter())
            return out, calls
        finally:
            self.stream_close()

    def timings(self):
        t = (C.c_float * 5)()
        self._check(self.L.AX_WHISPER_GetTimings(self.h, t), "GetTimings")
        return dict(frontend_ms=t[0], encoder_ms=t[1], decode_ms=t[2], wall_ms=t[3], steps=int(t[4]))

    def bench(self, what: str, batch: int = 1, arg: int = 0, iters: int = 10) -> float:
        ms = C.c_float()
        self._check(self.L.AX_WHISPER_Bench(self.h, what.encode(), batch, arg, iters, C.byref(ms)), "Bench")
        return ms.value
