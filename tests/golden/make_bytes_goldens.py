"""Fixtures for the two BYTE PATHS of the drop-in boundary, produced by the REFERENCE ITSELF compiled in this container
(oracle/_ref/libref_bytes.so = cpp/src/AudioFile.h + cpp/src/base64.cpp from /root/reference, oracle/Makefile target `ref`;
the nine lines of ax_whisper_api.cpp:105-113 and the loops of Whisper.cpp:115-127,224-229 are restated in
oracle/ref_bytes_driver.cpp because their files include the closed AXera headers):

  bytes_audio.npz   — small audio FILES (their bytes, written here with `struct`) in every container / sample format the
                      reference's AudioFile reads, and the mono f32 samples the reference hands to Whisper::run for them
  bytes_tokens.npz  — every line of multilingual.tiktoken (= the {type}-tokens.txt the exporter writes,
                      export_onnx.py:391-417) decoded by the reference's base64_decode, plus id sequences -> bytes

    python tests/golden/make_bytes_goldens.py          (needs /root/reference; the fixtures travel, the reference does not)
"""
import ctypes as C
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_bytes.so")


def ref_lib():
    L = C.CDLL(REF_SO)
    L.ref_load_audio_mono.restype = C.c_int
    L.ref_load_audio_mono.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    L.ref_base64_decode.restype = C.c_int
    L.ref_base64_decode.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    L.ref_load_tokens.restype = C.c_int
    L.ref_load_tokens.argtypes = [C.c_char_p]
    L.ref_token_entry.restype = C.c_int
    L.ref_token_entry.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.ref_detokenize.restype = C.c_int
    L.ref_detokenize.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
    return L


# ------------------------------------------------------------------------------------------------- audio files
def _ints(rng, n, ch, bits):
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    x = rng.integers(lo, hi + 1, size=(n, ch), dtype=np.int64)
    x[0], x[1], x[2], x[3] = lo, hi, 0, -1          # the extremes: -32768 / 32768 = -1.0, stereo (lo + hi) / 2
    if ch == 2:
        x[4] = (lo, lo)
        x[5] = (hi, hi)
        x[6] = (hi, lo)
    return x


def _pack_int(x, bits, big):
    n = bits // 8
    out = bytearray()
    for v in x.ravel():
        v = int(v)
        if bits == 8 and not big:
            out += struct.pack("B", v + 128)        # WAV 8-bit is unsigned
            continue
        out += int(v & ((1 << bits) - 1)).to_bytes(n, "big" if big else "little")
    return bytes(out)


def wav_bytes(x, bits, fmt=1, extensible=False, extra_chunk=False, rate=16000):
    n, ch = x.shape
    data = x.astype("<f4").tobytes() if fmt == 3 else _pack_int(x, bits, big=False)
    if extensible:
        sub = struct.pack("<H", fmt) + bytes.fromhex("000000001000800000aa00389b71")
        fmt_chunk = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits, 22, bits, (1 << ch) - 1) + sub
    else:
        fmt_chunk = struct.pack("<HHIIHH", fmt, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk
    if extra_chunk:   # an odd-sized LIST chunk (padded to even) in front of the data
        body += b"LIST" + struct.pack("<I", 5) + b"INFOx" + b"\0"
    body += b"data" + struct.pack("<I", len(data)) + data
    if len(data) & 1:
        body += b"\0"
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _ext80(rate):
    import math

    m, e = math.frexp(float(rate))
    return struct.pack(">HQ", e - 1 + 16383, int(m * (1 << 64)))


def aiff_bytes(x, bits, aifc=None, rate=16000):
    n, ch = x.shape
    if aifc == "fl32":
        data = x.astype(">f4").tobytes()
    else:
        data = _pack_int(x, bits, big=True)
    comm = struct.pack(">hIh", ch, n, bits) + _ext80(rate)
    if aifc:
        name = b"not compressed" if aifc == "NONE" else b"32-bit float"
        comm += aifc.encode() + bytes([len(name)]) + name + (b"\0" if (len(name) + 1) & 1 else b"")
    ssnd = struct.pack(">II", 0, 0) + data
    body = (b"AIFC" if aifc else b"AIFF")
    if aifc:
        body += b"FVER" + struct.pack(">II", 4, 0xA2805140)
    body += b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    if len(ssnd) & 1:
        body += b"\0"
    return b"FORM" + struct.pack(">I", len(body)) + body


def audio_cases():
    rng = np.random.Generator(np.random.PCG64(20260501))
    n = 700
    cases = {}
    for ch in (1, 2):
        for bits in (8, 16, 24, 32):
            cases[f"wav_int{bits}_{ch}ch"] = wav_bytes(_ints(rng, n, ch, bits), bits)
            cases[f"aiff_int{bits}_{ch}ch"] = aiff_bytes(_ints(rng, n, ch, bits), bits)
        f = np.clip(rng.standard_normal((n, ch)) * 0.4, -1.5, 1.5).astype(np.float32)
        f[0] = 1.0
        f[1] = -1.0
        cases[f"wav_f32_{ch}ch"] = wav_bytes(f, 32, fmt=3)
        cases[f"aifc_fl32_{ch}ch"] = aiff_bytes(f, 32, aifc="fl32")
    cases["wav_int16_3ch"] = wav_bytes(_ints(rng, n, 3, 16), 16)          # > 2 channels: channel 0 (ax_whisper_api.cpp:105)
    cases["wav_int16_6ch"] = wav_bytes(_ints(rng, n, 6, 16), 16)
    cases["wav_int16_2ch_extensible"] = wav_bytes(_ints(rng, n, 2, 16), 16, extensible=True)
    cases["wav_int24_1ch_extensible"] = wav_bytes(_ints(rng, n, 1, 24), 24, extensible=True)
    cases["wav_int16_1ch_list_chunk"] = wav_bytes(_ints(rng, n, 1, 16), 16, extra_chunk=True)
    cases["wav_int16_1ch_odd_frames"] = wav_bytes(_ints(rng, 333, 1, 8), 8)   # odd data size -> pad byte
    cases["wav_int16_1ch_8k"] = wav_bytes(_ints(rng, n, 1, 16), 16, rate=8000)  # not 16 kHz: no resampling anywhere
    cases["aifc_none_int16_2ch"] = aiff_bytes(_ints(rng, n, 2, 16), 16, aifc="NONE")
    cases["aiff_int16_1ch_44k"] = aiff_bytes(_ints(rng, n, 1, 16), 16, rate=44100)
    demo = open(os.path.join(HERE, "demo.wav"), "rb").read()
    assert demo[36:40] == b"data"   # the reference's own sample file (python/../demo.wav), its first 2000 samples
    head = demo[:40] + struct.pack("<I", 4000) + demo[44:4044]
    cases["demo_wav_head"] = head[:4] + struct.pack("<I", len(head) - 8) + head[8:]
    return cases


def make_audio(L):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, blob in audio_cases().items():
            path = os.path.join(td, name + (".wav" if name.startswith(("wav", "demo")) else ".aiff"))
            with open(path, "wb") as f:
                f.write(blob)
            info = (C.c_int * 3)()
            n = L.ref_load_audio_mono(path.encode(), None, 0, info)
            out[name + ".file"] = np.frombuffer(blob, dtype=np.uint8)
            if n < 0:
                out[name + ".ok"] = np.int32(0)
                print(f"{name:32s} reference: load fails")
                continue
            buf = np.empty(max(n, 1), dtype=np.float32)
            L.ref_load_audio_mono(path.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), n, info)
            out[name + ".ok"] = np.int32(1)
            out[name + ".samples"] = buf[:n].copy()
            out[name + ".info"] = np.array(list(info), dtype=np.int32)
            print(f"{name:32s} reference: {n} samples, rate {info[0]}, channels {info[1]}, bits {info[2]}, first {buf[:4]}")
    np.savez_compressed(os.path.join(HERE, "bytes_audio.npz"), **out)


# ------------------------------------------------------------------------------------------------- token table
def make_tokens(L):
    path = os.path.join(HERE, "multilingual.tiktoken")
    n = L.ref_load_tokens(path.encode())
    lens = np.zeros(n, dtype=np.int32)       # base64_decode's return value j: the decoded length
    cstr = []                                 # what its strcpy leaves in str: the bytes up to the first NUL
    ent = C.create_string_buffer(256)
    buf = C.create_string_buffer(1024)
    for i in range(n):
        m = L.ref_token_entry(i, ent, 256)
        if m & 3:   # the last line, "= 50256": base64.cpp:87 asserts (code_len & 3) == 0; the entry stands for no bytes
            lens[i] = -1
            cstr.append(b"")
            print(f"tokens: entry {i} = {ent.raw[:m]!r} is not a multiple of 4 characters: the reference asserts; expected bytes: none")
            continue
        lens[i] = L.ref_base64_decode(ent.raw[:m], m, buf)
        cstr.append(buf.value)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(b) for b in cstr])
    blob = np.frombuffer(b"".join(cstr), dtype=np.uint8)
    rng = np.random.Generator(np.random.PCG64(77))
    seqs = [rng.integers(0, n, size=k).astype(np.int32) for k in (1, 5, 40, 444)]
    seqs.append(np.array([50256, 0, 50257, 51864, 220, -1, 50256], dtype=np.int32))   # out-of-table ids are skipped (fix of B8)
    seqs.append(np.arange(0, n, 97, dtype=np.int32))
    out = {"n": np.int32(n), "lens": lens, "offs": offs, "blob": blob, "n_seq": np.int32(len(seqs))}
    for k, s in enumerate(seqs):
        b = C.create_string_buffer(1 << 16)
        m = L.ref_detokenize(s.ctypes.data_as(C.POINTER(C.c_int)), len(s), b, 1 << 16)
        out[f"seq{k}.ids"] = s
        out[f"seq{k}.bytes"] = np.frombuffer(b.raw[:m], dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "bytes_tokens.npz"), **out)
    longest = int(lens.max())
    nul = [i for i in range(n) if len(cstr[i]) != lens[i]]
    print(f"tokens: {n} entries, longest {longest} bytes (ids {np.nonzero(lens == longest)[0].tolist()}), entries holding a NUL: {nul}")


if __name__ == "__main__":
    if not os.path.exists(REF_SO):
        sys.exit("oracle/_ref/libref_bytes.so is missing: `make -C oracle ref` in the container that has /root/reference")
    L = ref_lib()
    make_audio(L)
    make_tokens(L)
