"""The two BYTE PATHS of the drop-in boundary, bit-exact against the REFERENCE ITSELF.

tests/golden/bytes_audio.npz and bytes_tokens.npz were produced by the reference's own AudioFile.h and base64.cpp compiled
here (oracle/_ref/libref_bytes.so, oracle/ref_bytes_driver.cpp; generator: tests/golden/make_bytes_goldens.py):

  * audio file -> the mono f32 samples AX_WHISPER_RunFile feeds the engine (cpp/src/AudioFile.h:450-501 container dispatch,
    :1241-1243 int16 / 32768 and its 8 / 24 / 32-bit and float siblings; ax_whisper_api.cpp:105-113 channel 0 + stereo average):
    29 files — WAV and AIFF / AIFF-C, 8 / 16 / 24 / 32-bit integers and IEEE float, mono / stereo / 3 / 6 channels,
    WAVE_FORMAT_EXTENSIBLE, odd data sizes, other sample rates, the head of the reference's own demo.wav;
  * token ids -> bytes (Whisper.cpp:115-127 table load, :224-229 + base64.cpp:84-120): all 50 257 lines of
    multilingual.tiktoken, and id sequences (ids outside the table are skipped: the reference indexes out of bounds, B8).
Integer / byte work: the bar is BIT-EXACT, no tolerance anywhere in this file.

CPU (-m "not gpu"): the host-only entry points AX_WHISPER_LoadAudioFile / AX_WHISPER_DetokenizeWithTable against the
fixtures; where oracle/_ref exists (the build container) the fixtures are re-derived from the reference build and must not
have moved. GPU: AX_WHISPER_RunFile == AX_WHISPER_RunPCM on the fixture's samples, AX_WHISPER_Detokenize per id.

Where this library deliberately differs from the reference (all three are reference defects, none reachable by a well-formed
request): a chunk of ODD size in front of the WAV data — the reference's chunk walk does not skip the pad byte and fails to
load, this reader loads it; the last table line `= 50256` — base64.cpp:87 asserts on it, here it decodes to nothing; the one
33-byte token overflows the reference's char str[32] (Whisper.cpp:226), here it is just 33 bytes (the fixture was produced
with a wide buffer)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

TOK = os.path.join(GOLDEN, "multilingual.tiktoken")


def _audio():
    g = np.load(os.path.join(GOLDEN, "bytes_audio.npz"))
    return g, sorted({k.split(".")[0] for k in g.files})


def _write(g, name, td):
    p = os.path.join(str(td), name + (".wav" if name.startswith(("wav", "demo")) else ".aiff"))
    with open(p, "wb") as f:
        f.write(g[name + ".file"].tobytes())
    return p


def test_audio_files_decode_bit_equal_to_the_reference(built_lib, tmp_path):
    g, names = _audio()
    assert len(names) >= 29
    n_ok = 0
    for name in names:
        path = _write(g, name, tmp_path)
        if not int(g[name + ".ok"]):   # the reference refuses the file (its chunk walk trips over an odd-sized chunk): ours may load it
            assert name == "wav_int16_1ch_list_chunk"
            a, sr, ch = built_lib.load_audio_file(path)
            assert len(a) == 700 and (sr, ch) == (16000, 1)
            continue
        a, sr, ch = built_lib.load_audio_file(path)
        want, info = g[name + ".samples"], g[name + ".info"]
        assert (sr, ch) == (int(info[0]), int(info[1])), name
        assert a.dtype == np.float32 and a.shape == want.shape, (name, a.shape, want.shape)
        assert np.array_equal(a.view(np.uint32), want.view(np.uint32)), (name, float(np.abs(a - want).max()))
        n_ok += 1
    assert n_ok >= 28
    with pytest.raises(RuntimeError):
        built_lib.load_audio_file(os.path.join(str(tmp_path), "missing.wav"))


def test_token_table_decodes_bit_equal_to_the_reference(built_lib):
    t = np.load(os.path.join(GOLDEN, "bytes_tokens.npz"))
    n, offs, blob = int(t["n"]), t["offs"], t["blob"].tobytes()
    assert n == 50257 and int(t["lens"].max()) == 33
    # every entry, in order, as one string: equal to the concatenation of the reference's per-entry decodes
    assert built_lib.detokenize_with_table(TOK, np.arange(n)) == blob
    # entry by entry where it matters: the NUL token (the reference's strcpy leaves nothing), the 33-byte token, the last
    # line, and a seeded sample of the rest
    rng = np.random.Generator(np.random.PCG64(5))
    for i in [0, 188, 38538, 50255, 50256] + [int(x) for x in rng.integers(0, n, 200)]:
        assert built_lib.detokenize_with_table(TOK, [i]) == blob[int(offs[i]):int(offs[i + 1])], i
    assert built_lib.detokenize_with_table(TOK, [188]) == b"" and built_lib.detokenize_with_table(TOK, [50256]) == b""
    for k in range(int(t["n_seq"])):
        assert built_lib.detokenize_with_table(TOK, t[f"seq{k}.ids"]) == t[f"seq{k}.bytes"].tobytes(), k
    # independent of the reference: python's own base64 on the same lines
    import base64

    lines = open(TOK, "rb").read().split(b"\n")
    for i in (1, 1000, 38538, 50000):
        assert base64.b64decode(lines[i].split(b" ")[0]) == blob[int(offs[i]):int(offs[i + 1])]


def test_fixtures_are_what_the_reference_build_produces():
    """Only where the reference has been compiled (oracle/_ref: the build container): re-derive and compare."""
    import make_bytes_goldens as mk

    if not os.path.exists(mk.REF_SO):
        pytest.skip("oracle/_ref/libref_bytes.so is not built here (no /root/reference on this machine)")
    import ctypes as C
    import tempfile

    L = mk.ref_lib()
    g, names = _audio()
    cases = mk.audio_cases()
    assert sorted(cases) == names
    with tempfile.TemporaryDirectory() as td:
        for name in names:
            assert cases[name] == g[name + ".file"].tobytes(), name   # the generator still writes the same files
            path = _write(g, name, td)
            info = (C.c_int * 3)()
            n = L.ref_load_audio_mono(path.encode(), None, 0, info)
            assert (n >= 0) == bool(int(g[name + ".ok"])), name
            if n < 0:
                continue
            buf = np.empty(n, dtype=np.float32)
            L.ref_load_audio_mono(path.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), n, info)
            assert np.array_equal(buf.view(np.uint32), g[name + ".samples"].view(np.uint32)), name
    t = np.load(os.path.join(GOLDEN, "bytes_tokens.npz"))
    assert L.ref_load_tokens(TOK.encode()) == int(t["n"])
    ent, buf = C.create_string_buffer(256), C.create_string_buffer(1024)
    blob, offs = t["blob"].tobytes(), t["offs"]
    for i in range(0, int(t["n"]) - 1, 7):
        m = L.ref_token_entry(i, ent, 256)
        assert L.ref_base64_decode(ent.raw[:m], m, buf) == int(t["lens"][i])
        assert buf.value == blob[int(offs[i]):int(offs[i + 1])], i


@pytest.mark.gpu
def test_run_file_feeds_the_reference_samples_and_detokenize_per_id(built_lib, micro_case, tmp_path):
    """Through a handle: RunFile(file) must transcribe exactly what RunPCM(reference-decoded samples) does, and
    AX_WHISPER_Detokenize must return the reference's bytes for every id of the table."""
    import torch  # noqa: F401

    g, _ = _audio()
    e = built_lib.Whisper("micro", micro_case.root, "en", device=0)
    try:
        for name in ("wav_int24_2ch", "aiff_int16_2ch", "wav_int16_6ch", "wav_f32_1ch", "aifc_fl32_2ch", "demo_wav_head"):
            path = _write(g, name, tmp_path)
            pcm = g[name + ".samples"]
            a, _, _ = built_lib.load_audio_file(path)
            assert np.array_equal(a, pcm)
            assert e.run(path) == e.run(pcm), name
            mel_file = e.compute_mel(a)
            assert np.array_equal(mel_file, e.compute_mel(pcm))
        t = np.load(os.path.join(GOLDEN, "bytes_tokens.npz"))
        n, offs, blob = int(t["n"]), t["offs"], t["blob"].tobytes()
        for i in range(n):
            assert e.detokenize([i]) == blob[int(offs[i]):int(offs[i + 1])], i
        assert e.detokenize(np.arange(n)) == blob
        for k in range(int(t["n_seq"])):
            assert e.detokenize(t[f"seq{k}.ids"]) == t[f"seq{k}.bytes"].tobytes(), k
    finally:
        e.close()
