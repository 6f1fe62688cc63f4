"""GPU: other model shapes (mini: d=256/4 heads/3 decoder layers; tiny dims: d=384/6 heads/4 layers) against the
oracle and the transformers goldens; the CLI's stdout contract."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ModelCase, assert_ids_equal_or_tie, load_demo_pcm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_type,golden,seed", [("mini", "mini_synth", 13), ("tiny", "tiny_demo", 14), ("small", "small_demo", 0),
                                                    ("miniturbo", "miniturbo_synth", 21)])
def test_model_vs_oracle_and_transformers(built_lib, oracle_mod, tmp_path, model_type, golden, seed):
    from make_model_goldens_inputs import golden_mel

    case = ModelCase(tmp_path, model_type, seed)
    g = np.load(os.path.join(GOLDEN, f"model_{golden}.npz"))
    mel = golden_mel(golden)
    codes, toks = case.cfg["all_language_codes"].split(","), [int(t) for t in case.cfg["all_language_tokens"].split(",")]
    lang = codes[toks.index(int(g["sot_seq"][1]))]  # the language the golden was generated with (miniturbo: yue)
    e = built_lib.Whisper(model_type, case.root, lang, device=0)
    assert e.sot_seq == [int(x) for x in g["sot_seq"]]
    e.encode_mel(mel)
    k, v = e.get_cross_kv(0)
    kb, vb = case.oracle_bf16.encoder(mel)
    print(model_type, "cross kv err vs bf16 policy", np.abs(k - kb).max(), np.abs(v - vb).max())
    assert np.abs(k - kb).max() < 2e-2 and np.abs(v - vb).max() < 2e-2  # measured: exactly one bf16 ulp at |x| ~ 2 (1.56e-2)
    e_hf = np.abs(k[:, ::53, ::7] - g["cross_k_sub"]).max()
    print(model_type, "cross k err vs transformers golden (fp32)", e_hf)
    assert e_hf < 3e-2  # HF fp32 golden: half a bf16 ulp of storage (7.8e-3 at |x| ~ 2, 1.6e-2 at ~ 4) + the arithmetic
    n_new = int(g["n_new"])
    hf_ids = [int(x) for x in g["ids"][:n_new]]
    logits, am = e.decode_forced(1, np.array([hf_ids]))
    top = np.take_along_axis(logits[0], g["top_ids"][: n_new + 1], axis=1)
    err = np.abs(top - g["top_vals"][: n_new + 1]).max()
    print(model_type, "logits err vs transformers golden", err)
    assert err < 5e-3  # measured 6.3e-4 (tiny) .. 1.1e-3 (small)
    ids_o, lg_o = case.oracle_bf16.greedy(kb, vb, lang, max_new=n_new, forced=hf_ids, want_logits=True)
    e_o = np.abs(logits[0] - lg_o).max()
    print(model_type, "logits err vs bf16-policy oracle", e_o)
    assert e_o < 6e-3  # measured 1.8e-3 at Whisper-small dims, less below
    e.close()


def test_cli_stdout_contract(built_lib, micro_case):
    """whisper_cli.cpp:63-66,90,102-103: the printed lines and the RTF definition."""
    cli = os.path.join(os.path.dirname(built_lib.LIB_PATH), "whisper_cli")
    wav = os.path.join(GOLDEN, "demo.wav")
    r = subprocess.run([cli, "-w", wav, "-t", "micro", "-p", micro_case.root, "--language", "zh"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0] == f"wav_file: {wav}" and out[1] == f"model_path: {micro_case.root}"
    assert out[2] == "model_type: micro" and out[3] == "language: zh"
    assert out[4].startswith("Init whisper success, take ") and out[4].endswith("seconds")
    assert out[5].startswith("Result: ") and out[-1].startswith("RTF: ")
    assert float(out[-1].split()[1]) > 0
    r = subprocess.run([cli, "-t", "micro"], capture_output=True, text=True)
    assert r.returncode != 0 and "need option: --wav" in r.stderr


def test_turbo_shaped_model(built_lib, oracle_mod, tmp_path):
    """128 mel bins, the 100-language token layout (n_vocab 51866) and enc_layers != dec_layers (large-v3-turbo's shape,
    BASELINE configs[3]) at reduced width: front-end with 128 mels vs the reference golden, end-to-end ids vs the oracle."""
    from conftest import load_demo_pcm

    case = ModelCase(tmp_path, "miniturbo", 21)
    e = built_lib.Whisper("miniturbo", case.root, "yue", device=0, max_batch=6)
    assert e.sot_seq == [50258, 50358, 50360, 50364]  # yue is the 100th language; transcribe/notimestamps shift by one
    pcm = load_demo_pcm()
    g = np.load(os.path.join(GOLDEN, "frontend_demo_128.npz"))
    mel = e.compute_mel(pcm)
    assert np.abs(mel[:, : int(g["n_frames"])] - g["mel_real"]).max() < 2e-4
    kb, vb = case.oracle_bf16.encoder(mel)
    ids, lg = case.oracle_bf16.greedy(kb, vb, "yue", max_new=10, want_logits=True)
    got = e.run_tokens(pcm, max_new=10)
    e.encode_mel(mel)
    logits, am = e.decode_forced(1, np.array([ids]))
    err = np.abs(logits[0] - lg).max()
    print("miniturbo logits err", err)
    assert err < 1.2e-3  # measured 2.4e-4
    assert_ids_equal_or_tie(e, mel, got, ids, lg, "miniturbo")
    import modelgen

    clips = [pcm] + [modelgen.synth_clip(i, 200000) for i in range(1, 6)]  # 6 clips: the batched MFMA decode path
    batch_ids = e.run_tokens_batch(clips, max_new=6)
    assert all(len(x) == 6 for x in batch_ids)
    # clip 0 inside the 6-clip MFMA batch against the same oracle run (its first 6 ids)
    assert_ids_equal_or_tie(e, mel, batch_ids[0], ids[:6], lg[:7], "miniturbo, batched")
    e.close()


@pytest.mark.parametrize("scheduler", ["slots", "batches"])
def test_server_asr_round_trip(built_lib, micro_case, scheduler):
    """whisper_srv: POST /asr with raw f32 PCM (WhisperHTTPServer.hpp:37-100), JSON reply, 400s, concurrent clients —
    through both schedulers (refilled slots / micro-batches) — and the connection limits: body cap, receive timeout,
    a body that ends early."""
    import json
    import socket
    import threading
    import time
    import urllib.error
    import urllib.request

    from conftest import load_demo_pcm

    srv = os.path.join(os.path.dirname(built_lib.LIB_PATH), "whisper_srv")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    proc = subprocess.Popen([srv, "--port", str(port), "-t", "micro", "-p", micro_case.root, "-l", "zh", "--max_batch", "4",
                             "--scheduler", scheduler, "--max_body_mb", "2", "--recv_timeout_s", "2"],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        base = f"http://127.0.0.1:{port}"
        for _ in range(200):
            try:
                if json.load(urllib.request.urlopen(base + "/health", timeout=2))["status"] == "ok":
                    break
            except Exception:
                time.sleep(0.1)
        else:
            raise AssertionError("server did not come up")
        pcm = load_demo_pcm()

        def post(body, ctype="application/octet-stream"):
            req = urllib.request.Request(base + "/asr", data=body, headers={"Content-Type": ctype}, method="POST")
            try:
                r = urllib.request.urlopen(req, timeout=120)
                return r.status, json.load(r)
            except urllib.error.HTTPError as err:
                return err.code, json.load(err)

        st, js = post(pcm.tobytes())
        assert st == 200 and js["success"] is True and isinstance(js["text"], str)
        e = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
        assert js["text"] == e.run(pcm)
        e.close()
        assert post(pcm.tobytes(), "text/plain")[0] == 400
        assert post(b"")[0] == 400
        assert post(pcm.tobytes()[:-1])[0] == 400
        out = [None] * 6
        th = [threading.Thread(target=lambda i=i: out.__setitem__(i, post(pcm.tobytes()))) for i in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert all(o[0] == 200 and o[1]["text"] == js["text"] for o in out)  # micro-batched requests agree with a single one
        # a poisoned request (NaN samples) gets its 400 without failing the strangers batched with it
        bad = pcm.copy()
        bad[77] = np.nan
        out = [None] * 5
        th = [threading.Thread(target=lambda i=i: out.__setitem__(i, post((bad if i == 2 else pcm).tobytes()))) for i in range(5)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert out[2][0] == 400 and all(o[0] == 200 and o[1]["text"] == js["text"] for i, o in enumerate(out) if i != 2)
        # low load: two, then three DIFFERENT clips at once on an idle device share one multi-clip persistent launch
        # (slots scheduler: the idle-device path; micro-batches: AX_WHISPER_RunPCMBatch of 2 / 3) — every reply its own clip's text;
        # and a poisoned request in such a group fails alone
        import modelgen

        e = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
        assert e.L.AX_WHISPER_GetConfigInt(e.h, b"persistent_max_clips") == 3
        others = [modelgen.synth_clip(70 + i, 50000 + 30000 * i) for i in range(3)]
        texts = [e.run(c) for c in others]
        e.close()
        for n in (2, 3, 2):
            out = [None] * n
            th = [threading.Thread(target=lambda i=i: out.__setitem__(i, post(others[i].tobytes()))) for i in range(n)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert [o[0] for o in out] == [200] * n and [o[1]["text"] for o in out] == texts[:n]
            time.sleep(0.05)
        out = [None] * 2
        th = [threading.Thread(target=lambda i=i: out.__setitem__(i, post((bad if i == 0 else others[1]).tobytes()))) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert out[0][0] == 400 and out[1][0] == 200 and out[1][1]["text"] == texts[1]
        # nine requests at once on four slots: five of them wait in the queue and take slots as they free up
        out = [None] * 9
        th = [threading.Thread(target=lambda i=i: out.__setitem__(i, post(pcm.tobytes()))) for i in range(9)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert all(o[0] == 200 and o[1]["text"] == js["text"] for o in out)
        # limits: a body above --max_body_mb is refused unread; a body that ends (or stalls) before its Content-Length is
        # a 400, never a transcript of half a clip
        def raw(payload, half_close, clen=40000):
            c = socket.create_connection(("127.0.0.1", port), timeout=10)
            c.sendall(b"POST /asr HTTP/1.1\r\nHost: x\r\nContent-Type: application/octet-stream\r\nContent-Length: %d\r\n\r\n" % clen + payload)
            if half_close:
                c.shutdown(socket.SHUT_WR)
            data = b""
            while True:
                chunk = c.recv(65536)
                if not chunk:
                    break
                data += chunk
            c.close()
            return data

        assert b"413" in raw(b"", False, clen=3200000).split(b"\r\n")[0]  # answered on the headers alone, nothing of the body is read
        t0 = time.time()
        assert b"400" in raw(pcm.tobytes()[:20000], True).split(b"\r\n")[0]
        r = raw(pcm.tobytes()[:20000], False)  # the peer just stops sending: --recv_timeout_s 2
        assert b"400" in r.split(b"\r\n")[0] and b"incomplete" in r and time.time() - t0 < 8
        # malformed and hostile requests (the server parses HTTP itself): whatever they get, the server keeps serving
        hdr = b"POST /asr HTTP/1.1\r\nHost: x\r\nContent-Type: application/octet-stream\r\n"
        rng = np.random.default_rng(5)
        junk = [b"", b"\r\n\r\n", b"\x00" * 5000, rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(),
                b"GET /../../etc/passwd HTTP/1.1\r\n\r\n", b"POST /asr\r\n\r\n", b"OPTIONS /asr HTTP/1.1\r\n\r\n",
                hdr + b"Content-Length: -5\r\n\r\n", hdr + b"Content-Length: 99999999999999999999999\r\n\r\n",
                hdr + b"Content-Length: abc\r\n\r\n", hdr + b"Content-Length: 8\r\n\r\n" + np.array([np.nan, np.inf], np.float32).tobytes(),
                hdr + b"Content-Length: 4\r\n\r\n" + np.float32(0.25).tobytes(),
                hdr + b"Transfer-Encoding: chunked\r\n\r\n5\r\nabcde\r\n0\r\n\r\n",
                hdr + b"Content-Length: 16\r\nContent-Length: 4\r\n\r\n" + b"\x00" * 16,
                b"POST /asr HTTP/1.1\r\n" + b"X-Pad: " + b"a" * 1200000,  # headers that never end: dropped at 1 MB
                hdr.replace(b"octet-stream", b"OCTET-STREAM") + b"Content-Length: 64\r\n\r\n" + b"\x00" * 64]

        def shoot(payload):
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=5)
                try:
                    c.sendall(payload)
                    c.shutdown(socket.SHUT_WR)
                    while c.recv(65536):
                        pass
                finally:
                    c.close()
            except OSError:
                pass  # reset by the server: fine

        th = [threading.Thread(target=shoot, args=(junk[i % len(junk)],)) for i in range(4 * len(junk))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert proc.poll() is None
        st2, js2 = post(pcm.tobytes())
        assert st2 == 200 and js2["text"] == js["text"]
        h = json.load(urllib.request.urlopen(base + "/health", timeout=2))
        assert h["status"] == "ok" and h["persistent_giveups"] == 0 and h["served"] >= 20 and h["busy_slots"] == 0 and h["devices"] == 1
    finally:
        proc.kill()
        assert f"scheduler: {scheduler}" in proc.stdout.read(600)


def test_server_one_batcher_per_device(built_lib, micro_case):
    """whisper_srv --devices a,b,...: one handle and one batcher thread per listed device on the shared queue (a one-GPU
    box lists its device twice: two engines, two batchers)."""
    import json
    import socket
    import threading
    import time
    import urllib.request

    from conftest import load_demo_pcm

    srv = os.path.join(os.path.dirname(built_lib.LIB_PATH), "whisper_srv")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    proc = subprocess.Popen([srv, "--port", str(port), "-t", "micro", "-p", micro_case.root, "-l", "zh", "--max_batch", "2",
                             "--devices", "0,0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        base = f"http://127.0.0.1:{port}"
        for _ in range(200):
            try:
                if json.load(urllib.request.urlopen(base + "/health", timeout=2))["status"] == "ok":
                    break
            except Exception:
                time.sleep(0.1)
        else:
            raise AssertionError("server did not come up")
        pcm = load_demo_pcm()
        e = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
        want = e.run(pcm)
        e.close()

        def post(body):
            req = urllib.request.Request(base + "/asr", data=body, headers={"Content-Type": "application/octet-stream"}, method="POST")
            r = urllib.request.urlopen(req, timeout=120)
            return r.status, json.load(r)

        # three bursts: the first one makes both handles grow their slot buffers and capture their step graphs AT THE SAME TIME
        # on one device (round 3: without the per-device capture mutex they invalidated each other's capture and every later
        # request of that handle failed)
        for _ in range(3):
            out = [None] * 9
            th = [threading.Thread(target=lambda i=i: out.__setitem__(i, post(pcm.tobytes()))) for i in range(9)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert all(o is not None and o[0] == 200 and o[1]["text"] == want for o in out), out
    finally:
        proc.kill()
        head = proc.stdout.read(400)
        assert "devices: 2" in head


def test_zh_transcript_goes_through_t2s(built_lib, micro_case, monkeypatch, tmp_path):
    """Whisper.cpp:231-236: for language zh the Run* text result is the detokenised bytes after OpenCC's t2s pass
    (when the reference's t2s.json + dictionaries are found); other languages and the token-id entry points are raw."""
    import ctypes as C

    from conftest import GOLDEN, load_demo_pcm

    pcm = load_demo_pcm()
    from test_t2s import opencc_t2s_dir

    cfg = opencc_t2s_dir(str(tmp_path / "opencc"))
    monkeypatch.setenv("AX_WHISPER_OPENCC_DIR", os.path.dirname(cfg))
    zh = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=1)
    en = built_lib.Whisper("micro", micro_case.root, "en", device=0, max_batch=1)
    try:
        assert zh.L.AX_WHISPER_GetConfigInt(zh.h, b"t2s") == 1 and en.L.AX_WHISPER_GetConfigInt(en.h, b"t2s") == 0
        for e in (zh, en):
            ids = e.run_tokens(pcm, max_new=48)
            raw = e.detokenize(ids)
            out = C.c_void_p()
            a = np.ascontiguousarray(pcm, dtype=np.float32)
            assert e.L.AX_WHISPER_RunPCM(e.h, a.ctypes.data_as(C.POINTER(C.c_float)), len(a), C.byref(out)) == 0
            got = C.string_at(out.value)
            e.L._free(out.value)
            if e is en:
                assert got.startswith(raw[:16]) or raw.startswith(got[:16])  # same decode, unconverted
                continue
            exp = C.c_void_p()
            full = e.detokenize(e.run_tokens(pcm))
            assert e.L.AX_WHISPER_ConvertT2S(cfg.encode(), full, C.byref(exp)) == 0
            assert got == C.string_at(exp.value)
            e.L._free(exp.value)
    finally:
        zh.close()
        en.close()
