"""whisper_srv soak over HTTP: N requests of mixed clip lengths from T client threads against the slot scheduler (and the
micro-batch scheduler for comparison); every reply is checked against the text the one-clip entry point gives for that
clip. Prints requests/s, latency percentiles and /health. Synthetic weights: every request decodes the whole context.
usage: soak_server.py [model micro] [n_requests 600] [threads 24] [slots 16]"""
import json
import os
import socket
import subprocess
import sys
import threading
import time
import urllib.request

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "micro"
n_req = int(sys.argv[2]) if len(sys.argv) > 2 else 600
n_thr = int(sys.argv[3]) if len(sys.argv) > 3 else 24
slots = int(sys.argv[4]) if len(sys.argv) > 4 else 16
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, model, model + ".safetensors")):
    modelgen.write_model_dir(mdir, model, seed=0, tiktoken_path=os.path.join(R, "tests", "golden", "multilingual.tiktoken"))
lens = [16000, 77777, 160000, 300000, 480000, 480000, 123457, 240000]
clips = [modelgen.synth_clip(300 + i, lens[i % len(lens)]) for i in range(16)]
e = wa.Whisper(model, mdir, "zh", device=0)
want = [e.run(c) for c in clips]
e.close()
srv = os.path.join(os.path.dirname(wa.LIB_PATH), "whisper_srv")
for sched in ("slots", "batches"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    proc = subprocess.Popen([srv, "--port", str(port), "-t", model, "-p", mdir, "-l", "zh", "--max_batch", str(slots), "--scheduler", sched],
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    base = f"http://127.0.0.1:{port}"
    for _ in range(300):
        try:
            if json.load(urllib.request.urlopen(base + "/health", timeout=2))["status"] == "ok":
                break
        except Exception:
            time.sleep(0.1)
    lat, bad = [], [0]
    nxt = [0]
    lock = threading.Lock()

    def worker():
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= n_req:
                return
            c = i % len(clips)
            req = urllib.request.Request(base + "/asr", data=clips[c].tobytes(), headers={"Content-Type": "application/octet-stream"}, method="POST")
            t0 = time.perf_counter()
            try:
                js = json.load(urllib.request.urlopen(req, timeout=300))
                ok = js.get("success") is True and js.get("text") == want[c]
            except Exception:
                ok = False
            with lock:
                lat.append(time.perf_counter() - t0)
                bad[0] += 0 if ok else 1

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker) for _ in range(n_thr)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    h = json.load(urllib.request.urlopen(base + "/health", timeout=5))
    proc.kill()
    a = np.sort(np.array(lat))
    print(f"{model} {sched}: {n_req} requests, {n_thr} clients, {slots} slots: {n_req / dt:.1f} req/s, latency p50 {a[len(a) // 2] * 1e3:.0f} ms p95 "
          f"{a[int(len(a) * 0.95)] * 1e3:.0f} ms max {a[-1] * 1e3:.0f} ms, wrong or failed replies {bad[0]}, health {h}", flush=True)
