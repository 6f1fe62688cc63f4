"""Experiment: R engine handles with B/R clips each, driven from R host threads on their own streams, against one handle
with B clips (Whisper-small, 30 s synthetic clips resident in HBM). Does the bandwidth-bound attention of one replica
overlap the latency-bound linear layers of the other?

    python profiles/scripts/replicas_b64.py [B] [R]
    AXW_CU_MASK=contig|stride python ...   # every replica on its own stream restricted to 256/R CUs (hipExtStreamCreateWithCUMask)
"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models")
if not os.path.exists(os.path.join(mdir, "small", "small.safetensors")):
    modelgen.write_model_dir(mdir, "small", modelgen.DIMS["small"], seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
dev = torch.device("cuda", 0)
n_samp = 480000
clips = np.stack([modelgen.synth_clip(i, n_samp) for i in range(B)])
d_pcm = torch.from_numpy(clips).to(dev)
torch.cuda.synchronize()


def masked_stream(r, R, kind, n_cu=256):
    hip = ctypes.CDLL("libamdhip64.so")
    bits = [0] * (n_cu // 32)
    for cu in range(n_cu):
        mine = (cu * R // n_cu == r) if kind == "contig" else (cu % R == r)
        if mine:
            bits[cu // 32] |= 1 << (cu % 32)
    arr = (ctypes.c_uint32 * len(bits))(*bits)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(bits)), arr)
    assert rc == 0, rc
    return st.value


def run(R, iters=2):
    per = B // R
    engs = [wa.Whisper("small", mdir, "zh", device=0, max_batch=per) for _ in range(R)]
    kind = os.environ.get("AXW_CU_MASK")
    if kind and R > 1:
        for r, e in enumerate(engs):
            e.set_stream(masked_stream(r, R, kind))
    outs = [None] * R

    def work(r):
        ptr = d_pcm[r * per:(r + 1) * per].data_ptr()
        for _ in range(iters + 1):
            outs[r] = engs[r].run_device_tokens(ptr, n_samp, [n_samp] * per)
            if _ == 0:
                bar.wait()

    bar = threading.Barrier(R + 1)
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    bar.wait()  # warm-up pass done in every replica
    t0 = time.time()
    for t in th:
        t.join()
    dt = time.time() - t0
    for e in engs:
        e.close()
    return B * iters / dt, outs


ref_rate, ref = run(1)
print(f"1 handle x {B} clips: {ref_rate:.1f} clips/s", flush=True)
for R in [int(a) for a in sys.argv[2:]] or [2]:
    rate, outs = run(R)
    flat = [ids for o in outs for ids in o]
    same = sum(a == b for a, b in zip(flat, ref[0]))
    print(f"{R} handles x {B // R} clips: {rate:.1f} clips/s ({rate / ref_rate:.2f}x), ids equal to the single handle for {same}/{B} clips", flush=True)
