"""GPU: handle lifecycle, threading and failure behaviour of the C ABI."""
import os
import threading

import numpy as np
import pytest
import torch  # noqa: F401

from conftest import load_demo_pcm

pytestmark = pytest.mark.gpu


def test_one_handle_from_many_threads(built_lib, micro_case):
    """The reference's handle is not re-entrant yet its server calls it from a thread pool; here a handle
    is mutex-serialised: concurrent callers all get the single-threaded answer."""
    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
    pcm = load_demo_pcm()
    want = e.run_tokens(pcm, max_new=10)
    out = [None] * 8
    th = [threading.Thread(target=lambda i=i: out.__setitem__(i, e.run_tokens(pcm, max_new=10))) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(o == want for o in out)
    e.close()


def test_two_handles_are_independent(built_lib, micro_case):
    a = built_lib.Whisper("micro", micro_case.root, "zh", device=0)
    b = built_lib.Whisper("micro", micro_case.root, "en", device=0)
    pcm = load_demo_pcm()
    ra, rb = a.run_tokens(pcm, max_new=8), b.run_tokens(pcm, max_new=8)
    assert a.sot_seq != b.sot_seq
    assert a.run_tokens(pcm, max_new=8) == ra and b.run_tokens(pcm, max_new=8) == rb
    a.close()
    b.close()


def test_init_uninit_cycles_do_not_leak_device_memory(built_lib, micro_case):
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=3)
        e.run_tokens(load_demo_pcm(), max_new=4)
        e.close()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)


def test_capacity_grows_on_demand(built_lib, micro_case):
    import modelgen

    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=1)
    clips = [modelgen.synth_clip(i, 100000) for i in range(7)]
    one = [e.run_tokens(c, max_new=6) for c in clips]
    assert e.run_tokens_batch(clips, max_new=6) == one  # 1 slot -> 7 slots, VALU path -> MFMA path
    e.close()


def test_corrupt_model_directories_fail_with_a_message(built_lib, micro_case, tmp_path):
    import shutil

    L = built_lib.load_library()
    src = os.path.join(micro_case.root, "micro")
    # 1. weights file truncated
    d1 = tmp_path / "a" / "micro"
    shutil.copytree(src, d1)
    p = d1 / "micro.safetensors"
    p.write_bytes(p.read_bytes()[: 4096])
    assert L.AX_WHISPER_Init(b"micro", str(tmp_path / "a").encode(), b"zh") is None
    assert L.AX_WHISPER_LastError(None)
    # 2. config is not JSON (the reference would throw through the C ABI)
    d2 = tmp_path / "b" / "micro"
    shutil.copytree(src, d2)
    (d2 / "micro_config.json").write_text("{ not json")
    assert L.AX_WHISPER_Init(b"micro", str(tmp_path / "b").encode(), b"zh") is None
    assert b"json" in L.AX_WHISPER_LastError(None)
    # 3. a tensor with the wrong shape
    import modelgen

    d3 = tmp_path / "c"
    w = dict(micro_case.weights)
    w["decoder.ln.weight"] = np.ones(64, dtype=np.float32)
    modelgen.write_model_dir(str(d3), "micro", micro_case.dims, weights=w)
    assert L.AX_WHISPER_Init(b"micro", str(d3).encode(), b"zh") is None
    assert b"decoder.ln.weight" in L.AX_WHISPER_LastError(None)
