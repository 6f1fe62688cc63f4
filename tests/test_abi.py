"""CPU: the C-ABI library loads, exports every symbol include/ax_whisper_api.h declares, and fails
loudly (NULL / -1, never a CPU fallback) where no GPU is visible. No compute is called here."""
import ctypes as C
import os
import re

import pytest


def declared_symbols(header_path):
    text = open(header_path).read()
    return sorted(set(re.findall(r"AX_WHISPER_API\s+[\w\s\*]+?\b(AX_WHISPER_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built_lib):
    names = declared_symbols(built_lib.HEADER_PATH)
    assert {"AX_WHISPER_Init", "AX_WHISPER_Uninit", "AX_WHISPER_RunFile", "AX_WHISPER_RunPCM"} <= set(names)
    lib = C.CDLL(built_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(built_lib.SYMBOLS) == names  # the Python binding covers the whole header


def test_only_the_abi_is_visible(built_lib):
    """-fvisibility=hidden as in the reference build (cpp/CMakeLists.txt:10): no C++ symbols leak."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert exported and all(s.startswith("AX_WHISPER_") for s in exported), exported


def test_product_does_not_link_or_import_the_oracle(built_lib):
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ldd = subprocess.run(["ldd", built_lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "torch" not in ldd
    pkg = os.path.join(root, "whisper.axera_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in src and "import oracle" not in src and "whisper_oracle" not in src, f


def test_null_arguments_follow_the_reference_contract(built_lib):
    """ax_whisper_api.cpp:91-93,143-145: NULL args -> -1; Uninit(NULL) is a no-op (:69-74)."""
    L = built_lib.load_library()
    out = C.c_void_p()
    assert L.AX_WHISPER_RunPCM(None, None, 0, C.byref(out)) == -1
    assert L.AX_WHISPER_RunFile(None, b"x.wav", C.byref(out)) == -1
    L.AX_WHISPER_Uninit(None)
    assert L.AX_WHISPER_InitEx(None, b"a", b"zh", -1, 0) is None


def test_init_without_gpu_or_model_fails_loudly(built_lib, tmp_path):
    import torch

    L = built_lib.load_library()
    h = L.AX_WHISPER_Init(b"small", str(tmp_path).encode(), b"zh")
    assert h is None
    err = L.AX_WHISPER_LastError(None).decode()
    assert err  # either "no HIP device visible ..." (CPU box) or "cannot open ..." (GPU box)
    if not torch.cuda.is_available():
        assert "no HIP device" in err
        with pytest.raises(RuntimeError):
            built_lib.Whisper("small", str(tmp_path))
