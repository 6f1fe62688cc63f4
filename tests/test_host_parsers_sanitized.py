"""CPU: the host byte parsers under AddressSanitizer + UndefinedBehaviorSanitizer, and the multi-device layer under
ThreadSanitizer (SURVEY §5 "Race detection / sanitizers"; plain g++, never on the GPU box).

tests/cpp/host_parsers_asan.cpp runs every parser that consumes outsider-controlled bytes — csrc/host_io.hpp (JSON,
safetensors header, token table / base64, WAV, AIFF / AIFF-C), csrc/t2s.hpp (OpenCC .ocd2 images), csrc/http_request.hpp
(whisper_srv's request head, network-facing) — on their well-formed fixtures and on >= 10 000 seeded mutations of them
(truncations, bit flips, 32- / 64-bit length fields smashed to boundary values, duplicated / dropped blocks, mutations of
mutations). A parser may refuse an input; it may not fault, hang or invoke undefined behaviour. The reference's counterparts
(cpp/src/AudioFile.h:450-776, cpp/src/utils/WhisperHTTPServer.hpp:50-71) carry no such run.
Defects fixed with this file (round 6). By its first run: memcpy from an empty vector's null data() in the ocd2 bit-vector
reader (UBSan). By inspection while the parsers were made callable on bytes, each now covered by a mutation kind: a WAV chunk
length of 0xFFFFFFF8 never advanced the chunk walk (32-bit wrap: a hang), an 80-bit AIFF sample rate beyond int's range was
cast (UB), `p + n` of the ocd2 reader wrapped for a 64-bit vector size taken from the file, ocd2 bit / flag / tail indices
taken from the file were not bounded, deeply nested JSON exhausted the stack, a malformed JSON number threw
std::invalid_argument through callers that catch runtime_error only, whisper_srv matched "content-length:" anywhere in the
head (an X-Content-Length header set the body length) and took it through strtoul (negative / hex / overlong values)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "whisper.axera_amd", "csrc")
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))


def _fixtures(td):
    """Every well-formed input the parsers see in the tests, as files in one directory (suffix = which parser)."""
    import modelgen

    g = np.load(os.path.join(GOLDEN, "bytes_audio.npz"))
    names = sorted({k.split(".")[0] for k in g.files})
    assert len(names) >= 29
    for name in names:
        suffix = ".wav" if name.startswith(("wav", "demo")) else ".aiff"
        data = g[name + ".file"].tobytes()
        if len(data) > 200000:  # the head of demo.wav is enough: its header and the first samples
            data = data[:65536]
        with open(os.path.join(td, name + suffix), "wb") as f:
            f.write(data)
    shutil.copy(os.path.join(GOLDEN, "multilingual.tiktoken"), os.path.join(td, "full.tiktoken"))
    with open(os.path.join(GOLDEN, "multilingual.tiktoken"), "rb") as f:  # a short table mutates much faster
        lines = f.read().split(b"\n")
    with open(os.path.join(td, "head.tiktoken"), "wb") as f:
        f.write(b"\n".join(lines[:40] + lines[180:200] + lines[-3:]))
    for n in ("TSCharacters.ocd2", "TSPhrases.ocd2"):
        shutil.copy(os.path.join(GOLDEN, "opencc", n), os.path.join(td, n))
    with open(os.path.join(td, "config.json"), "w") as f:
        json.dump(modelgen.make_config("micro", modelgen.DIMS["micro"]), f)
    with open(os.path.join(td, "t2s.json"), "w") as f:  # the shape of the reference's cpp/t2s.json
        json.dump({"name": "Traditional Chinese to Simplified Chinese",
                   "segmentation": {"type": "mmseg", "dict": {"type": "ocd2", "file": "TSPhrases.ocd2"}},
                   "conversion_chain": [{"dict": {"type": "group", "dicts": [{"type": "ocd2", "file": "TSPhrases.ocd2"},
                                                                            {"type": "ocd2", "file": "TSCharacters.ocd2"}]}}]}, f)
    with open(os.path.join(td, "escapes.json"), "w") as f:
        f.write('{"a": "\\u4ea4\\u6613 \\n\\t\\"x\\"", "b": [1, -2.5e3, true, false, null, {"c": []}], "n": "12"}')
    rng = np.random.Generator(np.random.PCG64(3))
    tensors = {"encoder.conv1.weight": rng.standard_normal((4, 3, 3)).astype(np.float32),
               "decoder.ln.weight": rng.standard_normal((8,)).astype(np.float32),
               "decoder.token_embedding.weight": rng.standard_normal((16, 8)).astype(np.float32)}
    modelgen.write_safetensors(os.path.join(td, "small_bf16.safetensors"), tensors, dtype="BF16")
    modelgen.write_safetensors(os.path.join(td, "small_f32.safetensors"), tensors, dtype="F32")
    body = b"\x00\x00\x80\x3f" * 8
    heads = {
        "asr.http": b"POST /asr HTTP/1.1\r\nHost: a\r\nContent-Type: application/octet-stream\r\nContent-Length: 32\r\n\r\n" + body,
        "asr_expect.http": b"POST /asr HTTP/1.1\r\ncontent-type: Application/Octet-Stream\r\nExpect: 100-continue\r\nCONTENT-LENGTH:   32  \r\n\r\n" + body,
        "health.bad.http": b"GET /health HTTP/1.1\r\nHost: a\r\n\r\n",  # (.bad: not a servable /asr request — refused unmutated, by design)
    }
    for n, b in heads.items():
        with open(os.path.join(td, n), "wb") as f:
            f.write(b)
    return len(os.listdir(td))


@pytest.mark.parametrize("seed", [1, 20261005])
def test_host_parsers_under_asan_ubsan_with_mutations(tmp_path, seed):
    exe = str(tmp_path / "host_parsers_asan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                    "-I", CSRC, os.path.join(ROOT, "tests", "cpp", "host_parsers_asan.cpp"), "-o", exe], check=True)
    fx = tmp_path / "fx"
    fx.mkdir()
    n = _fixtures(str(fx))
    assert n >= 29 + 10
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=2048",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, str(fx), "1000", str(seed)], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "0 faults" in r.stdout, r.stdout
    cases = int(r.stdout.split(" fixtures, ")[1].split(" cases")[0])
    refused = int(r.stdout.split(" cases, ")[1].split(" refused")[0])
    print(r.stdout.strip())
    assert cases >= 10000 and 0 < refused < cases  # mutations are refused AND accepted: both paths of every parser run


def test_http_head_contract():
    """The socket-free request parsing against the reference server's contract (WhisperHTTPServer.hpp:50-71), through a tiny
    driver: header names match at line starts only, Content-Length must be a plain decimal number."""
    import tempfile

    src = r'''
#include <cstdio>
#include "http_request.hpp"
int main() {
  using namespace axw;
  HttpHead h;
  int bad = 0;
  auto chk = [&](bool c, const char* what) { if (!c) { printf("FAIL %s\n", what); ++bad; } };
  chk(!parse_http_head("POST /asr HTTP/1.1\r\nContent-Length: 4\r\n", h), "incomplete head");
  chk(parse_http_head("POST /asr HTTP/1.1\r\nContent-Type: application/octet-stream\r\nContent-Length: 8\r\n\r\n12345678", h), "complete");
  chk(h.content_length == 8 && h.length_ok && http_route(h) == HttpRoute::Asr && asr_request_error(h, 8) == nullptr, "asr ok");
  chk(std::string(asr_request_error(h, 0)).find("empty") != std::string::npos, "empty body");
  chk(std::string(asr_request_error(h, 6)).find("multiple of 4") != std::string::npos, "odd size");
  chk(parse_http_head("POST /asr HTTP/1.1\r\nX-Content-Length: 99\r\nContent-Type: text/plain\r\n\r\n", h), "x- header");
  chk(h.content_length == 0 && h.length_ok, "x-content-length is not a content-length");
  chk(std::string(asr_request_error(h, 4)).find("Content-Type") != std::string::npos, "content type");
  chk(parse_http_head("POST /asr HTTP/1.1\r\nContent-Length: -1\r\n\r\n", h) && !h.length_ok, "negative length");
  chk(parse_http_head("POST /asr HTTP/1.1\r\nContent-Length: 0x20\r\n\r\n", h) && !h.length_ok, "hex length");
  chk(parse_http_head("POST /asr HTTP/1.1\r\nContent-Length: 99999999999999999999\r\n\r\n", h) && !h.length_ok, "overlong length");
  chk(parse_http_head("POST /asr HTTP/1.1\r\nEXPECT: 100-Continue\r\n\r\n", h) && h.expect_continue, "expect");
  chk(parse_http_head("GET /health HTTP/1.1\r\n\r\n", h) && http_route(h) == HttpRoute::Health, "health");
  chk(parse_http_head("OPTIONS /asr HTTP/1.1\r\n\r\n", h) && http_route(h) == HttpRoute::Options, "options");
  chk(parse_http_head("GET / HTTP/1.1\r\n\r\n", h) && http_route(h) == HttpRoute::NotFound, "404");
  printf(bad ? "contract FAILED\n" : "http contract ok\n");
  return bad;
}
'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "c.cpp")
        open(c, "w").write(src)
        exe = os.path.join(td, "c")
        subprocess.run(["g++", "-O1", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", CSRC, c, "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "http contract ok" in r.stdout, r.stdout + r.stderr


def test_device_group_under_thread_sanitizer(tmp_path):
    """csrc/multi_device.hpp (one worker thread per device, ordered join, error propagation) under ThreadSanitizer."""
    exe = str(tmp_path / "multi_device_tsan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I", CSRC,
                    os.path.join(ROOT, "tests", "cpp", "multi_device_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "multi_device ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr
