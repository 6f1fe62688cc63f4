"""CPU: the in-process multi-device layer behind AX_WHISPER_InitMulti (csrc/multi_device.hpp) with a stand-in engine:
contiguous ceil(B/G) blocks, concurrent workers, ordered join, error propagation, device-list parsing. No GPU, no HIP:
the header is compiled with plain g++. The multi-process form of the same partitioning is tests/test_dp_gloo.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_group_shards_and_joins(tmp_path):
    exe = str(tmp_path / "multi_device_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "whisper.axera_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "multi_device_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "multi_device ok" in r.stdout


def test_in_process_and_multi_process_partitions_agree():
    """dp.shard_range (one rank per GPU, bench.py) and axw::shard_range (one thread per GPU, the C ABI) are one rule."""
    sys.path.insert(0, ROOT)
    import whisper_axera_amd  # noqa: F401
    from whisper_axera_amd import dp

    src = open(os.path.join(ROOT, "whisper.axera_amd", "csrc", "multi_device.hpp")).read()
    assert "(n + world - 1) / world" in src
    for n in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            per = (n + world - 1) // world
            for r in range(world):
                lo = min(r * per, n)
                assert dp.shard_range(n, r, world) == (lo, min(lo + per, n))
