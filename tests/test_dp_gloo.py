"""CPU: the N>1 path — sharding + the single result gather — with world_size 2 over gloo (no GPU involved: the
per-rank engine is replaced by a deterministic stand-in, the distributed logic is the code bench.py runs)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_transcribe(clips):
    # ids depend only on the clip content: length and a checksum -> ragged rows incl. an empty one
    out = []
    for c in clips:
        n = int(c.sum().item()) % 7
        out.append([int(c[0].item()) + i for i in range(n)])
    return out


def _worker(rank, world, port, n_clips, ret):
    sys.path.insert(0, ROOT)
    import whisper_axera_amd  # noqa: F401
    from whisper_axera_amd import dp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clips = [torch.arange(5) + 3 * i for i in range(n_clips)]
    got = dp.transcribe_data_parallel(_fake_transcribe, clips, rank, world)
    ret[rank] = got
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [1, 4, 5])
def test_world2_gather_matches_single_process(n_clips):
    sys.path.insert(0, ROOT)
    import whisper_axera_amd  # noqa: F401
    from whisper_axera_amd import dp

    clips = [torch.arange(5) + 3 * i for i in range(n_clips)]
    want = _fake_transcribe(clips)
    assert dp.transcribe_data_parallel(_fake_transcribe, clips, 0, 1) == want
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 500) + n_clips
    mp.spawn(_worker, args=(2, port, n_clips, ret), nprocs=2, join=True)
    assert ret[0] == want and ret[1] == want


def test_shard_ranges_cover_everything_once():
    sys.path.insert(0, ROOT)
    import whisper_axera_amd  # noqa: F401
    from whisper_axera_amd import dp

    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = dp.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert dp.shard_range(512, 3, 8) == (192, 256)  # BASELINE configs[4]: 64 clips per GPU


def test_pack_unpack_round_trip():
    sys.path.insert(0, ROOT)
    import whisper_axera_amd  # noqa: F401
    from whisper_axera_amd import dp

    rows = [[1, 2, 3], [], list(range(444))]
    t = dp.pack_ids(rows, 5)
    assert t.shape == (5, 449) and dp.unpack_ids(t) == rows
