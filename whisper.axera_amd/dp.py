"""Utterance-level data parallelism across the GPUs of one node.

The hot path shards by independent units: every utterance's front-end, encoder and decode loop share
nothing but read-only weights (the reference itself is one utterance at a time, cpp/src/Whisper.cpp:186-239).
One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests), weights
replicated, contiguous blocks of ceil(B / world) clips per rank, and exactly ONE collective at the end: an
all_gather of fixed-shape int32 rows [count, ids...] — latency-bound (<= 1.8 KB per clip), no reduce.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

N_TEXT_CTX = 448


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block of ceil(n/world) items for `rank` (the last ranks may get fewer or none)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def pack_ids(ids_rows: Sequence[Sequence[int]], rows: int, n_ctx: int = N_TEXT_CTX):
    """Fixed-shape [rows, 1 + n_ctx] int32: column 0 = id count (-1 marks a padding row)."""
    import torch

    t = torch.full((rows, 1 + n_ctx), 0, dtype=torch.int32)
    t[:, 0] = -1
    for i, r in enumerate(ids_rows):
        t[i, 0] = len(r)
        if len(r):
            t[i, 1 : 1 + len(r)] = torch.tensor(list(r), dtype=torch.int32)
    return t


def unpack_ids(t) -> List[List[int]]:
    out = []
    for row in t.tolist():
        if row[0] >= 0:
            out.append(row[1 : 1 + row[0]])
    return out


def gather_ids(local_rows: Sequence[Sequence[int]], rows_per_rank: int, device=None, group=None, n_ctx: int = N_TEXT_CTX):
    """all_gather the per-rank id rows; every rank returns the ids of ALL clips in global order."""
    import torch
    import torch.distributed as dist

    t = pack_ids(local_rows, rows_per_rank, n_ctx)
    if device is not None:
        t = t.to(device)
    import os

    rehearse = os.environ.get("AXW_BENCH_FORCE_DIST") == "1"  # run the collective even with one rank
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not rehearse):
        return unpack_ids(t.cpu())
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t, group=group)
    out: List[List[int]] = []
    for p in parts:
        out.extend(unpack_ids(p.cpu()))
    return out


def transcribe_data_parallel(transcribe_batch: Callable[[list], List[List[int]]], clips: list, rank: int, world: int,
                             device=None, group=None) -> List[List[int]]:
    """Shard `clips` over the ranks, run `transcribe_batch` (this rank's engine) on the local block, gather."""
    lo, hi = shard_range(len(clips), rank, world)
    local = transcribe_batch(clips[lo:hi]) if hi > lo else []
    per = (len(clips) + world - 1) // world
    return gather_ids(local, per, device=device, group=group)
