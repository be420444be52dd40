"""TEST INFRASTRUCTURE (not product code): a pure-Python twin of the library's deal (csrc/node.cpp::deal) to check sbv2_deal against, and
a world-size-N transport over torch.distributed / gloo that moves PCM exactly as the library's plan says (sbv2_deal + sbv2_gather_plan through
the C ABI: rank r's message = its utterances in ascending caller index, messages in rank order, then the plan's permutation table), so that
the N > 1 path's counts / offsets / permutation logic is exercised on CPU with two real processes."""
from __future__ import annotations

import numpy as np


def deal(costs, world: int):
    """Longest-processing-time-first deal (utterances by descending cost, each to the least-loaded rank): returns, per rank, the list of
    utterance indices it synthesises.  Loads differ by at most one utterance's cost; the NUMBER of utterances per rank is not bounded
    by ceil(n / world) (one long utterance can balance many short ones).  Deterministic on every rank.  Same rule as the library's
    sbv2_deal (csrc/node.cpp)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    shards = [[] for _ in range(world)]
    load = [0.0] * world
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += costs[i]
    return shards


def gather_by_plan(model, lens, world, rank, dist, make_pcm):
    """Every rank: the library's deal of `lens` (samples per utterance = its cost here), this rank's message packed in plan order, an
    all_gather of padded messages over `dist`; the root applies the library's permutation table.  Returns the caller-order list on rank 0."""
    import ctypes as C

    import torch
    from sbv2_api_amd import _lib
    rank_of = model.deal(lens, world)
    n = len(lens)
    counts = np.zeros(world, np.int64)
    table = np.zeros(3 * max(n, 1), np.int64)
    ln = np.ascontiguousarray(lens, np.int64)
    ro = np.ascontiguousarray(rank_of, np.int32)
    _lib.check(_lib.lib().sbv2_gather_plan(n, ln.ctypes.data_as(_lib.i64p), ro.ctypes.data_as(C.POINTER(C.c_int32)), world,
                                           counts.ctypes.data_as(_lib.i64p), table.ctypes.data_as(_lib.i64p)))
    mine = [i for i in range(n) if rank_of[i] == rank]
    msg = np.concatenate([make_pcm(i) for i in mine]) if mine else np.zeros(0, np.float32)
    assert msg.size == counts[rank]
    cap = max(int(counts.max()), 1)
    send = torch.zeros(cap, dtype=torch.float32)
    send[:msg.size] = torch.from_numpy(msg)
    recv = [torch.empty(cap, dtype=torch.float32) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    stage = np.concatenate([recv[r].numpy()[:counts[r]] for r in range(world)])
    total = int(counts.sum())
    ordered = np.empty(total, np.float32)
    for e in range(n):
        so, do, ll = table[3 * e:3 * e + 3]
        ordered[do:do + ll] = stage[so:so + ll]
    return np.split(ordered, np.cumsum(lens)[:-1]) if n else []
