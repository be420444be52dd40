"""Utterance sharding across the GPUs of one node and the gather of PCM to rank 0.

Utterances are independent, so the only exchange on the path is the final gather; `torch.distributed` is plumbing
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
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


def gather_pcm(local_ids, local_pcm, n_total: int, dist, device="cpu", dst: int = 0):
    """Gather variable-length PCM arrays (torch tensors or numpy) of this rank's utterances to rank `dst`.

    One all_gather of the (index, length) table + one padded gather of the samples.  Returns on `dst` a list of
    n_total numpy arrays in the original utterance order, elsewhere None."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    # rows of the (index, length) table = the LARGEST shard of any rank: deal() balances cost, not count, so a rank can hold more than
    # ceil(n_total / world) utterances (costs [512, 32, 32, 32, 32, 40] on two ranks deal 1 + 5)
    kk = torch.tensor([len(local_ids)], dtype=torch.int64, device=device)
    dist.all_reduce(kk, op=dist.ReduceOp.MAX)
    k = max(1, int(kk.item()))
    meta = torch.full((k, 2), -1, dtype=torch.int64, device=device)
    for j, (i, p) in enumerate(zip(local_ids, local_pcm)):
        meta[j, 0], meta[j, 1] = i, len(p)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    tot = [int(m[:, 1].clamp(min=0).sum()) for m in metas]
    cap = max(max(tot), 1)
    send = torch.zeros(cap, dtype=torch.float32, device=device)
    off = 0
    for p in local_pcm:
        t = p if isinstance(p, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32))
        send[off:off + len(p)] = t.to(device)
        off += len(p)
    recv = [torch.empty(cap, dtype=torch.float32, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    out = [None] * n_total
    for r in range(world):
        off = 0
        buf = recv[r].cpu().numpy()
        for i, n in metas[r].cpu().tolist():
            if i >= 0:
                out[i] = buf[off:off + n].copy()
                off += n
    return out
