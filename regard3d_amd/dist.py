"""Multi-GPU driver: image pairs shard embarrassingly, one process per GPU, one all-gather.

The reference is single-process (OpenMP over the J's of one I, std::map + omp critical:
/root/reference/src/R3DComputeMatches.cpp:465,481-487); pairs are independent units there too.
Here every rank holds all descriptors (replicated in HBM), takes a cost-balanced slice of the
pair list, runs match + ratio + F-filter locally, and ONE exchange step reassembles the
pairwise match graph on every rank: an all-gather of the per-rank CSR (sizes first, then the
padded payload) over torch.distributed -- backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests.  No collective sits on the matching data path itself.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import api


def shard_pairs(pairs: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Cost-balanced split that keeps the pairs of one I together (per-I tile reuse in L2):
    rows I are dealt to ranks in snake order r = 0..R-1, R-1..0, ... by decreasing pair count."""
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    if world <= 1:
        return pairs
    # the deal itself is the library's (r3dm_shard_pairs: the same rule the single-process multi-GPU entry uses)
    return pairs[api.shard_owner(pairs, world) == rank]


def _pack(g: api.Graph) -> np.ndarray:
    """uint32 wire format: [P, M_lo, M_hi, pairs (2P), counts (P), matches (2M)] -- 8 B per match on the wire"""
    p, o, m = g.pairs, g.offsets, g.matches
    M = int(m.shape[0])
    head = np.array([p.shape[0], M & 0xFFFFFFFF, M >> 32], np.uint32)
    counts = np.diff(o.astype(np.int64)).astype(np.uint32)
    return np.concatenate([head, p.ravel(), counts, m.ravel()])


def _unpack(buf: np.ndarray) -> api.Graph:
    P = int(buf[0]); M = int(buf[1]) | (int(buf[2]) << 32)
    p = buf[3:3 + 2 * P].reshape(-1, 2)
    counts = buf[3 + 2 * P:3 + 3 * P]
    o = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))]).astype(np.uint64)
    m = buf[3 + 3 * P:3 + 3 * P + 2 * M].reshape(-1, 2)
    return api.Graph.from_csr(p, o, m)


def all_gather_graphs(local: Sequence[api.Graph], device=None, group=None, force_collective: bool = False) -> List[api.Graph]:
    """ONE exchange step for several graphs of this rank (e.g. putative + F-filtered): every rank ends
    up with the union over ranks of each graph, ordered by (I, J).  force_collective runs the exchange even in a
    one-rank group (test hook: exercises the RCCL tensor path on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return list(local)
    if dist.get_world_size(group) == 1 and not force_collective:
        return list(local)
    world = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    packs = [_pack(g) for g in local]
    head = np.array([len(packs)] + [p.size for p in packs], np.uint32)
    # torch has no uint32 collectives: ship the same bytes as int32
    payload = torch.from_numpy(np.concatenate([head] + packs).view(np.int32)).to(dev)
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)                       # 8 bytes per rank
    mx = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(mx, dtype=torch.int32, device=dev)
    padded[:payload.numel()] = payload
    bufs = [torch.empty(mx, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)                   # the one payload exchange (RCCL over xGMI)
    per_graph: List[List[api.Graph]] = [[] for _ in local]
    for r in range(world):
        buf = bufs[r][: int(sizes[r].item())].cpu().numpy().view(np.uint32)
        k = int(buf[0]); lens = buf[1:1 + k]; at = 1 + k
        for gi in range(k):
            per_graph[gi].append(_unpack(buf[at:at + int(lens[gi])])); at += int(lens[gi])
    return [api.Graph.merge(parts) for parts in per_graph]


def all_gather_graph(local: api.Graph, device=None, group=None) -> api.Graph:
    return all_gather_graphs([local], device=device, group=group)[0]
