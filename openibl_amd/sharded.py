"""Gallery-sharded matching across the GPUs of a node.

The reference all-gathers every descriptor to every rank and computes the full distance matrix
redundantly on each host CPU (ibl/evaluators.py:76-101, 105-130).  Here the gallery never moves:
rank r keeps the contiguous slice of gallery descriptors it extracted (the slice the reference's
DistributedSliceSampler hands it, ibl/utils/data/sampler.py:208-214), computes
[all queries] x [its slice] distances + a local top-k with GLOBAL gallery indices on its own GPU,
and only the (value, index) lists cross xGMI:

    all_gather(queries)            <= Q x d fp32                   (skipped if already replicated)
    all_gather(top-k values, idx)  =  Q x k x 8 bytes per rank
    k-way merge on every rank      (same top-k kernel, fed the gathered index lists)

One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm).  The local compute steps
are injectable so that the shard / gather / merge logic is testable on CPU with gloo.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def slice_bounds(length: int, rank: int, world_size: int) -> Tuple[int, int, int]:
    """Contiguous slice [start, start + per) of a dataset of `length` items owned by `rank`, as
    DistributedSliceSampler deals them (per = ceil(length / world_size); the tail of the last
    slices wraps around to the first items and is padding).  Returns (start, per, n_valid)."""
    per = -(-length // world_size)
    start = rank * per
    n_valid = max(0, min(per, length - start))
    return start, per, n_valid


def hip_local_topk(q: torch.Tensor, g: torch.Tensor, k: int, index_base: int, precision
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """[Q][d] x [n][d] -> k nearest of the local slice per query (HIP kernels)."""
    Q = q.shape[0]
    if g.shape[0] == 0:
        return (torch.full((Q, k), float("inf"), device=q.device),
                torch.full((Q, k), -1, dtype=torch.int32, device=q.device))
    return ops.sqdist_topk(q, g, k, index_base=index_base, precision=precision)


def hip_merge_topk(vals: torch.Tensor, idx: torch.Tensor, k: int):
    return ops.row_topk(vals.contiguous(), k, idx_in=idx.contiguous())


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate equally-shaped per-rank tensors along dim 0 (one all_gather)."""
    rank, world = _world(group)
    if world == 1:
        return x
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x.contiguous(), group=group)
    return torch.cat(out, dim=0)


def sharded_topk(q_all: torch.Tensor, g_local: torch.Tensor, k: int, index_base: int,
                 precision="fp32", group=None,
                 local_topk_fn: Optional[Callable] = None,
                 merge_fn: Optional[Callable] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """k nearest gallery rows (squared L2) of every query over ALL ranks' gallery slices.

    q_all   [Q][d]  the full query set, identical on every rank
    g_local [n][d]  this rank's valid gallery rows (padding rows removed)
    index_base      global gallery index of g_local[0]
    Returns (values [Q][k] ascending, indices [Q][k] int32 global), identical on every rank.
    Ties are broken towards the lowest global index, so the result does not depend on the number
    of shards."""
    local_topk_fn = local_topk_fn or hip_local_topk
    merge_fn = merge_fn or hip_merge_topk
    rank, world = _world(group)
    v, i = local_topk_fn(q_all, g_local, k, index_base, precision)
    if world == 1:
        return v, i
    vs = [torch.empty_like(v) for _ in range(world)]
    is_ = [torch.empty_like(i) for _ in range(world)]
    dist.all_gather(vs, v.contiguous(), group=group)
    dist.all_gather(is_, i.contiguous(), group=group)
    return merge_fn(torch.cat(vs, dim=1), torch.cat(is_, dim=1), k)
