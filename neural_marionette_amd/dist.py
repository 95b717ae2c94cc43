"""Clip sharding across the GPUs of one node.

The path shards by clip: every quantity of the forward is per-clip except batch means of the
scalar losses, weights are replicated, and the skeleton tree is a function of the weights
only.  Inference therefore needs no data-path collective; these helpers only partition the
batch and (optionally) gather per-clip results / average the scalar losses.  One process per
GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI) on the GPU box, "gloo" in
the CPU tests.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def clip_shard(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the clips owned by `rank`: contiguous, sizes differ by at most one."""
    base, extra = divmod(n_clips, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_clips(local: torch.Tensor, n_clips: int) -> torch.Tensor:
    """All-gather a per-clip tensor (first dim = this rank's clips) into the global clip order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [clip_shard(n_clips, r, world) for r in range(world)]
    most = max(b - a for a, b in sizes)
    pad = torch.zeros(most, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)], dim=0)


def mean_losses(losses: Dict[str, torch.Tensor], n_local: int, n_clips: int) -> Dict[str, torch.Tensor]:
    """Batch-mean scalars of the sharded forward -> the global batch mean (clip-weighted)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return losses
    out = {}
    for k, v in losses.items():
        t = v.detach().clone().to(torch.float64) * n_local
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out[k] = (t / n_clips).to(v.dtype)
    return out
