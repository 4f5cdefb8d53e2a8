"""Multi-GPU: volumes (or patches) are independent, so the path shards with NO collective on the compute
path; the only exchange is collecting the per-volume maps on rank 0: 3 f32 maps + C mean-prob
maps + a u8 mask = 5.5 MB per 64^3 volume, one gather per step over RCCL/xGMI (backend "nccl" on ROCm) -- a
latency-bound message, so a plain gather (point-to-point sends into rank 0, which has a direct xGMI link to
every peer) rather than any ring collective.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

MAP_KEYS = ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty")
_bufs = {}


def shard_range(n_items: int, world: int, rank: int):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_maps(out: Dict[str, torch.Tensor]):
    """(V, 3+C, *spatial) f32 [pred_entropy, aleatoric, epistemic, mean_softmax...] and (V, *spatial) u8."""
    f = torch.cat([out[k].unsqueeze(1) for k in MAP_KEYS] + [out["mean_softmax"]], dim=1)
    return f.contiguous(), out["pred_seg_mean"].contiguous()


def unpack_maps(f: torch.Tensor, seg: torch.Tensor) -> Dict[str, torch.Tensor]:
    out = {k: f[:, i] for i, k in enumerate(MAP_KEYS)}
    out["mean_softmax"] = f[:, len(MAP_KEYS):]
    out["pred_seg_mean"] = seg
    return out


def gather_maps(out: Dict[str, torch.Tensor], world: int, rank: int, dst: int = 0) -> Optional[Dict[str, torch.Tensor]]:
    """Collect every rank's maps on `dst` (rank order = volume order for contiguous shards of equal size).
    Returns the concatenated dict on dst, None elsewhere; world == 1 is a no-op."""
    if world == 1:
        return out
    import torch.distributed as dist
    f, seg = pack_maps(out)
    lists = None
    if rank == dst:
        key = (tuple(f.shape), tuple(seg.shape), str(f.device), world)
        lists = _bufs.get(key)
        if lists is None:
            lists = ([torch.empty_like(f) for _ in range(world)], [torch.empty_like(seg) for _ in range(world)])
            _bufs.clear()
            _bufs[key] = lists
    dist.gather(f, lists[0] if lists else None, dst=dst)
    dist.gather(seg, lists[1] if lists else None, dst=dst)
    if rank != dst:
        return None
    return unpack_maps(torch.cat(lists[0], 0), torch.cat(lists[1], 0))
