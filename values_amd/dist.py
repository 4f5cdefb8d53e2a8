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


def float_keys(out: Dict[str, torch.Tensor]):
    """the single-plane float maps of a result dict that travel: the reference's three + the variance map if present"""
    return MAP_KEYS + (("softmax_variance",) if "softmax_variance" in out else ())


def use_all_gather() -> bool:
    """Chosen ONCE, up front, from the backend's capabilities (never by catching an error mid-collective: ranks that
    disagree about which collective comes next hang).  gloo, nccl (= RCCL on ROCm) and mpi implement gather."""
    import torch.distributed as dist
    return str(dist.get_backend()).lower() not in ("gloo", "nccl", "mpi")


def pack_maps(out: Dict[str, torch.Tensor]):
    """(V, K+C, *spatial) f32 [pred_entropy, aleatoric, epistemic, (variance,) mean_softmax...] and (V, *spatial) u8."""
    f = torch.cat([out[k].unsqueeze(1) for k in float_keys(out)] + [out["mean_softmax"]], dim=1)
    return f.contiguous(), out["pred_seg_mean"].contiguous()


def unpack_maps(f: torch.Tensor, seg: torch.Tensor, keys=MAP_KEYS) -> Dict[str, torch.Tensor]:
    out = {k: f[:, i] for i, k in enumerate(keys)}
    out["mean_softmax"] = f[:, len(keys):]
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
    if not use_all_gather():
        dist.gather(f, lists[0] if lists else None, dst=dst)
        dist.gather(seg, lists[1] if lists else None, dst=dst)
    else:
        # a backend without gather: all_gather is universally available (every rank then holds the maps)
        if lists is None:
            lists = ([torch.empty_like(f) for _ in range(world)], [torch.empty_like(seg) for _ in range(world)])
        dist.all_gather(lists[0], f)
        dist.all_gather(lists[1], seg)
    if rank != dst:
        return None
    return unpack_maps(torch.cat(lists[0], 0), torch.cat(lists[1], 0), float_keys(out))


class MapGatherPipeline:
    """gather_maps with the collective OFF the compute path: `submit(out)` packs the step's maps into a send buffer;
    the asynchronous gather of a step (RCCL runs it on its own stream, so later steps' kernels overlap the transfer --
    at ~2 300 volumes/s a step of 32 volumes is 14 ms of compute and 176 MB of maps per rank, i.e. 1.2 GB into rank 0
    per step at 8 GPUs) is started one submit LATER, once the host has seen the step's completion event: a collective
    enqueued right behind its step would sit blocked in a hardware queue that kernels of the next step may share (see
    values_amd.predict.HostPipeline, which measured exactly that with copies).  `collect()` waits for the OLDEST gather
    in flight and returns its maps on dst (None elsewhere); submit() collects first when `depth` gathers are in flight.
    Buffer slots: depth gathers in flight + the step packed but not yet started + the result the caller still holds as
    zero-copy views of a receive buffer (valid until the NEXT submit()).  world == 1: submit/collect are a FIFO."""

    def __init__(self, world: int, rank: int, dst: int = 0, depth: int = 2):
        self.world, self.rank, self.dst, self.depth = world, rank, dst, max(1, depth)
        self._slots = self.depth + 2
        self._send = [None] * self._slots
        self._recv = [None] * self._slots
        self._inflight = []      # [(slot, works)] oldest first
        self._pending = None     # (slot, completion event or None) of the step packed last
        self._next = 0
        self._keys = MAP_KEYS
        self._use_all_gather = use_all_gather() if world > 1 else False

    def _start(self):
        """start the gather of the pending step"""
        import torch.distributed as dist
        slot, ev = self._pending
        self._pending = None
        if ev is not None:
            ev.synchronize()     # host-side: nothing the collective waits for is still running
        sf, ss = self._send[slot]
        if self._recv[slot] is None and (self.rank == self.dst or self._use_all_gather):
            self._recv[slot] = (torch.empty((self.world,) + tuple(sf.shape), dtype=sf.dtype, device=sf.device),
                                torch.empty((self.world,) + tuple(ss.shape), dtype=ss.dtype, device=ss.device))
        have_recv = self._recv[slot] is not None
        rl = (list(self._recv[slot][0].unbind(0)), list(self._recv[slot][1].unbind(0))) if have_recv else (None, None)
        if not self._use_all_gather:
            works = [dist.gather(sf, rl[0] if self.rank == self.dst else None, dst=self.dst, async_op=True),
                     dist.gather(ss, rl[1] if self.rank == self.dst else None, dst=self.dst, async_op=True)]
        else:
            works = [dist.all_gather(rl[0], sf, async_op=True), dist.all_gather(rl[1], ss, async_op=True)]
        self._inflight.append((slot, works))

    def submit(self, out: Dict[str, torch.Tensor]):
        done = None
        if self.world == 1:
            if len(self._inflight) >= self.depth:
                done = self.collect()
            self._inflight.append((0, out))
            return done
        if self._pending is not None:
            if len(self._inflight) >= self.depth:
                done = self.collect()
            self._start()
        slot = self._next
        self._next = (self._next + 1) % self._slots
        self._keys = float_keys(out)
        parts = [out[k].unsqueeze(1) for k in self._keys] + [out["mean_softmax"]]
        seg = out["pred_seg_mean"]
        fshape = (seg.shape[0], sum(p.shape[1] for p in parts)) + tuple(seg.shape[1:])
        if self._send[slot] is None or tuple(self._send[slot][0].shape) != fshape:
            self._send[slot] = (torch.empty(fshape, dtype=torch.float32, device=seg.device), torch.empty_like(seg))
            self._recv[slot] = None
        sf, ss = self._send[slot]
        torch.cat(parts, dim=1, out=sf)      # pack straight into the send buffer (the step's tensors may be reused)
        ss.copy_(seg)
        ev = None
        if seg.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(seg.device))
        self._pending = (slot, ev)
        return done

    def collect(self) -> Optional[Dict[str, torch.Tensor]]:
        """Maps of the oldest gather in flight (dst) / None.  The returned tensors are views of the slot's receive
        buffer: valid until the next submit()."""
        if not self._inflight:
            return None
        slot, works = self._inflight.pop(0)
        if self.world == 1:
            return works
        for w in works:
            w.wait()
        if self.rank != self.dst:
            return None
        rf, rs = self._recv[slot]
        return unpack_maps(rf.view((-1,) + tuple(rf.shape[2:])), rs.view((-1,) + tuple(rs.shape[2:])), self._keys)

    def flush(self) -> List[Optional[Dict[str, torch.Tensor]]]:
        res = []
        if self.world > 1 and self._pending is not None:
            if len(self._inflight) >= self.depth:
                res.append(self.collect())
            self._start()
        while self._inflight:
            res.append(self.collect())
        return res


# ---------------------------------------------------------------------------------------------------------------
# Member-sharded ensembles (BASELINE config C3: 5 members over 8 GPUs).  Work items are (member, volume-block) pairs
# dealt round-robin over the ranks; a rank adds the sufficient statistics of its items into one buffer
# [V][C+1][vox]; ONE sum-reduce to rank 0 (RCCL) combines members that ran on different ranks; rank 0 finalises.
def ensemble_work_items(n_members: int, n_volumes: int, world: int):
    """[(member, v_lo, v_hi)] per rank, the same NUMBER of items on every rank whenever the volumes allow it: the
    volumes are split into world / gcd(members, world) blocks, so members x blocks is a multiple of the rank count
    (5 members on 8 ranks: 8 blocks = 40 items, 5 per rank; round 2 dealt 10 items over 8 ranks, two ranks carrying twice
    the work of the other six).  Fewer volumes than that: as many blocks as there are volumes."""
    import math
    blocks = world // math.gcd(max(1, n_members), world)
    blocks = max(1, min(blocks, max(1, n_volumes)))
    items = []
    for m in range(n_members):
        for b in range(blocks):
            lo, hi = shard_range(n_volumes, blocks, b)
            if hi > lo:
                items.append((m, lo, hi))
    return [items[r::world] for r in range(world)]


def passes_per_member(model, n_pred: int = 1, tta: bool = False, n_aleatoric_samples: int = 10) -> int:
    """Passes predict_cases makes per member (test_3D.py:426-482): 16 TTA views, n_aleatoric_samples draws of an
    aleatoric head (n_pred is ignored there, :458-469), else n_pred MC samples."""
    if tta:
        return 16
    if bool(getattr(model, "aleatoric_loss", False)):
        return int(n_aleatoric_samples)
    return int(n_pred)


def ensemble_uncertainty_sharded(models, x: torch.Tensor, world: int, rank: int, n_pred: int = 1, dst: int = 0,
                                 seeds=None, group=None, tta: bool = False, x_noise=None,
                                 n_aleatoric_samples: int = 10, range_check: str = "fallback") -> Optional[Dict[str, torch.Tensor]]:
    """models: the FULL member list (every rank holds all checkpoints; only its items run).  x: (V,1,D,H,W), the
    same on every rank.  Returns the maps on `dst` (None elsewhere).  The number of passes of a member is taken from the
    logits it produced (and checked against passes_per_member); the total every rank finalises with is the sum over
    ALL members, whichever rank ran them.
    range_check: the fp16 range guard (values_amd.predict.guarded) around THIS rank's items, BEFORE the collective: a rank
    whose forwards overflowed recomputes its own statistics on the native-fp32 kernels and then joins the one sum-reduce --
    the ranks never disagree about which collective comes next."""
    from . import _lib
    from .predict import derive_seed, guarded, pin_seeds, predict_logits
    lib = _lib.load()
    dev = x.device
    V = x.shape[0]
    Cc = models[0].num_classes
    spatial = tuple(x.shape[2:])
    nvox = 1
    for s_ in spatial:
        nvox *= s_
    per_member = [passes_per_member(m, n_pred, tta, n_aleatoric_samples) for m in models]
    stats = torch.zeros((V, Cc + 1) + spatial, dtype=torch.float32, device=dev)
    if seeds is None and range_check != "off":
        seeds = pin_seeds(models, {}, tta=tta).get("seeds")      # a second run of this rank's items replays the same dropout bits

    def my_items():
        stats.zero_()
        for (m, lo, hi) in ensemble_work_items(len(models), V, world)[rank]:
            # one dropout stream per (member, volume block).  The block index enters through an odd-constant stride, as
            # predict_logits decorrelates its chunks: UNet3D.next_seed() rises by ONE per call, so `seed + lo` made block
            # lo = k of call c reuse the masks of block 0 of call c + k (round-4 advice)
            kw = {"seeds": [derive_seed(int(seeds[m]), 1, lo)]} if seeds is not None else {}
            if tta:
                kw["x_noise"] = None if x_noise is None else x_noise[lo:hi]
            logits = predict_logits([models[m]], x[lo:hi], n_pred=n_pred, tta=tta, n_aleatoric_samples=n_aleatoric_samples,
                                    **kw).contiguous()  # (hi-lo, T_m, C, ...)
            T_m = int(logits.shape[1])
            if T_m != per_member[m] or tuple(logits.shape[:3]) != (hi - lo, T_m, Cc):
                raise _lib.VxError(f"ensemble_uncertainty_sharded: member {m} produced logits {tuple(logits.shape)}, "
                                   f"expected {per_member[m]} passes of {Cc} classes")
            _lib.check(lib.vx_unc_stats_accumulate(_lib.ptr(logits), hi - lo, T_m, Cc, nvox, _lib.ptr(stats[lo:hi]),
                                                   _lib.stream_ptr()), "vx_unc_stats_accumulate")

    guarded(models, my_items, range_check, "ensemble_uncertainty_sharded")
    if world > 1:
        import torch.distributed as dist
        dist.reduce(stats, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if rank != dst:
            return None
    return finalize_stats(stats, sum(per_member))


def finalize_stats(stats: torch.Tensor, t_total: int) -> Dict[str, torch.Tensor]:
    from . import _lib
    lib = _lib.load()
    V, C1 = stats.shape[:2]
    Cc = C1 - 1
    spatial = tuple(stats.shape[2:])
    dev = stats.device
    nvox = stats[0, 0].numel()
    out = {k: torch.empty((V,) + spatial, dtype=torch.float32, device=dev) for k in MAP_KEYS}
    out["mean_softmax"] = torch.empty((V, Cc) + spatial, dtype=torch.float32, device=dev)
    out["pred_seg_mean"] = torch.empty((V,) + spatial, dtype=torch.uint8, device=dev)
    _lib.check(lib.vx_unc_stats_finalize(_lib.ptr(stats), V, t_total, Cc, nvox, _lib.ptr(out["mean_softmax"]),
                                         _lib.ptr(out["pred_entropy"]), _lib.ptr(out["aleatoric_uncertainty"]),
                                         _lib.ptr(out["epistemic_uncertainty"]), _lib.ptr(out["pred_seg_mean"]),
                                         _lib.stream_ptr()), "vx_unc_stats_finalize")
    return out
