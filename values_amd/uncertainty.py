"""Per-voxel uncertainty reduction on MI355X.

Drop-in for the two functions other reference scripts import from test_3D
(`from uncertainty_modeling.test_3D import calculate_uncertainty, calculate_one_minus_msr`,
test_2D.py:16-22): same names, arguments, returned keys, dtypes and NaN-skip behaviour
(test_3D.py:486-525).  The arithmetic runs in libvalues_amd.so (vx_unc_reduce); there is no CPU path.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib


def _dev_tensor(t: torch.Tensor):
    _lib.require_gpu()
    dev = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    if t.dtype not in (torch.float32, torch.float64):
        t = t.float()
    return t.detach().to(dev).contiguous(), dev


def alloc_uncertainty_maps(B, T, Cc, spatial, dev, want_mean=True, want_argmax=True, want_sample_argmax=False,
                           want_variance=False):
    out = {
        "pred_entropy": torch.empty((B,) + spatial, dtype=torch.float32, device=dev),
        "expected_entropy": torch.empty((B,) + spatial, dtype=torch.float32, device=dev),
        "mutual_information": torch.empty((B,) + spatial, dtype=torch.float32, device=dev),
    }
    if want_mean:
        out["mean_softmax"] = torch.empty((B, Cc) + spatial, dtype=torch.float32, device=dev)
    if want_argmax:
        out["argmax"] = torch.empty((B,) + spatial, dtype=torch.uint8, device=dev)
    if want_sample_argmax:
        out["sample_argmax"] = torch.empty((B, T) + spatial, dtype=torch.uint8, device=dev)
    if want_variance:
        out["softmax_variance"] = torch.empty((B,) + spatial, dtype=torch.float32, device=dev)
    return out


def uncertainty_maps(x: torch.Tensor, from_logits: bool = False, want_mean: bool = True, want_argmax: bool = True,
                     want_sample_argmax: bool = False, out: Dict[str, torch.Tensor] = None, want_variance: bool = False,
                     in_count: Optional[torch.Tensor] = None, out_count: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """x: (B, T, C, *spatial) probabilities or logits (f32/f64) -> dict of device tensors:
    pred_entropy, expected_entropy, mutual_information (B,*spatial) f32, mean_softmax (B,C,*spatial) f32,
    argmax (B,*spatial) u8, sample_argmax (B,T,*spatial) u8, softmax_variance (B,*spatial) f32 (want_variance: the
    north star's fourth map, same pass).  `out`: write into these (contiguous, e.g. the rows of a larger batch's maps
    from alloc_uncertainty_maps) instead of allocating.  in_count / out_count (B,*spatial) f32: divide the inputs / the
    maps by max(count, 1) inside the pass (sliding-window sums; vx_unc_reduce_ex)."""
    lib = _lib.load()
    xd, dev = _dev_tensor(x)
    B, T, Cc = xd.shape[:3]
    spatial = tuple(xd.shape[3:])
    nvox = 1
    for s in spatial:
        nvox *= s
    if from_logits and Cc > 8:
        # the fused logit kernel keeps a voxel's C logits of one sample in registers (C <= 8); wider heads (Cityscapes:
        # 19 classes) take the planar softmax kernel first and the probability reduction after, as predict2d does
        if xd.dtype != torch.float32:
            raise ValueError("uncertainty_maps: from_logits with more than 8 classes needs float32 logits")
        probs = torch.empty_like(xd)
        _lib.check(lib.vx_softmax_planar(_lib.ptr(xd), B * T, Cc, nvox, _lib.ptr(probs), _lib.stream_ptr()), "vx_softmax_planar")
        xd, from_logits = probs, False
    if out is None:
        out = alloc_uncertainty_maps(B, T, Cc, spatial, dev, want_mean, want_argmax, want_sample_argmax, want_variance)
    else:
        for k, t in out.items():
            if not t.is_contiguous() or t.shape[0] != B or t.device != dev:
                raise ValueError(f"uncertainty_maps: out[{k!r}] must be a contiguous device tensor with {B} rows")
    o = _lib.UncOutputs()
    o.mean_prob, o.pred_entropy = _lib.ptr(out.get("mean_softmax")), _lib.ptr(out["pred_entropy"])
    o.exp_entropy, o.mutual_info = _lib.ptr(out["expected_entropy"]), _lib.ptr(out["mutual_information"])
    o.variance = _lib.ptr(out.get("softmax_variance"))
    o.argmax, o.sample_argmax = _lib.ptr(out.get("argmax")), _lib.ptr(out.get("sample_argmax"))
    counts = []
    for cnt in (in_count, out_count):
        if cnt is not None:
            cnt = cnt.to(dev, torch.float32).contiguous()
            if cnt.numel() != B * nvox:
                raise ValueError("uncertainty_maps: a count map must have B * prod(spatial) elements")
        counts.append(cnt)
    o.in_count, o.out_count = _lib.ptr(counts[0]), _lib.ptr(counts[1])
    import ctypes as C
    rc = lib.vx_unc_reduce_ex(_lib.ptr(xd), _lib.VX_F64 if xd.dtype == torch.float64 else _lib.VX_F32, int(from_logits),
                              B, T, Cc, nvox, C.byref(o), _lib.stream_ptr())
    _lib.check(rc, "vx_unc_reduce_ex")
    return out


def softmax_variance(x: torch.Tensor, from_logits: bool = False) -> torch.Tensor:
    """x (B, T, C, *spatial) float32 probabilities or logits -> (B, *spatial): mean over classes of the variance over the
    T samples.  Named in BASELINE.json's north_star; the reference has no such map, so the definition is ours."""
    lib = _lib.load()
    xd, dev = _dev_tensor(x)
    xd = xd.to(torch.float32).contiguous()
    B, T, Cc = xd.shape[:3]
    spatial = tuple(xd.shape[3:])
    nvox = 1
    for s_ in spatial:
        nvox *= s_
    out = torch.empty((B,) + spatial, dtype=torch.float32, device=dev)
    _lib.check(lib.vx_softmax_variance(_lib.ptr(xd), int(from_logits), B, T, Cc, nvox, _lib.ptr(out), _lib.stream_ptr()),
               "vx_softmax_variance")
    return out


def calculate_uncertainty(softmax_preds: torch.Tensor, ssn: bool = False) -> Dict[str, torch.Tensor]:
    """test_3D.py:486-518.  softmax_preds: (T, C, *spatial).  Returns float32 maps on the input's device
    under the reference's keys; `ssn=True` swaps aleatoric/epistemic like test_3D.py:510-516."""
    if softmax_preds.dim() < 2:
        raise ValueError("softmax_preds must be (T, C, *spatial)")
    m = uncertainty_maps(softmax_preds.unsqueeze(0), from_logits=False, want_mean=False, want_argmax=False)
    back = softmax_preds.device
    pe = m["pred_entropy"][0].to(back)
    ee = m["expected_entropy"][0].to(back)
    mi = m["mutual_information"][0].to(back)
    out = {"pred_entropy": pe}
    if not ssn:
        out["aleatoric_uncertainty"] = ee
        out["epistemic_uncertainty"] = mi
    else:
        out["aleatoric_uncertainty"] = mi
        out["epistemic_uncertainty"] = ee
    return out


def calculate_one_minus_msr(softmax_pred: torch.Tensor) -> Dict[str, torch.Tensor]:
    """test_3D.py:521-525: {"pred_entropy": 1 - max over dim 0}, in the input dtype."""
    lib = _lib.load()
    xd, dev = _dev_tensor(softmax_pred)
    Cc = xd.shape[0]
    nvox = xd[0].numel()
    out = torch.empty(xd.shape[1:], dtype=xd.dtype, device=dev)
    rc = lib.vx_one_minus_msr(_lib.ptr(xd), _lib.VX_F64 if xd.dtype == torch.float64 else _lib.VX_F32, Cc, nvox,
                              _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "vx_one_minus_msr")
    return {"pred_entropy": out.to(softmax_pred.device)}
