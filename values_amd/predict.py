"""Batched multi-pass inference: the loops of predict_cases (test_3D.py:399-483) flattened onto the device.

The reference runs T x members x views sequential forwards per patch with a host round trip each; here all
passes of all volumes of a batch are ONE sample batch per ensemble member (MC samples and TTA views are extra
samples that read the same volume through `src` / `flip`), logits land directly in their `pred_idx` slot, and
one fused kernel turns the slots into the uncertainty maps.  pred_idx order is the reference's:
    for model in models:  [TTA: for x in (orig, noisy): identity, then flips (2,),(3,),(4,),(2,3),(2,4),(3,4),(2,3,4)]
                          [else: n_pred MC samples]
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .uncertainty import uncertainty_maps

# flip code bit0 = dim 2 (D), bit1 = dim 3 (H), bit2 = dim 4 (W); order of test_3D.py:430
FLIP_DIMS = [(2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]
TTA_FLIP_CODES = [0] + [sum(1 << (d - 2) for d in dims) for dims in FLIP_DIMS]  # [0,1,2,4,3,5,6,7]


def gaussian_noise_view(x: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The noisy TTA input.  The reference uses batchgenerators' GaussianNoiseTransform (variance ~ U(0, 0.1),
    test_3D.py:428) -- third-party, absent here, RNG-stream dependent: parity UNPINNED.  Callers that need the
    reference's exact view pass it in as `x_noise`."""
    var = torch.rand(x.shape[0], device=x.device, generator=generator) * 0.1
    noise = torch.randn(x.shape, device=x.device, generator=generator, dtype=x.dtype)
    return x + noise * var.sqrt().view(-1, *([1] * (x.dim() - 1)))


@torch.no_grad()
def _predict_logits_aleatoric(models, x, n_samples, eps=None, seeds=None):
    lib = _lib.load()
    V, _, D, H, W = x.shape
    C = models[0].num_classes
    nvox = D * H * W
    out = torch.empty((V, n_samples * len(models), C, D, H, W), dtype=torch.float32, device=x.device)
    for mi, model in enumerate(models):
        mu, s = model(x)                                   # (V, C, ...) each, views of one (V, 2C, ...) tensor
        mu_s = torch.cat([mu, s], 1).contiguous()
        part = torch.empty((V, n_samples, C, D, H, W), dtype=torch.float32, device=x.device)
        e = None if eps is None else eps[mi].to(x.device, torch.float32).contiguous()
        seed = int(seeds[mi]) if seeds is not None else 1234 + mi
        _lib.check(lib.vx_aleatoric_sample(_lib.ptr(mu_s), _lib.ptr(e), seed & 0xFFFFFFFF, V, n_samples, C, nvox,
                                           _lib.ptr(part), None, _lib.stream_ptr()), "vx_aleatoric_sample")
        out[:, mi * n_samples:(mi + 1) * n_samples] = part
    return out


def predict_logits_ssn(model, x: torch.Tensor, n_pred: int = 1, eps_w=None, eps_d=None, seed=None) -> torch.Tensor:
    """predict_cases_ssn (test_3D.py:361-396) for a batch of volumes: x (V,1,D,H,W) -> (V, n_pred, C, D,H,W) logit
    samples of the network's low-rank normal (softmax + reduction follow in uncertainty_maps)."""
    _lib.require_gpu()
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    dist = model(x.to(dev, torch.float32))
    return dist.sample_volumes(n_pred, eps_w=eps_w, eps_d=eps_d, seed=seed)


def predict_logits(models: Sequence, x: torch.Tensor, n_pred: int = 1, tta: bool = False,
                   x_noise: Optional[torch.Tensor] = None, dropout_masks=None, seeds=None,
                   n_aleatoric_samples: int = 10, eps=None, **kw_ssn) -> torch.Tensor:
    """x: (V,1,D,H,W).  Returns logits (V, n_total, C, D,H,W) f32 on the device, n_total = passes per volume in
    pred_idx order.  dropout_masks: optional [member][pass] -> 17 masks (parity tests)."""
    _lib.require_gpu()
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = x.to(dev, torch.float32)
    V, _, D, H, W = x.shape
    if hasattr(models[0], "rank") and hasattr(models[0], "cov_factor_conv"):
        # predict_cases_ssn (test_3D.py:361-396): ONE forward -> distribution; n_pred draws of it
        return predict_logits_ssn(models[0], x, n_pred, eps_w=kw_ssn.get("eps_w"), eps_d=kw_ssn.get("eps_d"),
                                  seed=seeds[0] if seeds is not None else None)
    aleatoric = bool(getattr(models[0], "aleatoric_loss", False)) and not tta
    if aleatoric:
        # test_3D.py:458-469: ONE forward -> (mu, s); n_pred := n_aleatoric_samples draws of mu + exp(s/2) * eps
        return _predict_logits_aleatoric(models, x, n_aleatoric_samples, eps=eps, seeds=seeds)
    per_model = 16 if tta else n_pred
    n_total = per_model * len(models)
    C = models[0].num_classes
    logits = torch.empty((V, n_total, C, D, H, W), dtype=torch.float32, device=dev)
    flat = logits.view(V * n_total, C, D, H, W)
    vidx = torch.arange(V, device=dev, dtype=torch.int32)
    for mi, model in enumerate(models):
        base = mi * per_model
        if tta:
            if x_noise is None:
                x_noise = gaussian_noise_view(x)
            xin = torch.cat([x, x_noise.to(dev, torch.float32)], 0)  # volumes [0,V) orig, [V,2V) noisy
            k = torch.arange(16, device=dev, dtype=torch.int32)
            codes = torch.tensor(TTA_FLIP_CODES * 2, device=dev, dtype=torch.int32)
            src = (vidx[:, None] + (k[None, :] // 8) * V).reshape(-1)           # sample (v,k) reads orig / noisy v
            flip = codes[None, :].expand(V, 16).reshape(-1)
            dst = (vidx[:, None] * n_total + base + k[None, :]).reshape(-1)
            model(xin, src=src, flip=flip, dst=dst, out=flat)
        else:
            k = torch.arange(n_pred, device=dev, dtype=torch.int32)
            dst = (vidx[:, None] * n_total + base + k[None, :]).reshape(-1)
            kw = {}
            if dropout_masks is not None:
                kw["dropout_masks"] = dropout_masks[mi]
            if seeds is not None:
                kw["seed"] = seeds[mi]
            model(x, n_samples=n_pred, dst=dst, out=flat, **kw)
    return logits


@torch.no_grad()
def predict_uncertainty(models: Sequence, x: torch.Tensor, n_pred: int = 1, tta: bool = False,
                        x_noise: Optional[torch.Tensor] = None, ssn: bool = False, want_sample_argmax: bool = False,
                        **kw) -> Dict[str, torch.Tensor]:
    """Forward passes + fused reduction.  Returns device tensors keyed like the reference's results:
    pred_entropy / aleatoric_uncertainty / epistemic_uncertainty (V,D,H,W) f32 (test_3D.py:509-516),
    mean_softmax (V,C,D,H,W), pred_seg_mean (V,D,H,W) u8 (data_carrier_3D.py:254-255), logits."""
    logits = predict_logits(models, x, n_pred=n_pred, tta=tta, x_noise=x_noise, **kw)
    m = uncertainty_maps(logits, from_logits=True, want_sample_argmax=want_sample_argmax)
    out = {"pred_entropy": m["pred_entropy"], "mean_softmax": m["mean_softmax"], "pred_seg_mean": m["argmax"],
           "logits": logits}
    if not ssn:
        out["aleatoric_uncertainty"] = m["expected_entropy"]
        out["epistemic_uncertainty"] = m["mutual_information"]
    else:
        out["aleatoric_uncertainty"] = m["mutual_information"]
        out["epistemic_uncertainty"] = m["expected_entropy"]
    if want_sample_argmax:
        out["pred_seg"] = m["sample_argmax"]
    return out


def crop_indices(image_shape, patch_size: int, patch_overlap: float):
    """Sliding-window crop list of get_val_test_data_samples (toy_datamodule_3D.py:637-655,
    lidc_idri_datamodule_3D.py:723-741): z outermost, x innermost, step int(patch * overlap)."""
    step = int(patch_size * patch_overlap)
    if step <= 0:
        raise ValueError("patch_size * patch_overlap must be >= 1")
    out = []
    z = 0
    while z <= image_shape[2] - patch_size:
        y = 0
        while y <= image_shape[1] - patch_size:
            xx = 0
            while xx <= image_shape[0] - patch_size:
                out.append(((xx, xx + patch_size), (y, y + patch_size), (z, z + patch_size)))
                xx += step
            y += step
        z += step
    return out
