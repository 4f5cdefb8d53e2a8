"""Sliding-window inference over whole images: get_val_test_data_samples + predict_cases + concat_data +
caculcate_uncertainty_multiple_pred (toy_datamodule_3D.py:637-655, test_3D.py:417-534, data_carrier_3D.py:99-179,
208-217) with the patches batched on the device.

compat=True reproduces the reference exactly, including quirk D10: `calculate_uncertainty` is applied to the
UN-normalised sums of overlapping patches and only the resulting maps are divided by clip(count, 1) at save time
(data_carrier_3D.py:323-337).  With the shipped `patch_overlap: 1` the count is 1 and both modes agree.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import _lib
from .predict import crop_indices, derive_seed, guarded, pin_seeds, predict_logits
from .uncertainty import uncertainty_maps


@torch.no_grad()
def predict_image_sliding(models: Sequence, image: torch.Tensor, patch_size: int = 64, patch_overlap: float = 1,
                          n_pred: int = 1, tta: bool = False, patch_batch: int = 8, compat: bool = True,
                          seeds=None, noise_fn=None, n_aleatoric_samples: int = 10, ssn: bool = False,
                          range_check: str = "fallback", **predict_kw) -> Dict[str, torch.Tensor]:
    """image: (X, Y, Z) float tensor (the preprocessed .npy of load_image).  Returns device tensors:
    softmax_sum (T, C, X,Y,Z), num_predictions (X,Y,Z), pred_entropy / aleatoric_uncertainty / epistemic_uncertainty
    (X,Y,Z) -- already divided by clip(count,1) like save_data --, mean_softmax (C, X,Y,Z), pred_seg_mean (X,Y,Z) u8.
    The number of passes T is whatever predict_logits produced for the model kind (n_pred, 16 TTA views, or
    n_aleatoric_samples for an aleatoric head: test_3D.py:458-469 sets n_pred := n_aleatoric_samples there), times the
    number of members.  ssn=True swaps the aleatoric / epistemic maps like calculate_uncertainty(ssn=True)
    (test_3D.py:510-516).
    range_check: the fp16 range guard of values_amd.predict.guarded over the WHOLE image -- the range word is zeroed before
    the first patch batch (no read), read once after the last accumulate, and on overflow the image is computed again on
    the native-fp32 kernels with the same seeds ("raise": VxError; "off": the caller checks model.check_range())."""
    _lib.require_gpu()
    if range_check != "off":
        if seeds is None:
            seeds = pin_seeds(models, {}, tta=tta).get("seeds")
        return guarded(models, lambda: predict_image_sliding(models, image, patch_size=patch_size, patch_overlap=patch_overlap,
                                                             n_pred=n_pred, tta=tta, patch_batch=patch_batch, compat=compat,
                                                             seeds=seeds, noise_fn=noise_fn,
                                                             n_aleatoric_samples=n_aleatoric_samples, ssn=ssn,
                                                             range_check="off", **predict_kw),
                       range_check, "predict_image_sliding")
    lib = _lib.load()
    dev = image.device if image.is_cuda else torch.device("cuda", torch.cuda.current_device())
    img = image.to(dev, torch.float32).contiguous()
    X, Y, Z = img.shape
    crops = crop_indices((X, Y, Z), patch_size, patch_overlap)
    P = patch_size
    C = models[0].num_classes
    T, ssum = None, None          # allocated once the first batch has shown how many passes a patch gets
    count = torch.zeros((X, Y, Z), dtype=torch.float32, device=dev)
    overlap = int(int(P * patch_overlap) < P)
    for b0 in range(0, len(crops), patch_batch):
        batch = crops[b0:b0 + patch_batch]
        x = torch.stack([img[c[0][0]:c[0][1], c[1][0]:c[1][1], c[2][0]:c[2][1]] for c in batch]).unsqueeze(1)
        kw = {}
        if seeds is not None:
            kw["seeds"] = [derive_seed(int(s), 1, b0) for s in seeds]   # (one stream per patch batch: predict.derive_seed)
        x_noise = noise_fn(x) if (tta and noise_fn is not None) else None
        logits = predict_logits(models, x, n_pred=n_pred, tta=tta, x_noise=x_noise,
                                n_aleatoric_samples=n_aleatoric_samples, **kw, **predict_kw)  # (B, T, C, P,P,P)
        if ssum is None:
            T = int(logits.shape[1])
            ssum = torch.zeros((T, C, X, Y, Z), dtype=torch.float32, device=dev)
        if tuple(logits.shape) != (len(batch), T, C, P, P, P):
            raise _lib.VxError(f"predict_image_sliding: logits {tuple(logits.shape)} != {(len(batch), T, C, P, P, P)}")
        logits = logits.contiguous()
        crop_t = torch.tensor([[c[0][0], c[1][0], c[2][0]] for c in batch], dtype=torch.int32, device=dev)
        rc = lib.vx_softmax_accumulate(_lib.ptr(logits), len(batch), T, C, P, P, P, _lib.ptr(crop_t), _lib.ptr(ssum),
                                       _lib.ptr(count), X, Y, Z, overlap, _lib.stream_ptr())
        _lib.check(rc, "vx_softmax_accumulate")
    # maps of the un-normalised sums divided by clip(count, 1) afterwards (compat: quirk D10), or of the normalised
    # sums -- either division happens inside the reduction pass (vx_unc_reduce_ex), no separate tensor arithmetic
    cnt = count.unsqueeze(0)
    m = uncertainty_maps(ssum.unsqueeze(0), from_logits=False, want_variance=True,
                         **({"out_count": cnt} if compat else {"in_count": cnt}))
    pe, ee, mi = (m[k][0] for k in ("pred_entropy", "expected_entropy", "mutual_information"))
    # mean over T of softmax / clip(count,1) and its argmax (data_carrier_3D.py:215-217, 254-255); the per-voxel
    # count does not change the argmax
    mean = m["mean_softmax"][0]
    if ssn:
        ee, mi = mi, ee
    return {"softmax_sum": ssum, "num_predictions": count, "pred_entropy": pe, "aleatoric_uncertainty": ee,
            "epistemic_uncertainty": mi, "mean_softmax": mean, "pred_seg_mean": m["argmax"][0],
            "softmax_variance": m["softmax_variance"][0]}
