"""2D multi-pass inference: Tester.predict_cases + process_output (uncertainty_modeling/test_2D.py:205-319) on the device.

pred order (test_2D.py:283-317): for model in models: [TTA: one forward per view, HorizontalFlip views un-flipped] or
[n_pred forwards]; an SSN model (hrnet_module.py:559-595) makes ONE forward and n_pred draws.  Every forward is its own batch, so training-mode BatchNorm sees the same batch
statistics as in the reference; for DROPOUT_FINAL models the n_pred forwards share the backbone (exact).
Logits land directly in per-image stacks (B, Npred, C, H, W); the per-image reduction is calculate_uncertainty
(Npred > 1) or calculate_one_minus_msr (Npred == 1), test_2D.py:245-248.  The reference appends an all-zero class
channel first (:208-218) -- it contributes exactly 0 to every map through the NaN-skip, so it is not materialised.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .uncertainty import uncertainty_maps


@torch.no_grad()
def tta_views_8(x: torch.Tensor, x_noisy: torch.Tensor):
    """The 8 views BASELINE config 4 words: {identity, HorizontalFlip, VerticalFlip, both} of the clean and of the
    noisy image (the reference's dataset builds 4: identity / HorizontalFlip x clean / GaussNoise,
    cityscapes_dataset.py:76-99; the noise is an input here as there).  Returns (views, hflip flags, vflip flags) in
    pred order: clean views first."""
    views, hf, vf = [], [], []
    for base in (x, x_noisy):
        for h, v in ((False, False), (True, False), (False, True), (True, True)):
            dims = ([-1] if h else []) + ([-2] if v else [])
            views.append(torch.flip(base, dims) if dims else base)
            hf.append(h)
            vf.append(v)
    return views, hf, vf


class NhwcViews:
    """The TTA views of a batch as ONE channels-last tensor (G, B, H, W, 4) -- what vx_tta_views_2d writes
    (values_amd.data.tta_views_2d_device / tta_views_8_device) -- plus which views the driver flips back.  predict_logits_2d
    (tta=True) forwards it as one batch of G statistics groups without any layout copy."""

    def __init__(self, t: torch.Tensor, hflip: Sequence[bool], vflip: Optional[Sequence[bool]] = None):
        if t.dim() != 5 or t.shape[-1] != 4 or len(hflip) != t.shape[0]:
            raise ValueError("NhwcViews: (G, B, H, W, 4) tensor and one flip flag per view")
        self.t, self.hflip, self.vflip = t.contiguous(), list(hflip), list(vflip) if vflip is not None else [False] * t.shape[0]


@torch.no_grad()
def predict_logits_2d(models: Sequence, data, n_pred: int = 1, tta: bool = False, hflip_views: Optional[Sequence[bool]] = None,
                      dropout_masks=None, seeds=None, vflip_views: Optional[Sequence[bool]] = None,
                      batch_views: bool = True, softmax: bool = False) -> torch.Tensor:
    """data: (B,3,H,W) tensor, or with tta a list of view tensors (the dataset's 4 views, cityscapes_dataset.py:76-99,
    or the 8 of tta_views_8) and hflip_views[i] = "HorizontalFlip" in transforms of view i (vflip_views likewise).
    Returns logits (B, Npred_total, C, H, W) -- with softmax=True their class softmax (F.softmax(output, dim=1) of every
    forward, test_2D.py:300-303), taken inside the model's final upsampling pass so that the full-resolution logits are
    never written (not for SSN members: their draws are logits); process_output_2d(probs=...) takes it.  batch_views: the TTA views travel as ONE batch whose BatchNorm statistics
    are kept per view (vx_bn_finalize_groups) -- the same numbers as one forward per view, an eighth of the launches."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    if isinstance(data, NhwcViews):
        if not tta or dropout_masks is not None or any(getattr(m, "ssn", False) for m in models):
            raise ValueError("predict_logits_2d: NhwcViews are the batched TTA views of deterministic / dropout members")
        G, B, H, W, _ = data.t.shape
        total = G * len(models)
        C = models[0].num_classes
        out = torch.empty((B * total, C, H, W), dtype=torch.float32, device=dev)
        codes = [(1 if data.hflip[g] else 0) | (2 if data.vflip[g] else 0) for g in range(G)]
        for mi, model in enumerate(models):
            model.forward_samples(data.t.view(G * B, H, W, 4), 1, seeds=None if seeds is None else [seeds[mi] * 131], out=out,
                                  slot_stride=total, slot_offset=mi * G, groups=G, group_flips=codes, softmax_out=softmax,
                                  nhwc=True)
        return out.view(B, total, C, H, W)
    views = list(data) if tta else [data]
    B, _, H, W = views[0].shape
    per_model = len(views) if tta else n_pred
    total = per_model * len(models)
    C = models[0].num_classes
    if getattr(models[0], "ssn", False) and not tta and len(models) == 1:
        # test_2D.py:285-299: one forward -> distribution, n_pred draws of it
        if softmax:
            raise ValueError("predict_logits_2d: softmax=True is for deterministic / dropout members (an SSN member's draws are logits)")
        return models[0].forward_ssn(views[0]).sample_images(n_pred, seed=None if seeds is None else seeds[0])
    out = torch.empty((B * total, C, H, W), dtype=torch.float32, device=dev)
    for mi, model in enumerate(models):
        base = mi * per_model
        if tta and batch_views and len(views) > 1 and dropout_masks is None:
            codes = [(1 if (hflip_views and hflip_views[vi]) else 0) | (2 if (vflip_views and vflip_views[vi]) else 0)
                     for vi in range(len(views))]
            model.forward_samples(torch.cat([v.to(dev, torch.float32) for v in views], 0), 1,
                                  seeds=None if seeds is None else [seeds[mi] * 131], out=out, slot_stride=total,
                                  slot_offset=base, groups=len(views), group_flips=codes, softmax_out=softmax)
        elif tta:
            for vi, view in enumerate(views):
                model.forward_samples(view, 1, hflip_back=bool(hflip_views[vi]) if hflip_views else False,
                                      vflip_back=bool(vflip_views[vi]) if vflip_views else False,
                                      seeds=None if seeds is None else [seeds[mi] * 131 + vi],
                                      out=out, slot_stride=total, slot_offset=base + vi, softmax_out=softmax)
        else:
            model.forward_samples(views[0], n_pred, dropout_masks=None if dropout_masks is None else dropout_masks[mi],
                                  seeds=None if seeds is None else [seeds[mi] * 131 + t for t in range(n_pred)],
                                  out=out, slot_stride=total, slot_offset=base, softmax_out=softmax)
    return out.view(B, total, C, H, W)


class GraphedPredictor2D:
    """predict_logits_2d + process_output_2d for ONE set of view shapes as a single captured hipGraph.  The eager walk of
    an HRNet forward is ~950 launches from Python -- host-bound below ~25 ms per forward -- and the independent branches
    run on side streams; the capture keeps that fork / join structure and replays it with one host call.
    `gp(views)` copies the views into the graph's inputs, replays, and returns the graph's OUTPUT tensors (valid until
    the next call).  Deterministic members and TTA views; for DROPOUT_FINAL models the hash seeds are the ones given at
    construction (every replay draws the same masks -- use the eager path for fresh MC-dropout samples)."""

    def __init__(self, models: Sequence, example, n_pred: int = 1, tta: bool = False, hflip_views=None, vflip_views=None,
                 seeds=None, ssn: bool = False, keep_logits: bool = True):
        """keep_logits=False: the graph holds no full-resolution logits (`self.logits` is None) -- every forward's softmax
        is taken in its upsampling pass (predict_logits_2d(softmax=True)); the maps are the same bits.  (An SSN member's
        draws ARE logits: with ssn=True the logits are kept whatever the flag says.)"""
        _lib.require_gpu()
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.models, self.tta = list(models), tta
        if any(getattr(m, "dropout_final", False) for m in self.models) and seeds is None:
            raise ValueError("GraphedPredictor2D: DROPOUT_FINAL members need explicit seeds (they are baked into the graph)")
        self._nhwc = isinstance(example, NhwcViews)
        if self._nhwc:      # the views as ONE channels-last tensor (vx_tta_views_2d): self.x[0] is the graph's input -- write
            # the next step's views straight into it (tta_views_8_device(..., out=gp.x[0])) and call gp() without arguments
            self.x = [example.t.detach().to(self.dev, torch.float32).clone()]
            data = NhwcViews(self.x[0], example.hflip, example.vflip)
            self._kw = dict(n_pred=n_pred, tta=True, seeds=seeds)
        else:
            ex = list(example) if tta else [example]
            self.x = [v.detach().to(self.dev, torch.float32).clone() for v in ex]
            data = self.x if tta else self.x[0]
            self._kw = dict(n_pred=n_pred, tta=tta, hflip_views=hflip_views, vflip_views=vflip_views, seeds=seeds)
        self._ssn = ssn

        fused = not keep_logits and not ssn

        def run():
            if fused:
                pr = predict_logits_2d(self.models, data, softmax=True, **self._kw)
                return None, process_output_2d(None, ssn=ssn, probs=pr)
            lg = predict_logits_2d(self.models, data, **self._kw)
            return lg, process_output_2d(lg, ssn=ssn)

        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):      # warm-up ON the capture stream: weights packed, zero-tail buffers, slot tables, side streams
            for _ in range(2):
                run()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.logits, self.out = run()
        self._keep = [getattr(m, "_hold_last", None) for m in self.models]
        # the two warm-up walks left their intermediates cached in the side stream's pool (a forward holds every
        # intermediate until its end: tens of GB at 1024 x 512 x 32 views); the graph has its own pool -- give the
        # warm-up blocks back, or a later eager forward in this process may not fit beside them
        for m in self.models:
            if hasattr(m, "_hold_last"):
                m._hold_last = None
        torch.cuda.empty_cache()

    def __call__(self, views=None) -> Dict[str, torch.Tensor]:
        if views is None:                 # the caller wrote the inputs into self.x itself
            self.graph.replay()
            return self.out
        if self._nhwc:
            views = [views.t if isinstance(views, NhwcViews) else views]
        vs = list(views) if (self.tta or self._nhwc) else [views]
        if len(vs) != len(self.x):
            raise ValueError("GraphedPredictor2D: %d views, captured with %d" % (len(vs), len(self.x)))
        for dst, v in zip(self.x, vs):
            if tuple(v.shape) != tuple(dst.shape):
                raise ValueError("GraphedPredictor2D: view shape %s, captured with %s" % (tuple(v.shape), tuple(dst.shape)))
            dst.copy_(v, non_blocking=True)
        self.graph.replay()
        return self.out


@torch.no_grad()
def process_output_2d(logits: Optional[torch.Tensor], ssn: bool = False, probs: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """logits (B, Npred, C, H, W) -- or probs, their class softmax from predict_logits_2d(softmax=True) -- -> per-image maps like process_output: softmax_pred (B, Npred, C, H, W),
    mean_softmax (B, C, H, W), pred_seg (B, H, W) u8 argmax of the mean (save_prediction, test_2D.py:116-149) and
    pred_entropy [/ aleatoric_uncertainty / epistemic_uncertainty] (B, H, W)."""
    lib = _lib.load()
    if probs is None and logits is None:
        raise ValueError("process_output_2d: give logits or probs")
    if probs is None:
        B, T, Cc, H, W = logits.shape
        probs = torch.empty_like(logits)
        _lib.check(lib.vx_softmax_planar(_lib.ptr(logits), B * T, Cc, H * W, _lib.ptr(probs), _lib.stream_ptr()), "vx_softmax_planar")
    B, T, Cc, H, W = probs.shape
    m = uncertainty_maps(probs, from_logits=False)
    out = {"softmax_pred": probs, "mean_softmax": m["mean_softmax"], "pred_seg": m["argmax"]}
    if T > 1:
        out["pred_entropy"] = m["pred_entropy"]
        a, e = ("aleatoric_uncertainty", "epistemic_uncertainty") if not ssn else ("epistemic_uncertainty", "aleatoric_uncertainty")
        out[a], out[e] = m["expected_entropy"], m["mutual_information"]
    else:  # calculate_one_minus_msr: 1 - max softmax under the key "pred_entropy" (test_3D.py:521-525)
        msr = torch.empty((B, H, W), dtype=torch.float32, device=probs.device)
        for b in range(B):
            _lib.check(lib.vx_one_minus_msr(_lib.ptr(probs[b, 0]), _lib.VX_F32, Cc, H * W, _lib.ptr(msr[b]), _lib.stream_ptr()),
                       "vx_one_minus_msr")
        out["pred_entropy"] = msr
    return out
