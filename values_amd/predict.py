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
from .uncertainty import alloc_uncertainty_maps, uncertainty_maps

# flip code bit0 = dim 2 (D), bit1 = dim 3 (H), bit2 = dim 4 (W); order of test_3D.py:430
FLIP_DIMS = [(2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]
TTA_FLIP_CODES = [0] + [sum(1 << (d - 2) for d in dims) for dims in FLIP_DIMS]  # [0,1,2,4,3,5,6,7]



def derive_seed(seed: int, kind: int, index: int) -> int:
    """32-bit dropout seed of sub-stream `index` of a driver level: kind 0 = a volume chunk inside predict_logits, kind 1 = an outer
    block of a driver that calls predict_* per block (dist.ensemble_uncertainty_sharded, sliding.predict_image_sliding).  Index 0
    keeps the seed (a one-chunk run IS the plain run); every other index goes through a multiply-xorshift finaliser with a
    constant per kind, so that (block k, chunk 0) and (block 0, chunk k) never meet -- until round 5 both levels ADDED
    0x9E3779B1 * index, which replayed the same masks for them (round-5 advice)."""
    seed &= 0xFFFFFFFF
    if index == 0:
        return seed
    h = (seed ^ ((0x85EBCA6B if kind == 0 else 0xC2B2AE35) * (index + 1))) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x7FEB352D) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x846CA68B) & 0xFFFFFFFF
    h ^= h >> 16
    return h


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


def range_watch_begin(models) -> None:
    """Zero the fp16-range words of `models` in stream order -- no read, no synchronisation."""
    for mdl in models:
        if hasattr(mdl, "range_reset"):
            mdl.range_reset()


def range_watch_overflowed(models) -> float:
    """The largest magnitude an un-normalised layer of `models` handed to a split-fp16 convolution since
    range_watch_begin (reading synchronises the current stream), or 0.0 when everything stayed below the limit / the
    native-fp32 kernels ran.  A value >= 65504 (or NaN) means the logits computed meanwhile are invalid."""
    if _lib.get_config().conv_fp32 != 0:
        return 0.0
    worst = max([mdl.range_max(reset=True) for mdl in models if hasattr(mdl, "range_max")] or [0.0])
    return worst if not worst < 65504.0 else 0.0


def guarded(models, run, mode: str = "fallback", what: str = "values_amd"):
    """THE fp16 range guard, shared by every driver that launches forwards (predict_uncertainty,
    predict_image_sliding, ensemble_uncertainty_sharded, GraphedPredictor): run() under a zeroed range word; if a
    magnitude >= 65504 reached a split-fp16 convolution, either raise (mode "raise") or compute again on the native-fp32
    matrix kernels (mode "fallback": run() once more under vx_config.conv_fp32 = 1 -- the caller pins its dropout seeds so
    that the second run draws the same bits).  mode "off": just run() (pipelined callers read the word themselves)."""
    if mode not in ("fallback", "raise", "off"):
        raise ValueError("range_check must be 'fallback', 'raise' or 'off'")
    if mode == "off":
        return run()
    range_watch_begin(models)
    out = run()
    worst = range_watch_overflowed(models)
    if worst:
        if mode == "raise":
            raise _lib.VxError(f"{what}: an activation of magnitude {worst:.4g} reached a split-fp16 convolution "
                               "(limit 65504); re-run under values_amd._lib.config(conv_fp32=1)")
        with _lib.config(conv_fp32=1):     # native-fp32 matrix kernels: no range limit; each family keeps its own packed weights
            out = run()
        # the norm / fan-out kernels record magnitudes whatever the conv family: the re-run raised the word again, and a
        # pipelined caller that reads it later (model.check_range()) would see a stale overflow for a batch computed correctly
        range_watch_begin(models)
    return out


def _draws_hash_dropout(m) -> bool:
    return bool(getattr(m, "training", False)) and float(getattr(m, "dropout_prob", 0.0)) > 0 and hasattr(m, "next_seed")


def pin_seeds(models, kw: dict, tta: bool = False) -> dict:
    """kw with explicit hash-dropout seeds: an un-seeded call draws them from the models' counters HERE, once, so that a
    second run of the same batch (the range fallback) replays the same dropout bits.  Only members that DRAW hash dropout
    get a counter seed: an aleatoric head keeps its sampler's fixed default seeds (1234 + member) and an SSN its own draw
    counter -- their `seeds` argument means something else, and pinning it changed their default maps from call to call."""
    if kw.get("seeds") is not None or kw.get("dropout_masks") is not None:
        return kw
    m0 = models[0]
    if hasattr(m0, "rank") and hasattr(m0, "cov_factor_conv"):
        return kw
    if bool(getattr(m0, "aleatoric_loss", False)) and not tta:
        return kw
    if not any(_draws_hash_dropout(m) for m in models):
        return kw
    return dict(kw, seeds=[m.next_seed() if _draws_hash_dropout(m) else 0 for m in models])


_side_streams: Dict[int, list] = {}
_index_cache: Dict[tuple, tuple] = {}


def _slot_indices(dev, v0: int, v1: int, n_total: int, base: int, per_model: int, tta: bool):
    """(src, flip, dst) int32 device tensors of a volume chunk's samples: sample (v, k) writes logits slot
    v * n_total + base + k; TTA sample k reads the original (k < 8) or the noisy (k >= 8) copy of volume v with flip
    code TTA_FLIP_CODES[k % 8].  Built on the host once per geometry and cached: no index arithmetic kernels per step."""
    key = (str(dev), v0, v1, n_total, base, per_model, tta)
    hit = _index_cache.get(key)
    if hit is not None:
        return hit
    import numpy as np
    Vc = v1 - v0
    v = np.arange(v0, v1, dtype=np.int64)[:, None]
    k = np.arange(per_model, dtype=np.int64)[None, :]
    dst = torch.from_numpy((v * n_total + base + k).reshape(-1).astype(np.int32)).to(dev)
    src = flip = None
    if tta:
        lidx = np.arange(Vc, dtype=np.int64)[:, None]
        src = torch.from_numpy((lidx + (k // 8) * Vc).reshape(-1).astype(np.int32)).to(dev)   # orig / noisy copy of v
        codes = np.asarray(TTA_FLIP_CODES * 2, dtype=np.int32)[None, :]
        flip = torch.from_numpy(np.broadcast_to(codes, (Vc, 16)).reshape(-1).copy()).to(dev)
    if len(_index_cache) > 256:
        _index_cache.clear()
    _index_cache[key] = (src, flip, dst)
    return src, flip, dst


def _volume_chunks(V: int, n_streams: Optional[int], pinned_inputs: bool, samples_per_volume: int = 1):
    """[(v0, v1)] volume ranges dealt over n_streams HIP streams.  OPT-IN (n_streams=2 or VX_STREAMS=2; the default is
    one stream: per-kernel timings stay clean and every sample of a batch draws from one hash-dropout stream).
    Two half batches on two streams hide most of one half's HBM-bound
    launches (normalise / pool, transposed convs) under the other's convolutions: +2 % on the 64^3 MC-dropout
    batch of bench.py (un-joined streams: +3..5 %, tools/exp_streams.py); more streams do not add to it.  One stream when masks / noise are injected per
    sample (parity tests) or the batch is too small to split."""
    import os
    if n_streams is None:
        n_streams = int(os.environ.get("VX_STREAMS", "1"))
    n = 1 if (pinned_inputs or V < 4) else max(1, min(int(n_streams), V // 2))
    while n > 1 and (V // n) * samples_per_volume < 64:   # small sample batches lose more in the kernels than overlap wins
        n -= 1
    from .dist import shard_range
    # several chunks per stream (round robin) let the two streams drift out of lockstep after the join that starts a
    # step: 3 per stream measured best on the 64^3 T = 10 batch (+1.5 % over one); chunks stay >= 48 samples
    per = os.environ.get("VX_CHUNKS_PER_STREAM")
    per = int(per) if per else min(3, ((V // n) * samples_per_volume) // 48)
    nc = n * max(1, per) if n > 1 else 1
    nc = max(n, min(nc, V))
    return [shard_range(V, nc, k) for k in range(nc)], n


def predict_logits(models: Sequence, x: torch.Tensor, n_pred: int = 1, tta: bool = False,
                   x_noise: Optional[torch.Tensor] = None, dropout_masks=None, seeds=None,
                   n_aleatoric_samples: int = 10, eps=None, n_streams: Optional[int] = None, _after_chunk=None,
                   seed_dev: Optional[torch.Tensor] = None, **kw_ssn) -> torch.Tensor:
    """x: (V,1,D,H,W).  Returns logits (V, n_total, C, D,H,W) f32 on the device, n_total = passes per volume in
    pred_idx order.  dropout_masks: optional [member][pass] -> 17 masks (parity tests).  n_streams: volume chunks
    run concurrently on that many HIP streams (default: VX_STREAMS or 1), each with its own workspace; every chunk
    writes straight into its pred_idx slots of the one logits tensor.  _after_chunk(logits, v0, v1): called on the
    chunk's stream once its forwards are enqueued (predict_uncertainty reduces the chunk there)."""
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
    if tta and x_noise is None:
        x_noise = gaussian_noise_view(x)
    if tta:
        x_noise = x_noise.to(dev, torch.float32)
    chunks, n_side = _volume_chunks(V, n_streams, pinned_inputs=dropout_masks is not None, samples_per_volume=per_model)
    main = torch.cuda.current_stream(dev)
    side = [main]
    if len(chunks) > 1:
        for model in models:          # pack the weights on the caller's stream: every chunk stream waits for it below
            model._ensure_packed(dev)
        pool = _side_streams.setdefault(dev.index if dev.index is not None else torch.cuda.current_device(), [])
        while len(pool) < n_side:
            pool.append(torch.cuda.Stream(device=dev))
        side = pool[:n_side]
        for st in side:
            st.wait_stream(main)          # x, x_noise and the logits buffer were produced on the caller's stream
    for ci, (v0, v1) in enumerate(chunks):
        with torch.cuda.stream(side[ci % len(side)]):
            xc = x[v0:v1]
            xin = torch.cat([xc, x_noise[v0:v1]], 0) if tta else None   # volumes [0,Vc) orig, [Vc,2Vc) noisy
            for mi, model in enumerate(models):
                base = mi * per_model
                src, flip, dst = _slot_indices(dev, v0, v1, n_total, base, per_model, tta)
                if tta:
                    # (a pinned seed reaches the TTA views too: a dropout member under tta=True then replays the same bits
                    # in the range fallback's second run -- round-4 advice)
                    kw = {"seed": derive_seed(int(seeds[mi]), 0, ci)} if seeds is not None else {}
                    if seed_dev is not None:
                        kw["seed_dev"] = seed_dev
                    model(xin, src=src, flip=flip, dst=dst, out=flat, **kw)
                else:
                    kw = {}
                    if dropout_masks is not None:
                        kw["dropout_masks"] = dropout_masks[mi]
                    if seeds is not None:
                        # one hash-dropout stream per (member seed, chunk): sample indices restart in every chunk
                        kw["seed"] = derive_seed(int(seeds[mi]), 0, ci)
                    if seed_dev is not None:
                        kw["seed_dev"] = seed_dev
                    model(xc, n_samples=n_pred, dst=dst, out=flat, **kw)
            if _after_chunk is not None:
                _after_chunk(logits, v0, v1)
    if len(chunks) > 1:
        for st in side:
            main.wait_stream(st)
    return logits


@torch.no_grad()
def predict_uncertainty(models: Sequence, x: torch.Tensor, n_pred: int = 1, tta: bool = False,
                        x_noise: Optional[torch.Tensor] = None, ssn: bool = False, want_sample_argmax: bool = False,
                        range_check: str = "fallback", **kw) -> Dict[str, torch.Tensor]:
    """Forward passes + fused reduction.  Returns device tensors keyed like the reference's results:
    pred_entropy / aleatoric_uncertainty / epistemic_uncertainty (V,D,H,W) f32 (test_3D.py:509-516),
    mean_softmax (V,C,D,H,W), pred_seg_mean (V,D,H,W) u8 (data_carrier_3D.py:254-255), logits; plus
    softmax_variance (V,D,H,W) f32 -- the north star's fourth map, from the same pass over the logits (no reference
    counterpart).
    range_check: the split-fp16 convolutions represent activations below 65504; every kernel that feeds them an
    un-normalised tensor records the largest magnitude it stored (UNet3D.range_max).  "fallback" (default): the word is
    zeroed in stream order at entry (no read), read once after the batch (the ONE synchronisation of this call) and, if
    the limit was reached, the batch is computed again on the native-fp32 kernels with the same dropout seeds -- NaN maps
    never leave this function; "raise": VxError instead; "off": no read and no reset (pipelined callers -- HostPipeline,
    bench.py -- read the word where they have waited anyway; it keeps the running maximum)."""
    if range_check not in ("fallback", "raise", "off"):
        raise ValueError("range_check must be 'fallback', 'raise' or 'off'")
    if range_check != "off":
        kw = pin_seeds(models, kw, tta=tta)
        if tta and x_noise is None:      # the generated noise view is drawn ONCE: a range fallback re-runs the same views
            x_noise = gaussian_noise_view(x.to(x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device()),
                                               torch.float32))
        return guarded(models, lambda: predict_uncertainty(models, x, n_pred=n_pred, tta=tta, x_noise=x_noise, ssn=ssn,
                                                           want_sample_argmax=want_sample_argmax, range_check="off", **kw),
                       range_check, "predict_uncertainty")
    m = None
    one_batch = (hasattr(models[0], "rank") and hasattr(models[0], "cov_factor_conv")) or \
        (bool(getattr(models[0], "aleatoric_loss", False)) and not tta)      # SSN / aleatoric head: no volume chunks
    if not one_batch:
        # the batch's maps, allocated on the caller's stream; every volume chunk reduces into its rows on its own stream
        _lib.require_gpu()
        dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        n_total = (16 if tta else n_pred) * len(models)
        m = alloc_uncertainty_maps(x.shape[0], n_total, models[0].num_classes, tuple(x.shape[2:]), dev,
                                   want_sample_argmax=want_sample_argmax, want_variance=True)

    def reduce_chunk(lg, v0, v1):
        uncertainty_maps(lg[v0:v1], from_logits=True, want_sample_argmax=want_sample_argmax, want_variance=True,
                         out={k: t[v0:v1] for k, t in m.items()})

    logits = predict_logits(models, x, n_pred=n_pred, tta=tta, x_noise=x_noise,
                            _after_chunk=None if one_batch else reduce_chunk, **kw)
    if one_batch:
        m = uncertainty_maps(logits, from_logits=True, want_sample_argmax=want_sample_argmax, want_variance=True)
    out = {"pred_entropy": m["pred_entropy"], "mean_softmax": m["mean_softmax"], "pred_seg_mean": m["argmax"],
           "softmax_variance": m["softmax_variance"], "logits": logits}
    if not ssn:
        out["aleatoric_uncertainty"] = m["expected_entropy"]
        out["epistemic_uncertainty"] = m["mutual_information"]
    else:
        out["aleatoric_uncertainty"] = m["mutual_information"]
        out["epistemic_uncertainty"] = m["expected_entropy"]
    if want_sample_argmax:
        out["pred_seg"] = m["sample_argmax"]
    return out


class GraphedPredictor:
    """predict_uncertainty for ONE batch geometry as a single captured hipGraph: the ~40 launches of a forward plus the
    reduction replay with one host call (single-volume latency: the eager path is bound by launch overhead there).
    `gp(x, seed=s)` copies x into the graph's input, sets the device seed word every dropout kernel adds to its seed
    (a graph replays with the arguments it was captured with), replays, and returns the graph's OUTPUT tensors -- valid
    until the next call.  MC-dropout / deterministic UNet3D members with hash or no dropout.  The fp16 range word is
    part of the captured work (every replay raises it if a split-fp16 conv saw >= 65504); `gp(x, check=True)` zeroes it
    before the replay, reads it after (one synchronisation) and, on overflow, returns the eager result of the native-fp32
    kernels instead -- without the flag the caller checks `gp.check_range()` where it synchronises anyway."""

    def __init__(self, models: Sequence, shape, n_pred: int = 1, device=None, **predict_kw):
        _lib.require_gpu()
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.models, self.n_pred = list(models), n_pred
        self.x = torch.zeros(tuple(shape), dtype=torch.float32, device=self.dev)
        self.seed_word = torch.zeros(1, dtype=torch.int32, device=self.dev)
        kw = dict(predict_kw, range_check="off", n_streams=1, seeds=[1000003 * (i + 1) for i in range(len(self.models))],
                  seed_dev=self.seed_word)
        self._kw = dict(kw)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):          # warm-up on the capture stream: weights packed, workspaces and index tensors exist
            for _ in range(2):
                predict_uncertainty(self.models, self.x, n_pred=n_pred, **kw)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.out = predict_uncertainty(self.models, self.x, n_pred=n_pred, **kw)
        self._calls = 0

    def __call__(self, x: torch.Tensor, seed: Optional[int] = None, check: bool = False) -> Dict[str, torch.Tensor]:
        self.x.copy_(x, non_blocking=True)
        self._calls += 1
        s = (self._calls if seed is None else int(seed)) & 0x7FFFFFFF
        self.seed_word.fill_(s)
        if check:
            range_watch_begin(self.models)
        self.graph.replay()
        if check and range_watch_overflowed(self.models):
            # the graph holds the split-fp16 family's weight pointers (kept alive per family by the models); the fallback
            # runs eagerly on the native-fp32 family with the seeds the replay used
            with _lib.config(conv_fp32=1):
                return predict_uncertainty(self.models, self.x, n_pred=self.n_pred,
                                           **dict(self._kw, seed_dev=self.seed_word))
        return self.out

    def check_range(self):
        """raise VxError if any replay since the last check overflowed the fp16 range (synchronises)"""
        for mdl in self.models:
            if hasattr(mdl, "check_range"):
                mdl.check_range()


class HostPipeline:
    """predict_uncertainty for callers whose volumes and results live in host memory (the reference reads .npy
    patches and writes NIfTI volumes): a step's maps travel to pinned host buffers while later steps compute, and a
    step's input upload runs ahead of its kernels.  submit(x_host) returns the host maps of the step submitted three
    calls earlier (None while the pipeline fills); flush() returns the rest, oldest first.  The returned arrays are
    views of pinned buffers (three sets): valid until the next submit().

    The rule that makes it overlap on this platform: NO COPY IS EVER ENQUEUED BEFORE WHAT IT WAITS FOR HAS FINISHED.
    Copies and kernels of different HIP streams can share an in-order hardware queue (which streams do is not ours to
    choose), and a download enqueued right behind its step sits there blocked; whatever lands behind it -- the next
    upload, the next step's kernels -- starts only after that download, so every step pays it (measured: 17.6 ms per
    32-volume step instead of 13.5, or not, depending on the order the process happened to create its streams in).
    Here step i's download is enqueued in submit(i + 2), after the HOST has seen step i's completion event; step
    i + 1 is queued on the GPU meanwhile, so the device never idles (tools/exp_hostpipe.py, exp_hostpipe2.py).

    fp16 range guard without a stall: every step runs with range_check="off" (no read on the submit path); a device copy
    of the range word is taken right behind the step's kernels (the word is zeroed in stream order in front of them), and
    travels to the host WITH the step's maps; _collect() -- where the host has waited for that download anyway -- looks at
    it and, if the step overflowed, computes that step again on the native-fp32 kernels (same seeds) before handing its
    maps out (range_check="raise": VxError there instead)."""

    KEYS = ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty", "softmax_variance", "mean_softmax",
            "pred_seg_mean")

    def __init__(self, models: Sequence, device=None, range_check: str = "fallback", **predict_kw):
        _lib.require_gpu()
        if range_check not in ("fallback", "raise", "off"):
            raise ValueError("range_check must be 'fallback', 'raise' or 'off'")
        self.models, self.kw, self.range_check = list(models), predict_kw, range_check
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._up = torch.cuda.Stream(device=self.dev)
        self._down = torch.cuda.Stream(device=self.dev)
        self._slots = [None] * 3     # pinned result buffers
        self._computing = []         # [(device maps, uploaded input, completion event)] oldest first, at most 2
        self._copying = []           # [(slot, copy event, tensors kept alive)] oldest first
        self._next = 0

    def _start_download(self):
        maps, xd, done, rng, kw = self._computing.pop(0)
        done.synchronize()           # host-side: the step has finished, its download will not wait in any queue
        slot = self._next
        self._next = (self._next + 1) % len(self._slots)
        bufs = self._slots[slot]
        if bufs is None or any(tuple(bufs[k].shape) != tuple(maps[k].shape) for k in self.KEYS):
            bufs = {k: torch.empty(maps[k].shape, dtype=maps[k].dtype).pin_memory() for k in self.KEYS}
            bufs["_range"] = torch.zeros(max(1, len(self.models)), dtype=torch.int32).pin_memory()
            self._slots[slot] = bufs
        with torch.cuda.stream(self._down):
            for k in self.KEYS:
                bufs[k].copy_(maps[k], non_blocking=True)
            if rng is not None:
                bufs["_range"][:rng.numel()].copy_(rng, non_blocking=True)
            else:
                bufs["_range"].zero_()
        ev = torch.cuda.Event()
        ev.record(self._down)
        self._copying.append((slot, ev, (maps, xd, kw)))   # the device tensors stay referenced until the copies are done

    def _collect(self):
        slot, ev, keep = self._copying.pop(0)
        ev.synchronize()
        bufs = self._slots[slot]
        if self.range_check != "off" and _lib.get_config().conv_fp32 == 0:
            worst = float(bufs["_range"].view(torch.float32).max().item())      # a pinned HOST tensor: no device work
            if not worst < 65504.0:
                if self.range_check == "raise":
                    raise _lib.VxError(f"HostPipeline: an activation of magnitude {worst:.4g} reached a split-fp16 "
                                       "convolution in this step (limit 65504)")
                _maps, xd, kw = keep
                with _lib.config(conv_fp32=1):     # rare: this step again on the native-fp32 kernels, synchronously
                    out = predict_uncertainty(self.models, xd, **{**self.kw, **kw, "range_check": "off"})
                return {k: out[k].cpu().numpy() for k in self.KEYS}
        return {k: bufs[k].numpy() for k in self.KEYS}

    def submit(self, x_host: torch.Tensor, **kw):
        done = self._collect() if self._copying else None       # the download started one submit ago
        main = torch.cuda.current_stream(self.dev)
        xh = x_host if x_host.is_pinned() else x_host.pin_memory()
        with torch.cuda.stream(self._up):
            xd = xh.to(self.dev, non_blocking=True)
        if len(self._computing) == 2:
            self._start_download()                               # of the step submitted two calls ago
        main.wait_stream(self._up)
        rng = None
        kw = pin_seeds(self.models, {**self.kw, **kw})
        kw.pop("range_check", None)
        if self.range_check != "off":
            range_watch_begin(self.models)                       # zeroed in stream order, not read
        out = predict_uncertainty(self.models, xd, range_check="off", **kw)
        if self.range_check != "off":
            flags = [f for m in self.models for d, f in getattr(m, "_range", {}).items() if d == str(self.dev)]
            if flags:
                rng = torch.cat(flags)                           # this step's word(s), copied on the device behind its kernels
        ev = torch.cuda.Event()
        ev.record(main)
        self._computing.append(({k: out[k] for k in self.KEYS}, xd, ev, rng, kw))
        return done

    def flush(self):
        res = []
        if self._copying:
            res.append(self._collect())
        while self._computing:
            self._start_download()
            res.append(self._collect())
        return res


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
