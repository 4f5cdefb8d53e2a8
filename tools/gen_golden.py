#!/usr/bin/env python3
"""Generate tests/golden/* by IMPORTING the reference (/root/reference) and running
its own functions/classes.  Runs only in the build container (the reference never
travels to the GPU box); the outputs are small data fixtures that do.

    python tools/gen_golden.py            # writes tests/golden/

Third-party modules the reference imports at module top but that are not installed
here (hydra, torchmetrics, batchgenerators, medpy, ...) are replaced by MagicMock --
none of them is on the arithmetic path captured below.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import json
import os
import sys
import tempfile
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MOCK_ROOTS = ("hydra", "omegaconf", "torchmetrics", "batchgenerators", "medpy", "pytorch_lightning",
              "torchvision", "cv2", "albumentations", "jsbeautifier", "pydantic_settings", "SimpleITK")


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MOCK_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _MockFinder())
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/uncertainty_modeling")
sys.path.insert(0, "/root/reference/evaluation")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import uncertainty_modeling.test_3D as ref_t3  # noqa: E402
from uncertainty_modeling.data_carrier_3D import DataCarrier3D  # noqa: E402
from uncertainty_modeling.models.unet3D_module import UNet3D as RefUNet3D  # noqa: E402
from uncertainty_modeling.toy_datamodule_3D import get_val_test_data_samples as ref_samples_toy  # noqa: E402
from uncertainty_modeling.lidc_idri_datamodule_3D import get_val_test_data_samples as ref_samples_lidc  # noqa: E402
import evaluation.uncertainty_aggregation.aggregate_uncertainties as ref_agg  # noqa: E402

from tests.formula import formula_tensor, formula_unet3d_state_dict, formula_volume, hash_uniform  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def _t(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


# ----------------------------------------------------------------------------- G1
def gen_unc_kat():
    out = {}
    # hand case: T=2, C=2, 4 voxels
    p = np.array([[1.0, 0.0], [0.9, 0.1], [0.5, 0.5], [0.0, 1.0]]).T  # (C, 4)
    q = np.array([[0.0, 1.0], [0.6, 0.4], [0.5, 0.5], [0.0, 1.0]]).T
    hand = np.stack([p, q])  # (2, 2, 4)
    out["hand_in"] = hand
    for ssn in (False, True):
        r = _t(ref_t3.calculate_uncertainty(torch.from_numpy(hand), ssn=ssn))
        for k, v in r.items():
            out[f"hand_{'ssn_' if ssn else ''}{k}"] = v
    # random (4,2,16,16,16) f64 softmax
    logits = formula_tensor((4, 2, 16, 16, 16), tag=11, scale=4.0)
    sm = F.softmax(torch.from_numpy(logits), dim=1)
    out["r3d_in"] = sm.numpy()
    for k, v in _t(ref_t3.calculate_uncertainty(sm)).items():
        out[f"r3d_{k}"] = v
    # 2D layout: (4, 25, 32, 48) f32 with an all-zero last channel (test_2D.py:208-218)
    logits = formula_tensor((4, 24, 16, 24), tag=12, scale=3.0).astype(np.float32)
    sm = F.softmax(torch.from_numpy(logits), dim=1)
    sm = torch.cat([sm, torch.zeros(4, 1, 16, 24)], dim=1)
    out["r2d_in"] = sm.numpy()
    for k, v in _t(ref_t3.calculate_uncertainty(sm)).items():
        out[f"r2d_{k}"] = v
    # exact 0/1 probabilities mixed with ordinary ones
    u = hash_uniform(3 * 512, tag=13).reshape(3, 512)
    p0 = np.where(u < -0.5, 0.0, np.where(u > 0.5, 1.0, (u + 1) / 2))
    ex = np.stack([p0, 1 - p0], axis=1).reshape(3, 2, 8, 8, 8)
    out["ex_in"] = ex
    for k, v in _t(ref_t3.calculate_uncertainty(torch.from_numpy(ex))).items():
        out[f"ex_{k}"] = v
    # one-minus-msr
    one = sm[0].numpy()
    out["msr_in"] = one
    out["msr_pred_entropy"] = _t(ref_t3.calculate_one_minus_msr(torch.from_numpy(one)))["pred_entropy"]
    np.savez_compressed(os.path.join(OUT, "unc_kat.npz"), **out)
    print("G1 unc_kat.npz", {k: v.shape for k, v in out.items() if k.endswith("pred_entropy")})


# ----------------------------------------------------------------------------- G2
def _ref_model(seed_tag=0, do_dropout=True, **kw):
    m = RefUNet3D(num_classes=2, do_dropout=do_dropout, **kw)
    sd = formula_unet3d_state_dict(seed_tag=seed_tag)
    m.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    return m.double()  # test_3D.py:425


def _hook_dropout_masks(model):
    """Capture the keep-mask of each of the 17 nn.Dropout modules, by module name."""
    store = {}
    handles = []

    def mk(name):
        def hook(mod, inp, out):
            x = inp[0]
            keep = (out != 0) | (x == 0)  # where the input is exactly 0 the mask is irrelevant
            store[name] = keep.detach().clone()
        return hook

    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Dropout):
            top = name.split(".")[0]
            handles.append(mod.register_forward_hook(mk(top)))
    return store, handles


def gen_unet(size, T, fname):
    torch.manual_seed(123)
    model = _ref_model(do_dropout=True)
    x = torch.from_numpy(formula_volume((1, 1, size, size, size)))
    store, handles = _hook_dropout_masks(model)
    out = {"input": x.numpy().astype(np.float32)}
    logits_all, sm_all = [], []
    for t in range(T):
        with torch.no_grad():
            logits = model.forward(x)
        sm = F.softmax(logits, dim=1)  # test_3D.py:472
        logits_all.append(logits[0].numpy())
        sm_all.append(sm[0])
        for name, m in store.items():
            out[f"mask_{t}_{name}"] = np.packbits(m.numpy().astype(np.uint8).ravel())
            out[f"maskshape_{name}"] = np.array(m.shape)
    for h in handles:
        h.remove()
    logits_all = np.stack(logits_all)  # (T, 2, S,S,S) f64
    sm_stack = torch.stack(sm_all)  # (T, 2, ...)
    out["logits"] = logits_all.astype(np.float32)
    # buffer exactly like concat_data: numpy float64 (T,2,...) -> calculate_uncertainty
    buf = sm_stack.numpy().astype(np.float64)
    unc = _t(ref_t3.calculate_uncertainty(torch.from_numpy(buf)))
    for k, v in unc.items():
        out[k] = v
    mean = np.mean(buf, axis=0)  # data_carrier_3D.py:254
    out["mean_softmax"] = mean.astype(np.float32)
    out["mean_seg"] = np.argmax(mean, axis=0).astype(np.uint8)  # :255
    out["pred_seg"] = np.argmax(buf, axis=1).astype(np.uint8)  # :282
    srt = np.sort(mean, axis=0)
    out["mean_margin"] = (srt[-1] - srt[-2]).astype(np.float32)
    # dropout disabled: deterministic pass + per-layer checksums for bisecting
    model_nd = _ref_model(do_dropout=False)
    acts = {}
    hs = []
    for name in ["contr_1_1", "contr_1_2", "contr_2_1", "contr_2_2", "contr_3_1", "contr_3_2", "contr_4_1",
                 "contr_4_2", "center", "expand_4_1", "expand_4_2", "upscale4", "expand_3_1", "expand_3_2",
                 "upscale3", "expand_2_1", "expand_2_2", "upscale2", "expand_1_1", "expand_1_2", "final"]:
        hs.append(getattr(model_nd, name).register_forward_hook(
            lambda mod, i, o, name=name: acts.__setitem__(name, o.detach().clone())))
    with torch.no_grad():
        logits_nd = model_nd.forward(x)
    for h in hs:
        h.remove()
    out["logits_nodrop"] = logits_nd[0].numpy().astype(np.float32)
    for name, a in acts.items():
        out[f"chk_{name}"] = np.array([a.mean().item(), a.abs().max().item(), a.std().item()])
    np.savez_compressed(os.path.join(OUT, fname), **out)
    sz = os.path.getsize(os.path.join(OUT, fname))
    print(f"G2 {fname}: {sz / 1e6:.2f} MB; logits range {logits_all.min():.3f}..{logits_all.max():.3f};"
          f" MI max {unc['epistemic_uncertainty'].max():.4f}")


# ----------------------------------------------------------------------------- G3
def gen_ensemble_tta():
    """3 members, TTA identity + 7 flips on [orig, noisy], through the reference's
    concat_data, in the loop order of test_3D.py:417-456.  The noisy input is
    formula data (batchgenerators' GaussianNoiseTransform is third-party and absent)."""
    size = 16
    models = [_ref_model(seed_tag=s, do_dropout=False) for s in range(3)]
    x = torch.from_numpy(formula_volume((1, 1, size, size, size)))
    noise = torch.from_numpy(formula_tensor((1, 1, size, size, size), tag=99, scale=0.1))
    x_noise = (x + noise).float().double()
    dc = DataCarrier3D()
    batch = {"image_paths": ["img0.npy"], "label_paths": [["l0"]], "crop_idx": [((0, size), (0, size), (0, size))],
             "org_image_size": [(size, size, size)], "data": x.clone(),
             "seg": torch.zeros(1, 1, size, size, size, dtype=torch.int32)}
    flip_dims = [(2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]
    n_pred = 2 * len(flip_dims) + 2
    pred_idx = 0
    with torch.no_grad():
        for model in models:
            for xi in [x, x_noise]:
                output = model.forward(xi)
                sm = F.softmax(output, dim=1)
                dc.concat_data(batch=batch, softmax_pred=sm, n_pred=n_pred * len(models), pred_idx=pred_idx)
                pred_idx += 1
                for fd in flip_dims:
                    output = torch.flip(model.forward(torch.flip(xi, fd)), fd)
                    sm = F.softmax(output, dim=1)
                    dc.concat_data(batch=batch, softmax_pred=sm, n_pred=n_pred * len(models), pred_idx=pred_idx)
                    pred_idx += 1
    buf = dc.data["img0.npy"]["softmax_pred"]
    unc = _t(ref_t3.calculate_uncertainty(torch.from_numpy(buf)))
    out = {"input": x.numpy().astype(np.float32), "input_noise": x_noise.numpy().astype(np.float32),
           "softmax_pred": buf.astype(np.float32), "num_predictions": dc.data["img0.npy"]["num_predictions"]}
    out.update(unc)
    mean = np.mean(buf, axis=0)
    out["mean_seg"] = np.argmax(mean, axis=0).astype(np.uint8)
    srt = np.sort(mean, axis=0)
    out["mean_margin"] = (srt[-1] - srt[-2]).astype(np.float32)
    # plain 3-member ensemble, n_pred=1 (config C3 ordering)
    dc2 = DataCarrier3D()
    with torch.no_grad():
        for i, model in enumerate(models):
            sm = F.softmax(model.forward(x), dim=1)
            dc2.concat_data(batch=batch, softmax_pred=sm, n_pred=len(models), pred_idx=i)
    buf2 = dc2.data["img0.npy"]["softmax_pred"]
    out["ens_softmax_pred"] = buf2.astype(np.float32)
    for k, v in _t(ref_t3.calculate_uncertainty(torch.from_numpy(buf2))).items():
        out["ens_" + k] = v
    np.savez_compressed(os.path.join(OUT, "ensemble_tta_16.npz"), **out)
    print("G3 ensemble_tta_16.npz", buf.shape, os.path.getsize(os.path.join(OUT, "ensemble_tta_16.npz")) / 1e6, "MB")


# ----------------------------------------------------------------------------- G4
def gen_patch_index():
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for sub in ("imagesTs", "labelsTs", "images", "labels"):
            os.makedirs(os.path.join(td, sub))
        cases = {"a64": (64, 64, 64), "b128": (128, 128, 128), "c96": (96, 64, 80)}
        for name, shp in cases.items():
            np.save(os.path.join(td, "imagesTs", name + ".npy"), np.zeros(shp, dtype=np.float32))
            np.save(os.path.join(td, "images", name + ".npy"), np.zeros(shp, dtype=np.float32))
        for patch, overlap in [(64, 1), (64, 0.5), (32, 0.5), (32, 1)]:
            for fn, tag in ((ref_samples_toy, "toy"), (ref_samples_lidc, "lidc")):
                samples = fn(base_dir=td, test=True, patch_size=patch, patch_overlap=overlap,
                             subject_ids=[n + ".npy" for n in cases])
                for s in samples:
                    key = f"{tag}|{os.path.basename(s['image_path']).split('.')[0]}|{patch}|{overlap}"
                    res.setdefault(key, []).append([list(c) for c in s["crop_idx"]])
    res["_shapes"] = {k: list(v) for k, v in cases.items()}
    with open(os.path.join(OUT, "patch_index.json"), "w") as f:
        json.dump(res, f)
    print("G4 patch_index.json", {k: len(v) for k, v in res.items() if k.startswith("toy")})

    # accumulate: 32^3 image, patch 16, overlap 0.5, through the reference concat_data
    size, patch = 24, 16
    # the crop LIST of get_val_test_data_samples for this cube (z outermost, x innermost, step int(patch * overlap));
    # patch_index.json above pins the same order against the reference's own function
    step = int(patch * 0.5)
    starts = list(range(0, size - patch + 1, step))
    crops = [((x, x + patch), (y, y + patch), (z, z + patch)) for z in starts for y in starts for x in starts]
    dc = DataCarrier3D()
    T = 3
    fake = formula_tensor((len(crops), T, 2, patch, patch, patch), tag=55, scale=1.0)
    fake = np.abs(fake)
    fake = fake / fake.sum(axis=2, keepdims=True)
    for pi, crop in enumerate(crops):
        batch = {"image_paths": ["img.npy"], "label_paths": [["l0"]], "crop_idx": [crop],
                 "org_image_size": [(size, size, size)], "data": torch.zeros(1, 1, patch, patch, patch),
                 "seg": torch.zeros(1, 1, patch, patch, patch, dtype=torch.int32)}
        for t in range(T):
            dc.concat_data(batch=batch, softmax_pred=torch.from_numpy(fake[pi, t][None]), n_pred=T, pred_idx=t)
    v = dc.data["img.npy"]
    unc = _t(ref_t3.calculate_uncertainty(torch.from_numpy(v["softmax_pred"])))  # on the UN-normalised sum (D10)
    norm = v["softmax_pred"] / np.clip(v["num_predictions"], 1, None)
    out = {"softmax_sum": v["softmax_pred"].astype(np.float32), "num_predictions": v["num_predictions"],
           "normalised": norm.astype(np.float32)}
    out.update({"unc_" + k: a for k, a in unc.items()})
    np.savez_compressed(os.path.join(OUT, "accum_24.npz"), **out)
    print("G4 accum_24.npz", len(crops), "patches; count max", v["num_predictions"].max())


# ----------------------------------------------------------------------------- G6
def gen_agg():
    res = {}
    for size, tag in ((24, 31), (64, 32)):
        img = np.abs(formula_tensor((size, size, size), tag=tag, scale=0.7)).astype(np.float32)
        # add a hot blob so the max patch is unique
        c = size // 3
        img[c:c + 6, c + 2:c + 9, c + 1:c + 7] += 0.5
        key = f"vol{size}"
        r = {}
        r["patch10"] = ref_agg.patch_level_aggregation(img, patch_size=10)
        r["patch10_mean"] = ref_agg.patch_level_aggregation(img, patch_size=10, mean=True)
        r["patch_5_7_9"] = ref_agg.patch_level_aggregation(img, patch_size=[5, 7, 9])
        r["image"] = ref_agg.image_level_aggregation(img)
        r["image_mean"] = ref_agg.image_level_aggregation(img, mean=True)
        for thr in (0.3, 0.6, 5.0):
            for mean in (True, False):
                t = ref_agg.threshold_aggregation(img, threshold=thr, mean=mean)
                r[f"thr_{thr}_{int(mean)}"] = {k: float(v) for k, v in t.items()}
        res[key] = r
        res[key + "_tag"] = tag
    # 2D map too (the aggregation is dimension-generic, patch_size int -> per-dim list)
    img2 = np.abs(formula_tensor((40, 56), tag=33, scale=1.0)).astype(np.float32)
    res["img2d"] = {"patch10": ref_agg.patch_level_aggregation(img2, patch_size=10),
                    "image": ref_agg.image_level_aggregation(img2)}
    with open(os.path.join(OUT, "agg_kat.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("G6 agg_kat.json", res["vol24"]["patch10"])


# ----------------------------------------------------------------------------- G5
class _Cfg(dict):
    """attribute + item access, like the yacs/omegaconf object HighResolutionNet expects (hrnet_module.py:343-346)."""
    def __getattr__(self, k):
        v = self[k]
        return _Cfg(v) if isinstance(v, dict) else v


def gen_hrnet_ssn():
    """G9 hrnet_ssn.npz: the reference HighResolutionNet with the SSN head (hrnet_module.py:430-453, 559-595; config
    keys of hrnet_config_ssn.yaml) at the small stage layout, training-mode BN, formula weights; distribution.sample as
    test_2D.py:285-299 calls it with the normals of LowRankMultivariateNormal.rsample captured."""
    import copy
    import torch.distributions.lowrank_multivariate_normal as lrm
    import uncertainty_modeling.models.hrnet_module as ref_hr
    from tests.formula import HRNET_SMALL_EXTRA, formula_state_dict_from_shapes
    extra = copy.deepcopy(HRNET_SMALL_EXTRA)
    extra["DROPOUT_FINAL"] = False
    ncls, R, S = 4, 10, 2
    cfg = _Cfg({"MODEL": {"EXTRA": extra, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3, "PRETRAINED": False,
                          "SSN": True, "SSN_RANK": R, "SSN_EPS": 1e-5},
                "DATASET": {"NUM_CLASSES": ncls}})
    model = ref_hr.HighResolutionNet(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = formula_state_dict_from_shapes(shapes)
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v).float()
    model.load_state_dict(full)
    x = torch.from_numpy(formula_tensor((2, 3, 64, 96), tag=83, scale=1.5)).float()
    torch.set_grad_enabled(False)
    dist = model.forward(x)
    drawn = []
    orig = lrm._standard_normal

    def fake_normal(shape, dtype, device):
        t = torch.from_numpy(formula_tensor(tuple(shape), tag=9100 + len(drawn), scale=1.7)).to(dtype)
        drawn.append(t)
        return t

    lrm._standard_normal = fake_normal
    try:
        samples = dist.sample([S])            # (S, B, C*H*W)
    finally:
        lrm._standard_normal = orig
        torch.set_grad_enabled(True)
    out = {"input": x.numpy(), "loc": dist.loc.numpy(), "cov_diag": dist.cov_diag.numpy(),
           "eps_w": drawn[0].numpy().astype(np.float32), "eps_d_tag": np.array(9101), "samples": samples.numpy(),
           "cov_factor_probe": dist.cov_factor[:, ::997].numpy(),      # every 997th row of (B, C*H*W, R)
           "shapes_json": np.frombuffer(json.dumps({k: list(v) for k, v in shapes.items()}).encode(), dtype=np.uint8)}
    assert tuple(drawn[1].shape) == tuple(samples.shape)
    np.savez_compressed(os.path.join(OUT, "hrnet_ssn.npz"), **out)
    print("G9 hrnet_ssn.npz", samples.shape, os.path.getsize(os.path.join(OUT, "hrnet_ssn.npz")) / 1e6, "MB; sample range",
          float(samples.min()), float(samples.max()), "diag range", float(dist.cov_diag.min()), float(dist.cov_diag.max()))


def gen_hrnet_w18():
    """G10 hrnet_w18s.npz: the reference HighResolutionNet at the HRNet-W18 widths (18/36/72/144, 270 concatenated;
    BASELINE config 4), 5 classes, training-mode BN, two passes with DROPOUT_FINAL masks captured + one without."""
    import copy
    import uncertainty_modeling.models.hrnet_module as ref_hr
    from tests.formula import HRNET_W18S_EXTRA, formula_state_dict_from_shapes
    extra = copy.deepcopy(HRNET_W18S_EXTRA)
    ncls = 5
    cfg = _Cfg({"MODEL": {"EXTRA": extra, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3, "PRETRAINED": False},
                "DATASET": {"NUM_CLASSES": ncls}})
    model = ref_hr.HighResolutionNet(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    full = model.state_dict()
    for k, v in formula_state_dict_from_shapes(shapes).items():
        full[k] = torch.from_numpy(v).float()
    model.load_state_dict(full)
    x = torch.from_numpy(formula_tensor((2, 3, 64, 96), tag=85, scale=1.5)).float()
    masks = []
    orig = F.dropout

    def spy(inp, p=0.5, training=True, inplace=False):
        out = orig(inp, p, training, inplace)
        masks.append(((out != 0) | (inp == 0)).clone())
        return out

    ref_hr.F.dropout = spy
    out = {"input": x.numpy(), "shapes_json": np.frombuffer(json.dumps({k: list(v) for k, v in shapes.items()}).encode(), dtype=np.uint8)}
    torch.set_grad_enabled(False)
    logits = []
    for t in range(2):
        masks.clear()
        logits.append(model.forward(x).numpy().copy())
        for i, m in enumerate(masks):
            out[f"mask_{t}_{i}"] = np.packbits(m.numpy().astype(np.uint8).ravel())
            out[f"maskshape_{i}"] = np.array(m.shape)
    ref_hr.F.dropout = orig
    out["logits"] = np.stack(logits)
    extra2 = copy.deepcopy(extra)
    extra2["DROPOUT_FINAL"] = False
    cfg2 = _Cfg({"MODEL": {"EXTRA": extra2, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3, "PRETRAINED": False},
                 "DATASET": {"NUM_CLASSES": ncls}})
    m2 = ref_hr.HighResolutionNet(cfg2)
    m2.load_state_dict(full)
    out["logits_nodrop"] = m2.forward(x).numpy()
    torch.set_grad_enabled(True)
    np.savez_compressed(os.path.join(OUT, "hrnet_w18s.npz"), **out)
    print("G10 hrnet_w18s.npz", out["logits"].shape, os.path.getsize(os.path.join(OUT, "hrnet_w18s.npz")) / 1e6, "MB; logit range",
          float(out["logits"].min()), float(out["logits"].max()))


def gen_hrnet_w18_full():
    """G11 hrnet_w18_256x478.npz: the reference HighResolutionNet in the FULL HRNet-W18 layout (BASELINE config 4:
    widths 18/36/72/144, 1/4/3 modules of 4 blocks, 19 classes, no DROPOUT_FINAL) on ONE 3 x 256 x 478 image -- the size
    the reference's test images have -- in training-mode BatchNorm as the reference runs it.  Kept small:
    the float32 logits on a stride-(4, 6) sub-grid, their float64 row and column sums over the WHOLE map (every pixel
    is pinned through them), the same from a float64 run of the reference class (its own float32-vs-float64 gap), and
    the VerticalFlip / HorizontalFlip views' logits on the sub-grid for the 8-view TTA of config 4."""
    import copy
    import uncertainty_modeling.models.hrnet_module as ref_hr
    from tests.formula import formula_state_dict_from_shapes
    from values_amd.hrnet_configs import hrnet_w18_extra
    extra = copy.deepcopy(hrnet_w18_extra(False))
    ncls = 19
    cfg = _Cfg({"MODEL": {"EXTRA": extra, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3, "PRETRAINED": False},
                "DATASET": {"NUM_CLASSES": ncls}})
    model = ref_hr.HighResolutionNet(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    full = model.state_dict()
    for k, v in formula_state_dict_from_shapes(shapes).items():
        full[k] = torch.from_numpy(v).float()
    model.load_state_dict(full)
    x = torch.from_numpy(formula_tensor((1, 3, 256, 478), tag=87, scale=1.5)).float()
    torch.set_grad_enabled(False)
    out = {"shapes_json": np.frombuffer(json.dumps({k: list(v) for k, v in shapes.items()}).encode(), dtype=np.uint8),
           "input_tag": np.array(87), "input_scale": np.array(1.5)}
    sy, sx = 4, 6

    def put(tag, y):
        y = y[0]
        out[f"{tag}_sub"] = y[:, ::sy, ::sx].astype(np.float32)
        out[f"{tag}_rowsum"] = y.astype(np.float64).sum(2)
        out[f"{tag}_colsum"] = y.astype(np.float64).sum(1)
        out[f"{tag}_absmax"] = np.array(np.abs(y).max())

    y32 = model.forward(x).numpy()
    put("logits", y32)
    put("logits_hflip", torch.flip(model.forward(torch.flip(x, [-1])), [-1]).numpy())      # test_2D.py:304-309
    put("logits_vflip", torch.flip(model.forward(torch.flip(x, [-2])), [-2]).numpy())      # the 8-view extension
    m64 = ref_hr.HighResolutionNet(cfg).double()
    m64.load_state_dict({k: v.double() for k, v in full.items()})
    y64 = m64.forward(x.double()).numpy()
    put("logits64", y64)
    out["ref_f32_f64_gap"] = np.array(np.abs(y32 - y64).max())
    torch.set_grad_enabled(True)
    path = os.path.join(OUT, "hrnet_w18_256x478.npz")
    np.savez_compressed(path, **out)
    print("G11 hrnet_w18_256x478.npz", y32.shape, os.path.getsize(path) / 1e6, "MB; logit range", float(y32.min()), float(y32.max()),
          "reference f32-vs-f64 gap", float(out["ref_f32_f64_gap"]))


def gen_evalmetrics():
    """G12 evalmetrics_kat.npz: the evaluation stage's downstream scalars from the reference's own modules --
    evaluation/metrics/aurc.py (rc_curve_stats, aurc, eaurc), ncc.py (compute_ncc), ace.py (platt_scale_confid,
    calib_stats, calc_ace, and the Platt fit it imports from the INSTALLED scikit-learn: the reference pins 1.2.2, this
    container has another version -- same objective, other optimiser), auroc.py's roc_curve + auc call."""
    import sklearn
    import evaluation.metrics.aurc as ref_aurc
    import evaluation.metrics.ncc as ref_ncc
    import evaluation.metrics.ace as ref_ace
    from sklearn.metrics import auc, roc_curve
    rng = np.random.default_rng(7)
    out = {"sklearn_version": np.frombuffer(sklearn.__version__.encode(), dtype=np.uint8)}
    # --- AURC / E-AURC: 40 images, ties in the confidences
    risks = rng.random(40)
    confids = -np.round(rng.random(40) * 12) / 12 - 0.3 * (1 - risks)
    confids[5] = confids[9] = confids[17]
    out.update(aurc_risks=risks, aurc_confids=confids, aurc=np.array(ref_aurc.aurc(risks, confids)),
               eaurc=np.array(ref_aurc.eaurc(risks, confids)))
    cov, sel, wts = ref_aurc.rc_curve_stats(risks, confids)
    out.update(rc_coverages=np.array(cov), rc_risks=np.array(sel), rc_weights=np.array(wts))
    # --- NCC: rater variance (float64) against a float32 uncertainty map
    segs = (rng.random((4, 12, 14, 10)) < 0.35).astype(np.int64)
    gt_unc = np.var(segs, axis=0)
    pred_unc = (0.6 * gt_unc + 0.1 * rng.random(gt_unc.shape)).astype(np.float32)
    out.update(ncc_gt=gt_unc, ncc_pred=pred_unc, ncc=np.array(ref_ncc.compute_ncc(gt_unc, pred_unc)))
    # --- ACE: 3 raters, labels {0, 1, 2}; with and without ignore_value = 2
    ref = rng.integers(0, 3, size=(3, 12, 14, 10))
    pred = np.where(rng.random((12, 14, 10)) < 0.7, ref[0], rng.integers(0, 3, size=(12, 14, 10)))
    agree = (ref == pred[None]).mean(0)
    unc = (0.7 * (1 - agree) + 0.3 * rng.random(pred.shape)).astype(np.float32) * 0.69
    out.update(ace_ref=ref.astype(np.int32), ace_pred=pred.astype(np.int32), ace_unc=unc)
    for tag, ign in (("all", None), ("ign2", 2)):
        p3 = np.repeat(pred[np.newaxis, :], 3, 0)
        u3 = np.repeat(unc[np.newaxis, :], 3, 0)
        correct = (ref == p3).astype(int)
        if ign is not None:
            keep = ref != ign
            a, b = ref_ace.calib(-u3[keep], correct[keep])
            conf = 1 / (1 + np.exp(-u3[keep] * a + b))           # platt_scale_confid's formula (it reads a JSON file)
            d, w, k = ref_ace.calib_stats(correct[keep], conf)
            ace = ref_ace.calc_ace(correct[keep], conf)
        else:
            a, b = ref_ace.calib(-u3.flatten(), correct.flatten())
            conf = 1 / (1 + np.exp(-u3.flatten() * a + b))
            d, w, k = ref_ace.calib_stats(correct.flatten(), conf)
            ace = ref_ace.calc_ace(correct.flatten(), conf)
        out.update({f"ace_{tag}_a": np.array(a), f"ace_{tag}_b": np.array(b), f"ace_{tag}_disc": d, f"ace_{tag}_w": w,
                    f"ace_{tag}_k": np.array(k), f"ace_{tag}": np.array(ace)})
    # the one-label case (every voxel correct): label_binarize gives zeros
    conf1 = 1 / (1 + np.exp(-unc.flatten() * 2.0 - 1.0))
    out.update(ace_onelabel=np.array(ref_ace.calc_ace(np.ones(unc.size, dtype=int), conf1)))
    # --- AUROC as auroc.py:126-127 computes it
    y = (rng.random(60) < 0.35).astype(int)
    sc = np.round(rng.random(60) * 20) / 20 + 0.4 * y
    fpr, tpr, _ = roc_curve(y, sc)
    out.update(auroc_y=y, auroc_score=sc, auroc=np.array(auc(fpr, tpr)))
    path = os.path.join(OUT, "evalmetrics_kat.npz")
    np.savez_compressed(path, **out)
    print("G12 evalmetrics_kat.npz", os.path.getsize(path) / 1e3, "KB; aurc", float(out["aurc"]), "eaurc", float(out["eaurc"]),
          "ncc", float(out["ncc"]), "ace", float(out["ace_all"]), float(out["ace_ign2"]), "a,b", float(out["ace_all_a"]),
          float(out["ace_all_b"]), "auroc", float(out["auroc"]), "sklearn", sklearn.__version__)


def gen_ssn():
    """G7 ssn_16.npz: the reference SsnUNet3D (ssn_unet3D_module.py) + distribution.sample as predict_cases_ssn
    calls it (test_3D.py:373-385), with the standard normals of LowRankMultivariateNormal.rsample replaced by
    formula tensors (captured), then softmax + calculate_uncertainty(ssn=True)."""
    import torch.distributions.lowrank_multivariate_normal as lrm
    from uncertainty_modeling.models.ssn_unet3D_module import SsnUNet3D as RefSsn
    from tests.formula import formula_ssn_state_dict
    NC, R, S, size = 2, 10, 3, 16
    model = RefSsn(num_classes=NC, rank=R)
    sd = formula_ssn_state_dict(NC, R)
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    model = model.double()  # test_3D.py:367
    x = torch.from_numpy(formula_volume((1, 1, size, size, size)))
    with torch.no_grad():
        dist = model.forward(x)
    drawn = []
    orig = lrm._standard_normal

    def fake_normal(shape, dtype, device):
        t = torch.from_numpy(formula_tensor(tuple(shape), tag=9000 + len(drawn), scale=1.7)).to(dtype)
        drawn.append(t)
        return t

    lrm._standard_normal = fake_normal
    try:
        samples = dist.sample([S])   # (S, 1, NC * vox)
    finally:
        lrm._standard_normal = orig
    assert len(drawn) == 2 and drawn[0].shape[-1] == R
    out = {"input": x.numpy().astype(np.float32),
           "loc": dist.loc.numpy().astype(np.float32), "cov_diag": dist.cov_diag.numpy().astype(np.float32),
           "cov_factor": dist.cov_factor.numpy().astype(np.float32),           # (1, NC * vox, R)
           "eps_w": drawn[0].numpy().astype(np.float32), "eps_d": drawn[1].numpy().astype(np.float32),
           "samples": samples.numpy().astype(np.float32)}
    vol = samples.view([S, 1, NC, size, size, size])
    sm = torch.stack([F.softmax(v, dim=1)[0] for v in vol])      # test_3D.py:386-388 -> buffer (S, NC, ...)
    unc = _t(ref_t3.calculate_uncertainty(torch.from_numpy(sm.numpy().astype(np.float64)), ssn=True))
    out.update(unc)
    np.savez_compressed(os.path.join(OUT, "ssn_16.npz"), **out)
    print("G7 ssn_16.npz", os.path.getsize(os.path.join(OUT, "ssn_16.npz")) / 1e6, "MB; sample range",
          float(samples.min()), float(samples.max()))


def gen_metrics():
    """G8 metrics_kat.npz: the reference's SoftDiceLoss (loss_modules.py) + torch NLLLoss exactly as
    calculate_test_metrics combines them (test_3D.py:262-273), on a formula mean-softmax (1, C, 12,10,8) and R = 3
    formula raters, for C = 2 and C = 3.  (The Dice/GED half needs torchmetrics, absent here: unpinned.)"""
    from loss_modules import SoftDiceLoss
    out = {}
    for C in (2, 3):
        logits = torch.from_numpy(formula_tensor((1, C, 12, 10, 8), 7100 + C, scale=2.0))
        sm = F.softmax(logits, dim=1)
        gt = torch.from_numpy(((formula_tensor((3, 12, 10, 8), 7200 + C) + 1.0) * 0.5 * C).astype(np.int64).clip(0, C - 1))
        losses = []
        for r in range(gt.shape[0]):
            g = torch.unsqueeze(gt[r], 0).type(torch.LongTensor)
            losses.append((SoftDiceLoss()(sm, g) + torch.nn.NLLLoss()(torch.log(sm), g)).item())
        out[f"softmax_{C}"] = sm.numpy()
        out[f"gt_{C}"] = gt.numpy().astype(np.uint8)
        out[f"loss_per_rater_{C}"] = np.array(losses)
        out[f"loss_{C}"] = np.mean(np.array(losses))
    # Cross-check material for the hard Dice (NOT a pin: torchmetrics 0.11.4, the reference's dependency, is absent).
    # scikit-learn's micro-averaged F1 over the labels that are not ignored is an independent implementation of the
    # same published quantity 2 TP / (2 TP + FP + FN) -- and of the same treatment of an ignored class (its column
    # dropped, mistakes INTO it still counted): tests compare oracle/metrics_oracle.tm_dice with it.
    from sklearn.metrics import f1_score
    rng = np.random.default_rng(11)
    for C in (2, 3, 5):
        a = rng.integers(0, C, size=(4, 6, 5, 4))
        b = np.where(rng.random(a.shape) < 0.6, a, rng.integers(0, C, size=a.shape))
        out[f"xc_pred_{C}"], out[f"xc_gt_{C}"] = a.astype(np.uint8), b.astype(np.uint8)
        out[f"xc_f1_all_{C}"] = np.array(f1_score(b.ravel(), a.ravel(), labels=list(range(C)), average="micro", zero_division=0))
        out[f"xc_f1_ign0_{C}"] = np.array(f1_score(b.ravel(), a.ravel(), labels=list(range(1, C)), average="micro", zero_division=0))
    np.savez_compressed(os.path.join(OUT, "metrics_kat.npz"), **out)
    print("G8 metrics_kat.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("loss")})


def gen_hrnet():
    import copy
    import uncertainty_modeling.models.hrnet_module as ref_hr
    from tests.formula import HRNET_SMALL_EXTRA, formula_state_dict_from_shapes
    extra = copy.deepcopy(HRNET_SMALL_EXTRA)
    ncls = 4
    cfg = _Cfg({"MODEL": {"EXTRA": extra, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3, "PRETRAINED": False},
                "DATASET": {"NUM_CLASSES": ncls}})
    torch.manual_seed(7)
    model = ref_hr.HighResolutionNet(cfg)  # stays in training mode like the reference
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = formula_state_dict_from_shapes(shapes)
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v).float()
    model.load_state_dict(full)
    x = torch.from_numpy(formula_tensor((2, 3, 64, 96), tag=81, scale=1.5)).float()
    # capture the 4 F.dropout keep-masks of every pass
    masks = []
    orig = F.dropout

    def spy(inp, p=0.5, training=True, inplace=False):
        out = orig(inp, p, training, inplace)
        masks.append(((out != 0) | (inp == 0)).clone())
        return out

    ref_hr.F.dropout = spy
    out = {"input": x.numpy(), "shapes_json": np.frombuffer(json.dumps({k: list(v) for k, v in shapes.items()}).encode(), dtype=np.uint8)}
    T = 3
    logits = []
    torch.set_grad_enabled(False)  # test_2D.py:329
    for t in range(T):
        masks.clear()
        y = model.forward(x)
        logits.append(y.numpy().copy())
        for i, m in enumerate(masks):
            out[f"mask_{t}_{i}"] = np.packbits(m.numpy().astype(np.uint8).ravel())
            out[f"maskshape_{i}"] = np.array(m.shape)
    ref_hr.F.dropout = orig
    logits = np.stack(logits)  # (T, B, C, H, W)
    out["logits"] = logits
    sm = F.softmax(torch.from_numpy(logits), dim=2)
    # process_output (test_2D.py:205-248): zero channel appended, per-image calculate_uncertainty on (T, C+1, H, W) f32
    sm1 = torch.cat([sm, torch.zeros(T, 2, 1, 64, 96)], dim=2)
    for b in range(2):
        unc = _t(ref_t3.calculate_uncertainty(sm1[:, b]))
        for k, v in unc.items():
            out[f"{k}_{b}"] = v
    # dropout off (eval-free): DROPOUT_FINAL False -> deterministic forward, plus stage-4 feature checksums
    extra2 = copy.deepcopy(extra)
    extra2["DROPOUT_FINAL"] = False
    cfg2 = _Cfg({"MODEL": {"EXTRA": extra2, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3, "PRETRAINED": False},
                 "DATASET": {"NUM_CLASSES": ncls}})
    m2 = ref_hr.HighResolutionNet(cfg2)
    m2.load_state_dict(full)
    out["logits_nodrop"] = m2.forward(x).numpy()
    torch.set_grad_enabled(True)
    np.savez_compressed(os.path.join(OUT, "hrnet_small.npz"), **out)
    print("G5 hrnet_small.npz", logits.shape, os.path.getsize(os.path.join(OUT, "hrnet_small.npz")) / 1e6, "MB; params",
          sum(int(np.prod(s)) for s in shapes.values()), "logit range", float(logits.min()), float(logits.max()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["unc", "unet16", "unet32", "tta", "patch", "agg", "hrnet", "ssn", "metrics", "hrnet_ssn", "hrnet_w18", "hrnet_w18_full", "evalmetrics"]
    if "unc" in which:
        gen_unc_kat()
    if "unet16" in which:
        gen_unet(16, 4, "unet3d_16.npz")
    if "unet32" in which:
        gen_unet(32, 4, "unet3d_32.npz")
    if "tta" in which:
        gen_ensemble_tta()
    if "patch" in which:
        gen_patch_index()
    if "agg" in which:
        gen_agg()
    if "hrnet" in which:
        gen_hrnet()
    if "ssn" in which:
        gen_ssn()
    if "metrics" in which:
        gen_metrics()
    if "hrnet_ssn" in which:
        gen_hrnet_ssn()
    if "hrnet_w18" in which:
        gen_hrnet_w18()
    if "hrnet_w18_full" in which:
        gen_hrnet_w18_full()
    if "evalmetrics" in which:
        gen_evalmetrics()
