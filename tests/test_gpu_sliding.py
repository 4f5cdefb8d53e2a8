"""GPU: sliding-window accumulation (K14), the whole-image driver and the checkpoint loader."""
import numpy as np
import pytest
import torch

from tests.helpers import formula_sd_torch, load_npz
from tests.formula import formula_tensor, formula_unet3d_state_dict, formula_volume

pytestmark = pytest.mark.gpu
KEYS = ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty")


def test_accumulate_matches_reference_concat_data():
    """tests/golden/accum_24.npz was filled by the reference's own DataCarrier3D.concat_data (24^3 image, patch 16,
    overlap 0.5: 8 overlapping patches, count up to 8)."""
    from values_amd import _lib, crop_indices, uncertainty_maps
    lib = _lib.load()
    g = load_npz("accum_24.npz")
    size, patch, T = 24, 16, 3
    crops = crop_indices((size,) * 3, patch, 0.5)
    fake = np.abs(formula_tensor((len(crops), T, 2, patch, patch, patch), tag=55, scale=1.0))
    fake = fake / fake.sum(axis=2, keepdims=True)
    logits = torch.from_numpy(np.log(fake)).float().cuda()  # softmax(log p) == p
    ssum = torch.zeros((T, 2, size, size, size), device="cuda")
    count = torch.zeros((size,) * 3, device="cuda")
    crop_t = torch.tensor([[c[0][0], c[1][0], c[2][0]] for c in crops], dtype=torch.int32, device="cuda")
    for b0 in range(0, len(crops), 3):  # uneven batches
        nb = min(3, len(crops) - b0)
        _lib.check(lib.vx_softmax_accumulate(_lib.ptr(logits[b0:b0 + nb].contiguous()), nb, T, 2, patch, patch, patch,
                                             _lib.ptr(crop_t[b0:b0 + nb].contiguous()), _lib.ptr(ssum), _lib.ptr(count),
                                             size, size, size, 1, _lib.stream_ptr()), "acc")
    torch.cuda.synchronize()
    np.testing.assert_allclose(ssum.cpu().numpy(), g["softmax_sum"], atol=2e-6)
    np.testing.assert_array_equal(count.cpu().numpy(), g["num_predictions"][0])
    norm = (ssum / count.clamp(min=1)).cpu().numpy()
    np.testing.assert_allclose(norm, g["normalised"], atol=1e-6)
    m = uncertainty_maps(ssum.unsqueeze(0))  # D10: on the un-normalised sums
    np.testing.assert_allclose(m["pred_entropy"][0].cpu().numpy(), g["unc_pred_entropy"], atol=3e-5)
    np.testing.assert_allclose(m["expected_entropy"][0].cpu().numpy(), g["unc_aleatoric_uncertainty"], atol=3e-5)
    np.testing.assert_allclose(m["mutual_information"][0].cpu().numpy(), g["unc_epistemic_uncertainty"], atol=3e-5)


@pytest.mark.parametrize("overlap", [1, 0.5])
def test_sliding_window_vs_oracle(overlap):
    from oracle import predict_oracle as po
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import unet3d_forward
    from tests.test_gpu_unet3d import make_model
    from values_amd import predict_image_sliding
    size, patch = 32, 16
    img = formula_volume((size, size, size), tag=41)
    sds = [formula_sd_torch(seed_tag=s) for s in range(2)]
    models = [make_model(seed_tag=s, do_dropout=False) for s in range(2)]
    out = predict_image_sliding(models, torch.from_numpy(img), patch_size=patch, patch_overlap=overlap, n_pred=1,
                                patch_batch=5)
    crops = po.crop_indices((size,) * 3, patch, overlap)
    acc = po.Accumulator(2, (size,) * 3)
    with torch.no_grad():
        for crop in crops:
            x = torch.from_numpy(img[crop[0][0]:crop[0][1], crop[1][0]:crop[1][1], crop[2][0]:crop[2][1]].copy())[None, None]
            for mi, sd in enumerate(sds):
                acc.add(crop, po.softmax_np(unet3d_forward(sd, x).numpy())[0], mi)
    ref = uo.calculate_uncertainty(acc.softmax_pred)
    cl = np.clip(acc.num_predictions, 1, None)[0]
    np.testing.assert_array_equal(out["num_predictions"].cpu().numpy(), acc.num_predictions[0])
    np.testing.assert_allclose(out["softmax_sum"].cpu().numpy(), acc.softmax_pred, atol=5e-5)
    for k in KEYS:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k] / cl, atol=1e-4)
    mean = (acc.softmax_pred / np.clip(acc.num_predictions, 1, None)).mean(0)
    np.testing.assert_allclose(out["mean_softmax"].cpu().numpy(), mean, atol=5e-5)
    srt = np.sort(mean, axis=0)
    clear = (srt[-1] - srt[-2]) > 1e-5
    assert (out["pred_seg_mean"].cpu().numpy() == mean.argmax(0))[clear].all()


def test_image_not_covered_by_patches_keeps_zero_count():
    """a (48, 32, 40) image with patch 32 / overlap 1 gets ONE patch (like the reference's while loops); the rest
    of the image keeps count 0, zero maps and class 0."""
    from tests.test_gpu_unet3d import make_model
    from values_amd import predict_image_sliding
    model = make_model(do_dropout=True)
    img = torch.from_numpy(formula_volume((48, 32, 40), tag=42))
    out = predict_image_sliding([model], img, patch_size=32, patch_overlap=1, n_pred=3, seeds=[1])
    cnt = out["num_predictions"].cpu().numpy()
    assert cnt[:32, :32, :32].min() == 1 and cnt.sum() == 32 ** 3
    assert out["pred_entropy"][32:].abs().max().item() == 0
    assert out["pred_seg_mean"][32:].max().item() == 0
    assert out["epistemic_uncertainty"][:32, :32, :32].max().item() > 0


def test_load_models_from_checkpoint_like_test_3D():
    from oracle.unet3d_oracle import unet3d_forward
    from values_amd import load_models_from_checkpoint
    sd = formula_unet3d_state_dict(seed_tag=3)
    ck = {"state_dict": {"model." + k: torch.from_numpy(v).float() for k, v in sd.items()},
          "hyper_parameters": {"model": {"_target_": "uncertainty_modeling.models.unet3D_module.UNet3D", "num_classes": 2,
                                         "do_dropout": False}, "seed": 123, "aleatoric_loss": None}}
    models = load_models_from_checkpoint([ck, ck])
    assert len(models) == 2 and type(models[0]).__module__ == "values_amd.unet3d"
    x = torch.from_numpy(formula_volume((1, 1, 16, 16, 16), tag=43))
    with torch.no_grad():
        ref = unet3d_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x).numpy()
        got = models[0](x.float().cuda()).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4
    ck2 = dict(ck)
    ck2["hyper_parameters"] = dict(ck["hyper_parameters"], aleatoric_loss=True)
    ck2["state_dict"] = dict(ck["state_dict"])
    ck2["state_dict"]["model.final_aleatoric.weight"] = torch.zeros(4, 8, 1, 1, 1)
    ck2["state_dict"]["model.final_aleatoric.bias"] = torch.zeros(4)
    m2 = load_models_from_checkpoint([ck2])[0]
    assert m2.aleatoric_loss is True


def test_sliding_with_aleatoric_head_takes_its_pass_count_from_the_logits():
    """An aleatoric_loss checkpoint with the default n_pred=1 makes n_aleatoric_samples passes per patch
    (test_3D.py:458-469 sets n_pred := n_aleatoric_samples); the accumulation buffers must follow that, patch by patch.
    Checked against the oracle with the same noise."""
    from oracle import predict_oracle as po
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import unet3d_forward
    from values_amd import UNet3D, predict_image_sliding
    sd = formula_unet3d_state_dict(seed_tag=4, aleatoric_loss=True)
    model = UNet3D(num_classes=2, aleatoric_loss=True)
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    model = model.cuda()
    size, patch, Ts = 32, 16, 4
    img = formula_volume((size, size, size), tag=47)
    crops = po.crop_indices((size,) * 3, patch, 1)
    eps = torch.from_numpy(formula_tensor((len(crops), Ts, 2, patch, patch, patch), 48, scale=1.3))
    out = predict_image_sliding([model], torch.from_numpy(img), patch_size=patch, patch_overlap=1, n_pred=1,
                                n_aleatoric_samples=Ts, patch_batch=len(crops), eps=[eps])
    assert out["softmax_sum"].shape[0] == Ts
    acc = po.Accumulator(Ts, (size,) * 3)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        for b, crop in enumerate(crops):
            x = torch.from_numpy(img[crop[0][0]:crop[0][1], crop[1][0]:crop[1][1], crop[2][0]:crop[2][1]].copy())[None, None]
            mu, s = unet3d_forward(sdt, x, aleatoric_loss=True, num_classes=2)
            for t in range(Ts):
                smp = mu.numpy() + np.exp(s.numpy() / 2) * eps[b, t].numpy()[None]
                acc.add(crop, po.softmax_np(smp)[0], t)
    ref = uo.calculate_uncertainty(acc.softmax_pred)
    np.testing.assert_allclose(out["softmax_sum"].cpu().numpy(), acc.softmax_pred, atol=5e-5)
    for k in KEYS:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], atol=1e-4)
    # ssn=True swaps the two maps like calculate_uncertainty(ssn=True) (test_3D.py:510-516)
    sw = predict_image_sliding([model], torch.from_numpy(img), patch_size=patch, patch_overlap=1, n_aleatoric_samples=Ts,
                               patch_batch=len(crops), eps=[eps], ssn=True)
    assert torch.equal(sw["aleatoric_uncertainty"], out["epistemic_uncertainty"])
    assert torch.equal(sw["epistemic_uncertainty"], out["aleatoric_uncertainty"])


@pytest.mark.parametrize("overlap,npatch", [(1, 8), (0.5, 27)])
def test_config_C5_full_size_128_T20(overlap, npatch):
    """BASELINE config C5 as worded: 128^3 image, 64^3 patches, T = 20 MC-dropout passes (hash dropout), MI maps.
    Size-independent properties (count map, MI = PE - EE on the sums, bit-identical reruns, normalised mode), and for
    overlap 1 the equality with the patch batch pushed through predict_uncertainty."""
    from tests.test_gpu_unet3d import make_model
    from values_amd import crop_indices, predict_image_sliding, predict_uncertainty
    S, P, T = 128, 64, 20
    model = make_model(do_dropout=True)
    img = torch.from_numpy(formula_volume((S, S, S), tag=52)).float().cuda()
    crops = crop_indices((S, S, S), P, overlap)
    assert len(crops) == npatch
    a = predict_image_sliding([model], img, patch_size=P, patch_overlap=overlap, n_pred=T, patch_batch=8, seeds=[5])
    b = predict_image_sliding([model], img, patch_size=P, patch_overlap=overlap, n_pred=T, patch_batch=8, seeds=[5])
    assert a["softmax_sum"].shape == (T, 2, S, S, S)
    for k in KEYS + ("mean_softmax", "softmax_variance"):
        assert not torch.isnan(a[k]).any(), k
        if overlap == 1:
            assert torch.equal(a[k], b[k]), k          # no atomics when the patches of a launch are disjoint
        else:
            assert (a[k] - b[k]).abs().max().item() < 1e-5, k   # float atomics: summation order only
    # count map = number of patches covering a voxel (the reference's num_predictions, data_carrier_3D.py:170-179)
    c1 = np.zeros(S)
    step = int(P * overlap)
    for s0 in range(0, S - P + 1, step):
        c1[s0:s0 + P] += 1
    want = c1[:, None, None] * c1[None, :, None] * c1[None, None, :]
    np.testing.assert_array_equal(a["num_predictions"].cpu().numpy(), want)
    cl = torch.from_numpy(np.clip(want, 1, None)).float().cuda()
    # every pass is a softmax: the un-normalised sums add up to the count
    assert (a["softmax_sum"].sum(1) - cl[None]).abs().max().item() < 1e-4
    assert (a["mean_softmax"].sum(0) - 1).abs().max().item() < 1e-5
    mi, pe, ee = a["epistemic_uncertainty"], a["pred_entropy"], a["aleatoric_uncertainty"]
    assert (mi - (pe - ee)).abs().max().item() < 1e-6
    assert a["softmax_variance"].min().item() >= 0 and a["softmax_variance"].max().item() <= 0.25 + 1e-6
    assert mi.mean().item() > 1e-5                     # the T samples really differ
    n = predict_image_sliding([model], img, patch_size=P, patch_overlap=overlap, n_pred=T, patch_batch=8, seeds=[5],
                              compat=False)
    if overlap == 1:
        for k in KEYS:
            assert torch.equal(n[k], a[k]), k          # count == 1: compat (quirk D10) and normalised agree
        # the same patches as ONE batch through the fused logit reduction: same logits, same maps to rounding
        x = torch.stack([img[c[0][0]:c[0][1], c[1][0]:c[1][1], c[2][0]:c[2][1]] for c in crops]).unsqueeze(1)
        pu = predict_uncertainty([model], x, n_pred=T, seeds=[5])
        for bi, c in enumerate(crops):
            sl = (slice(*c[0]), slice(*c[1]), slice(*c[2]))
            for k in KEYS:
                assert (a[k][sl] - pu[k][bi]).abs().max().item() < 2e-6, (k, bi)
            assert torch.equal(a["pred_seg_mean"][sl], pu["pred_seg_mean"][bi]) or \
                (a["mean_softmax"][(slice(None),) + sl] - pu["mean_softmax"][bi]).abs().max().item() < 1e-6
    else:
        # normalised mode: maps of probabilities -> entropy bounds hold (they do not on un-normalised sums)
        assert n["pred_entropy"].max().item() <= float(np.log(2)) + 1e-5 and n["pred_entropy"].min().item() >= -1e-6
        assert n["epistemic_uncertainty"].min().item() > -1e-5


def test_config_C5_one_patch_of_128_vs_oracle_with_exported_masks():
    """C5 against the float64 oracle where it is affordable: the 128^3 image at T = 2, overlap 1 (16 forwards); ONE of
    the 8 patches is restated on the CPU with the hash generator's exported masks of its two samples."""
    from tests.test_gpu_unet3d import _oracle_maps, make_model
    from values_amd import crop_indices, predict_image_sliding
    S, P, T, seed = 128, 64, 2, 99
    model = make_model(do_dropout=True)
    imgn = formula_volume((S, S, S), tag=53)
    out = predict_image_sliding([model], torch.from_numpy(imgn), patch_size=P, patch_overlap=1, n_pred=T, patch_batch=8,
                                seeds=[seed])
    crops = crop_indices((S, S, S), P, 1)
    b = 5
    c = crops[b]
    masks = [m[b * T:(b + 1) * T].cpu() for m in model.hash_dropout_masks(seed, len(crops) * T, P, P, P)]
    x = torch.from_numpy(imgn[c[0][0]:c[0][1], c[1][0]:c[1][1], c[2][0]:c[2][1]].copy())[None, None]
    _, ref = _oracle_maps(formula_sd_torch(), x, [[m[t:t + 1] for m in masks] for t in range(T)])
    sl = (slice(*c[0]), slice(*c[1]), slice(*c[2]))
    from tests.test_gpu_unet3d import MAP_TOL, REG_MAP_TOL, assert_close
    for k in KEYS:
        assert_close(np.abs(out[k][sl].cpu().numpy() - ref[k]).max(), MAP_TOL, REG_MAP_TOL, k)
