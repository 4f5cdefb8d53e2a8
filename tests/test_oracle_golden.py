"""CPU: pin the oracle (oracle/*.py) to the golden fixtures that tools/gen_golden.py
produced by running the imported reference.  If these fail the oracle is wrong and no
GPU parity claim means anything."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import aggregation_oracle as agg
from oracle import predict_oracle as pred
from oracle import uncertainty_oracle as unc
from oracle.unet3d_oracle import conv3d_k3_naive, unet3d_forward
from tests.helpers import GOLDEN, formula_sd_torch, load_npz, unpack_masks
from tests.formula import formula_tensor, formula_volume

KEYS = ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty")


def test_unc_hand_case_closed_form():
    g = load_npz("unc_kat.npz")
    r = unc.calculate_uncertainty(g["hand_in"])
    ln2 = np.log(2.0)
    np.testing.assert_allclose(r["pred_entropy"], [ln2, 0.5623351, ln2, 0.0], atol=1e-6)
    np.testing.assert_allclose(r["aleatoric_uncertainty"], [0.0, 0.4990473, ln2, 0.0], atol=1e-6)
    np.testing.assert_allclose(r["epistemic_uncertainty"], [ln2, 0.0632878, 0.0, 0.0], atol=1e-6)


@pytest.mark.parametrize("case", ["hand", "r3d", "r2d", "ex"])
def test_unc_matches_reference_outputs(case):
    g = load_npz("unc_kat.npz")
    r = unc.calculate_uncertainty(g[f"{case}_in"])
    for k in KEYS:
        assert r[k].dtype == np.float32 and g[f"{case}_{k}"].dtype == np.float32
        # f64 inputs: only the final f32 roundings can differ; f32 inputs (2D layout, entropies ~3):
        # numpy-vs-torch f32 log differs by 1-2 ulp (2.4e-7 each) and MI is a difference of two such sums
        atol = 1e-6 if case == "r2d" else 2e-7
        np.testing.assert_allclose(r[k], g[f"{case}_{k}"], atol=atol, rtol=0)
        assert not np.isnan(r[k]).any()


def test_unc_ssn_swaps_keys():
    g = load_npz("unc_kat.npz")
    r = unc.calculate_uncertainty(g["hand_in"], ssn=True)
    for k in KEYS:
        np.testing.assert_allclose(r[k], g[f"hand_ssn_{k}"], atol=2e-7)


def test_one_minus_msr():
    g = load_npz("unc_kat.npz")
    r = unc.calculate_one_minus_msr(g["msr_in"])
    np.testing.assert_array_equal(r["pred_entropy"], g["msr_pred_entropy"])


def test_conv_naive_vs_torch():
    x = formula_tensor((1, 3, 5, 6, 7), 1)
    w = formula_tensor((4, 3, 3, 3, 3), 2)
    b = formula_tensor((4,), 3)
    ref = torch.nn.functional.conv3d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
    np.testing.assert_allclose(conv3d_k3_naive(x, w, b), ref.numpy(), atol=1e-12)


@pytest.mark.parametrize("size", [16, 32])
def test_unet3d_oracle_matches_reference(size):
    g = load_npz(f"unet3d_{size}.npz")
    sd = formula_sd_torch()
    x = torch.from_numpy(formula_volume((1, 1, size, size, size)))
    np.testing.assert_array_equal(x.numpy().astype(np.float32), g["input"])
    T = g["logits"].shape[0]
    logits = []
    for t in range(T):
        masks = {k: torch.from_numpy(v) for k, v in unpack_masks(g, t).items()}
        with torch.no_grad():
            logits.append(unet3d_forward(sd, x, masks=masks)[0].numpy())
    logits = np.stack(logits)
    np.testing.assert_allclose(logits, g["logits"], atol=5e-7)  # fixture stores f32
    sm = unc.softmax(logits, axis=1)
    r = unc.calculate_uncertainty(sm)
    for k in KEYS:
        np.testing.assert_allclose(r[k], g[k], atol=1e-6)
    mean, mean_seg, pred_seg = unc.mean_and_argmax(sm)
    np.testing.assert_allclose(mean, g["mean_softmax"], atol=1e-6)
    np.testing.assert_array_equal(mean_seg, g["mean_seg"])
    np.testing.assert_array_equal(pred_seg, g["pred_seg"])
    with torch.no_grad():
        nd = unet3d_forward(sd, x, masks=None)[0].numpy()
    np.testing.assert_allclose(nd, g["logits_nodrop"], atol=5e-7)


def test_ensemble_tta_order():
    g = load_npz("ensemble_tta_16.npz")
    x = g["input"].astype(np.float64)
    xn = g["input_noise"].astype(np.float64)
    sds = [formula_sd_torch(seed_tag=s) for s in range(3)]

    def mk(sd):
        def f(xi, _):
            with torch.no_grad():
                return unet3d_forward(sd, torch.from_numpy(np.ascontiguousarray(xi)), masks=None).numpy()
        return f

    fw = [mk(sd) for sd in sds]
    preds = pred.predict_patch(fw, x, tta=True, x_noise=xn)
    assert preds.shape == g["softmax_pred"].shape == (48, 2, 16, 16, 16)
    np.testing.assert_allclose(preds, g["softmax_pred"], atol=5e-7)
    r = unc.calculate_uncertainty(preds)
    for k in KEYS:
        np.testing.assert_allclose(r[k], g[k], atol=1e-6)
    ens = pred.predict_patch(fw, x, n_pred=1)
    np.testing.assert_allclose(ens, g["ens_softmax_pred"], atol=5e-7)
    r = unc.calculate_uncertainty(ens)
    for k in KEYS:
        np.testing.assert_allclose(r[k], g["ens_" + k], atol=1e-6)


def test_patch_index_matches_reference():
    with open(os.path.join(GOLDEN, "patch_index.json")) as f:
        g = json.load(f)
    shapes = g.pop("_shapes")
    for key, crops in g.items():
        tag, name, patch, overlap = key.split("|")
        mine = pred.crop_indices(shapes[name], int(patch), float(overlap))
        assert [[list(c) for c in ci] for ci in mine] == crops, key


def test_accumulate_matches_reference_concat_data():
    g = load_npz("accum_24.npz")
    size, patch, T = 24, 16, 3
    crops = pred.crop_indices((size,) * 3, patch, 0.5)
    fake = np.abs(formula_tensor((len(crops), T, 2, patch, patch, patch), tag=55, scale=1.0))
    fake = fake / fake.sum(axis=2, keepdims=True)
    acc = pred.Accumulator(T, (size,) * 3)
    for pi, crop in enumerate(crops):
        for t in range(T):
            acc.add(crop, fake[pi, t], t)
    np.testing.assert_allclose(acc.softmax_pred, g["softmax_sum"], atol=1e-6)
    np.testing.assert_array_equal(acc.num_predictions, g["num_predictions"])
    np.testing.assert_allclose(acc.normalised(), g["normalised"], atol=1e-6)
    # D10: the reference applies calculate_uncertainty to the UN-normalised sum
    r = unc.calculate_uncertainty(acc.softmax_pred)
    for k in KEYS:
        np.testing.assert_allclose(r[k], g["unc_" + k], atol=2e-5, rtol=1e-6)


def test_aggregations_match_reference():
    with open(os.path.join(GOLDEN, "agg_kat.json")) as f:
        g = json.load(f)
    for size in (24, 64):
        tag = g[f"vol{size}_tag"]
        img = np.abs(formula_tensor((size,) * 3, tag=tag, scale=0.7)).astype(np.float32)
        c = size // 3
        img[c:c + 6, c + 2:c + 9, c + 1:c + 7] += 0.5
        r = g[f"vol{size}"]
        for key, kw in (("patch10", dict(patch_size=10)), ("patch10_mean", dict(patch_size=10, mean=True)),
                        ("patch_5_7_9", dict(patch_size=[5, 7, 9]))):
            mine = agg.patch_level_aggregation(img, **kw)
            # scipy picks fftconvolve here and transforms the float32 map in single precision,
            # so the reference value itself carries ~1e-7 relative FFT noise
            assert mine["max_score"] == pytest.approx(r[key]["max_score"], rel=1e-6)
            assert [list(b) for b in mine["bounding_box"]] == [list(b) for b in r[key]["bounding_box"]]
        assert agg.image_level_aggregation(img)["max_score"] == pytest.approx(r["image"]["max_score"], rel=1e-12)
        assert agg.image_level_aggregation(img, mean=True) == pytest.approx(r["image_mean"], rel=1e-12)
        for thr in (0.3, 0.6, 5.0):
            for mean in (True, False):
                mine = agg.threshold_aggregation(img, threshold=thr, mean=mean)
                assert float(mine["max_score"]) == pytest.approx(r[f"thr_{thr}_{int(mean)}"]["max_score"], rel=1e-6)
    img2 = np.abs(formula_tensor((40, 56), tag=33, scale=1.0)).astype(np.float32)
    mine = agg.patch_level_aggregation(img2, patch_size=10)
    assert mine["max_score"] == pytest.approx(g["img2d"]["patch10"]["max_score"], rel=1e-6)
    assert [list(b) for b in mine["bounding_box"]] == [list(b) for b in g["img2d"]["patch10"]["bounding_box"]]


def _hrnet_fixture():
    import json as _json
    from tests.formula import formula_state_dict_from_shapes
    g = load_npz("hrnet_small.npz")
    shapes = _json.loads(bytes(g["shapes_json"]).decode())
    sd = {k: torch.from_numpy(v).float() for k, v in formula_state_dict_from_shapes(shapes).items()}
    return g, shapes, sd


def hrnet_masks(g, t):
    out = []
    for i in range(4):
        shape = tuple(int(v) for v in g[f"maskshape_{i}"])
        n = int(np.prod(shape))
        out.append(np.unpackbits(g[f"mask_{t}_{i}"])[:n].astype(bool).reshape(shape))
    return out


def test_hrnet_oracle_matches_reference():
    """tests/golden/hrnet_small.npz: the reference HighResolutionNet (training-mode BN, DROPOUT_FINAL, float32 as
    test_2D.py runs it) with its own F.dropout masks captured; process_output's zero-channel + calculate_uncertainty."""
    from oracle.hrnet_oracle import hrnet_forward
    from tests.formula import HRNET_SMALL_EXTRA
    g, shapes, sd = _hrnet_fixture()
    x = torch.from_numpy(g["input"])
    np.testing.assert_array_equal(g["input"], formula_tensor((2, 3, 64, 96), tag=81, scale=1.5).astype(np.float32))
    T = g["logits"].shape[0]
    with torch.no_grad():
        for t in range(T):
            masks = [torch.from_numpy(m) for m in hrnet_masks(g, t)]
            y = hrnet_forward(HRNET_SMALL_EXTRA, sd, x, dropout_masks=masks).numpy()
            assert np.abs(y - g["logits"][t]).max() < 2e-5  # float32 both sides, same ATen kernels
        extra = dict(HRNET_SMALL_EXTRA, DROPOUT_FINAL=False)
        y = hrnet_forward(extra, sd, x).numpy()
        assert np.abs(y - g["logits_nodrop"]).max() < 2e-5
        # float64 oracle vs float32 reference: what float32 rounding costs on this net (bounds the GPU tolerance)
        sd64 = {k: v.double() for k, v in sd.items()}
        y64 = hrnet_forward(HRNET_SMALL_EXTRA, sd64, x.double(),
                            dropout_masks=[torch.from_numpy(m) for m in hrnet_masks(g, 0)]).numpy()
        assert np.abs(y64 - g["logits"][0]).max() < 1e-3
    sm = torch.softmax(torch.from_numpy(g["logits"]), dim=2)
    sm1 = torch.cat([sm, torch.zeros(T, 2, 1, 64, 96)], dim=2).numpy()
    for b in range(2):
        r = unc.calculate_uncertainty(sm1[:, b])
        for k in KEYS:
            np.testing.assert_allclose(r[k], g[f"{k}_{b}"], atol=1e-6)


def test_ssn_oracle_matches_reference():
    """SsnUNet3D.forward + distribution.sample (ssn_unet3D_module.py:39-70, test_3D.py:373-388) + ssn=True maps"""
    from oracle.ssn_oracle import lowrank_rsample, ssn_distribution
    from tests.formula import formula_ssn_state_dict
    g = load_npz("ssn_16.npz")
    NC, R = 2, 10
    sd = {k: torch.from_numpy(v) for k, v in formula_ssn_state_dict(NC, R).items()}
    x = torch.from_numpy(formula_volume((1, 1, 16, 16, 16)))
    np.testing.assert_array_equal(x.numpy().astype(np.float32), g["input"])
    with torch.no_grad():
        loc, diag, fac = ssn_distribution(sd, x, NC, R)
    np.testing.assert_allclose(loc.numpy(), g["loc"], atol=5e-7)
    np.testing.assert_allclose(diag.numpy(), g["cov_diag"], rtol=1e-6)
    np.testing.assert_allclose(fac.numpy(), g["cov_factor"], atol=5e-7)
    smp = lowrank_rsample(loc.numpy(), diag.numpy(), fac.numpy(), g["eps_w"], g["eps_d"])
    np.testing.assert_allclose(smp, g["samples"], atol=2e-6)
    S = smp.shape[0]
    sm = unc.softmax(smp.reshape(S, NC, 16, 16, 16), axis=1)
    r = unc.calculate_uncertainty(sm, ssn=True)
    for k in KEYS:
        np.testing.assert_allclose(r[k], g[k], atol=1e-6)


def test_metrics_oracle_loss_matches_reference_and_dice_by_definition():
    """SoftDiceLoss + NLLLoss (loss_modules.py:7-97, test_3D.py:262-273) pinned to the imported reference; the
    torchmetrics Dice restatement against the textbook definition on label masks (parity unpinned, see the oracle)."""
    from oracle import metrics_oracle as mo
    g = load_npz("metrics_kat.npz")
    for C in (2, 3):
        sm, gt = g[f"softmax_{C}"], g[f"gt_{C}"]
        per = [mo.soft_dice_loss(sm, gt[r][None]) + mo.nll_loss(np.log(sm), gt[r][None]) for r in range(gt.shape[0])]
        np.testing.assert_allclose(per, g[f"loss_per_rater_{C}"], rtol=1e-10)
        assert abs(mo.calculate_test_metrics(sm, gt)["loss"] - float(g[f"loss_{C}"])) < 1e-10
    a = np.array([[0, 1, 1, 0, 1, 0]]); b = np.array([[0, 1, 0, 0, 1, 1]])
    assert mo.tm_dice(a, b, ignore_index=0) == pytest.approx(2 * 2 / (3 + 3))          # foreground Dice
    assert mo.tm_dice(a, b) == pytest.approx(4 / 6)                                     # micro over both classes = accuracy
    assert mo.tm_dice(np.zeros((1, 5), int), np.zeros((1, 5), int), ignore_index=0) == 0.0   # empty foreground: 0/0 -> 0
    probs = np.stack([1 - a, a], 1).astype(np.float64) * 0.8 + 0.1                     # (1, 2, 6): arg-max == a
    assert mo.tm_dice(probs, b, ignore_index=0) == pytest.approx(4 / 6)
    # GED of identical predictions and raters is 0; of disjoint masks 2 * 1 - 0 - 0
    sm1 = np.repeat(probs, 3, 0)
    assert mo.calculate_ged(sm1, np.repeat(a, 2, 0))["ged"] == pytest.approx(0.0)
    anti = np.stack([a, 1 - a], 1).astype(np.float64) * 0.8 + 0.1
    assert mo.calculate_ged(np.repeat(anti, 2, 0), np.repeat(a, 2, 0))["ged"] == pytest.approx(2.0)
    # cross-check (not a pin: torchmetrics is absent) with scikit-learn's micro-averaged F1 over the kept labels, an
    # independent implementation of the same published quantity, on random label volumes of 2 / 3 / 5 classes
    for C in (2, 3, 5):
        p_, t_ = g[f"xc_pred_{C}"].astype(np.int64), g[f"xc_gt_{C}"].astype(np.int64)
        assert mo.tm_dice(p_, t_) == pytest.approx(float(g[f"xc_f1_all_{C}"]), abs=1e-15)
        assert mo.tm_dice(p_, t_, ignore_index=0) == pytest.approx(float(g[f"xc_f1_ign0_{C}"]), abs=1e-15)


def test_hrnet_ssn_oracle_matches_reference():
    """hrnet_ssn (hrnet_module.py:559-595) + distribution.sample (test_2D.py:285-299), normals captured"""
    import json as _json
    from oracle.hrnet_oracle import hrnet_forward
    from oracle.ssn_oracle import lowrank_rsample
    from tests.formula import HRNET_SMALL_EXTRA, formula_state_dict_from_shapes
    g = load_npz("hrnet_ssn.npz")
    shapes = _json.loads(bytes(g["shapes_json"]).decode())
    sd = {k: torch.from_numpy(v).float() for k, v in formula_state_dict_from_shapes(shapes).items()}
    x = torch.from_numpy(g["input"])
    extra = dict(HRNET_SMALL_EXTRA, DROPOUT_FINAL=False)
    with torch.no_grad():
        loc, diag, fac = hrnet_forward(extra, sd, x, ssn=(4, 10, 1e-5))
    assert np.abs(loc.numpy() - g["loc"]).max() < 2e-5
    np.testing.assert_allclose(diag.numpy(), g["cov_diag"], rtol=2e-5)
    assert np.abs(fac[:, ::997].numpy() - g["cov_factor_probe"]).max() < 2e-5
    S = g["samples"].shape[0]
    eps_d = formula_tensor(g["samples"].shape, tag=int(g["eps_d_tag"]), scale=1.7).astype(np.float32)
    smp = lowrank_rsample(loc.numpy(), diag.numpy(), fac.numpy(), g["eps_w"], eps_d)
    assert np.abs(smp - g["samples"]).max() < 1e-4


def test_hrnet_w18_widths_oracle_matches_reference():
    """HRNet-W18 widths (18/36/72/144 -> 270; BASELINE config 4): none is a multiple of 16"""
    import json as _json
    from oracle.hrnet_oracle import hrnet_forward
    from tests.formula import HRNET_W18S_EXTRA, formula_state_dict_from_shapes
    g = load_npz("hrnet_w18s.npz")
    shapes = _json.loads(bytes(g["shapes_json"]).decode())
    sd = {k: torch.from_numpy(v).float() for k, v in formula_state_dict_from_shapes(shapes).items()}
    x = torch.from_numpy(g["input"])
    with torch.no_grad():
        for t in range(2):
            masks = []
            for i in range(4):
                shape = tuple(int(v) for v in g[f"maskshape_{i}"])
                masks.append(torch.from_numpy(np.unpackbits(g[f"mask_{t}_{i}"])[:int(np.prod(shape))].astype(bool).reshape(shape)))
            y = hrnet_forward(HRNET_W18S_EXTRA, sd, x, dropout_masks=masks).numpy()
            assert np.abs(y - g["logits"][t]).max() < 2e-5
        y = hrnet_forward(dict(HRNET_W18S_EXTRA, DROPOUT_FINAL=False), sd, x).numpy()
        assert np.abs(y - g["logits_nodrop"]).max() < 2e-5


def test_hrnet_oracle_matches_reference_w18_full_layout_256x478():
    """the float64 oracle against the imported reference class in the FULL HRNet-W18 layout at 256 x 478 (config C4's
    network at the reference's image size): sub-grid and whole-map row / column sums of its float64 run"""
    import json
    from oracle.hrnet_oracle import hrnet_forward
    from tests.formula import formula_state_dict_from_shapes, formula_tensor
    from values_amd.hrnet_configs import hrnet_w18_extra
    g = dict(np.load(os.path.join(GOLDEN, "hrnet_w18_256x478.npz")))
    shapes = json.loads(bytes(g["shapes_json"]).decode())
    sd = {k: torch.from_numpy(v).double() for k, v in formula_state_dict_from_shapes(shapes).items()}
    x = torch.from_numpy(formula_tensor((1, 3, 256, 478), tag=int(g["input_tag"]), scale=float(g["input_scale"]))).float().double()
    with torch.no_grad():
        y = hrnet_forward(hrnet_w18_extra(False), sd, x).numpy()[0]
    assert np.abs(y[:, ::4, ::6] - g["logits64_sub"]).max() < 5e-7          # the sub-grid is stored as float32
    assert np.abs(y.sum(2) - g["logits64_rowsum"]).max() < 1e-9
    assert np.abs(y.sum(1) - g["logits64_colsum"]).max() < 1e-9
    assert np.abs(y[:, ::4, ::6] - g["logits_sub"]).max() < 2 * float(g["ref_f32_f64_gap"])


def test_metrics_oracle_hard_dice_and_ged_closed_forms():
    """f2 (parity unpinned for lack of torchmetrics 0.11.4): the restated micro Dice / GED against hand-counted label
    volumes -- perfect overlap, disjoint, half overlap, empty prediction, everything empty (0 / 0 -> 0), pooled pairs of
    two predictions and two raters, three classes with the ignored column deleted, and ignore_index = None (= accuracy)."""
    from oracle import metrics_oracle as mo
    from tests import dice_kat
    for c in dice_kat.cases():
        sm = dice_kat.onehot(np.stack(c["preds"]), c["C"])
        gt = np.stack(c["gts"])
        if c["dice"] is not None:
            assert mo.tm_dice(sm, gt, ignore_index=0) == pytest.approx(c["dice"], abs=1e-12), c["name"]
            assert mo.calculate_test_metrics(sm, gt)["dice"] == pytest.approx(c["dice"], abs=1e-12), c["name"]
        g = mo.calculate_ged(sm, gt, ignore_index=0)
        assert g["ged"] == pytest.approx(c["ged"], abs=1e-12), c["name"]
        if "max_dice_rater" in c:
            for r, v in enumerate(c["max_dice_rater"]):
                assert g["max dice rater {}".format(r)] == pytest.approx(v, abs=1e-7), c["name"]
            assert g["max dice pred"] == pytest.approx(c["max_dice_pred"], abs=1e-7), c["name"]
    p, g, acc = dice_kat.no_ignore_case()
    assert mo.tm_dice(p, g) == pytest.approx(acc, abs=1e-12)


def test_evalmetrics_oracle_matches_reference_fixture():
    """AURC / E-AURC, NCC, ACE (+ the Platt fit), AUROC: the numpy restatement against values produced by the
    reference's evaluation/metrics modules (tests/golden/evalmetrics_kat.npz, tools/gen_golden.py)"""
    from oracle import evalmetrics_oracle as em
    g = dict(np.load(os.path.join(GOLDEN, "evalmetrics_kat.npz")))
    cov, sel, wts = em.rc_curve_stats(g["aurc_risks"], g["aurc_confids"])
    np.testing.assert_allclose(cov, g["rc_coverages"], rtol=0, atol=0)
    np.testing.assert_allclose(sel, g["rc_risks"], rtol=1e-14)
    np.testing.assert_allclose(wts, g["rc_weights"], rtol=0, atol=0)
    assert abs(em.aurc(g["aurc_risks"], g["aurc_confids"]) - float(g["aurc"])) < 1e-14
    assert abs(em.eaurc(g["aurc_risks"], g["aurc_confids"]) - float(g["eaurc"])) < 1e-14
    assert abs(em.compute_ncc(g["ncc_gt"], g["ncc_pred"]) - float(g["ncc"])) < 1e-14
    assert abs(em.roc_auc(g["auroc_y"], g["auroc_score"]) - float(g["auroc"])) < 1e-14
    for tag, ign in (("all", None), ("ign2", 2)):
        F, y = em.rater_correct(g["ace_ref"], g["ace_pred"], g["ace_unc"], ign)
        a, b = em.sigmoid_calibration(F, y)
        # the fit: same optimum as the installed scikit-learn to ITS optimiser tolerance (gtol 1e-6, ftol 64 eps)
        assert abs(a - float(g[f"ace_{tag}_a"])) < 2e-3 * abs(a) and abs(b - float(g[f"ace_{tag}_b"])) < 2e-3 * abs(b), (a, b)
        # everything after the fit, with the reference's own (a, b): exact
        conf = em.platt_scale_confid(F, float(g[f"ace_{tag}_a"]), float(g[f"ace_{tag}_b"]))
        d, w, k = em.calib_stats(y, conf)
        np.testing.assert_allclose(d, g[f"ace_{tag}_disc"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(w, g[f"ace_{tag}_w"], rtol=1e-14)
        assert k == int(g[f"ace_{tag}_k"])
        assert abs(em.calc_ace(y, conf) - float(g[f"ace_{tag}"])) < 1e-14
    conf1 = 1 / (1 + np.exp(-g["ace_unc"].flatten() * 2.0 - 1.0))
    assert abs(em.calc_ace(np.ones(conf1.size, dtype=int), conf1) - float(g["ace_onelabel"])) < 1e-14


def test_evalmetrics_host_functions_match_reference_fixture():
    """the host halves of values_amd.evalmetrics (one scalar per image: AURC / E-AURC / AUROC) against the same fixture"""
    from values_amd import evalmetrics as vm
    g = dict(np.load(os.path.join(GOLDEN, "evalmetrics_kat.npz")))
    cov, sel, wts = vm.rc_curve_stats(g["aurc_risks"], g["aurc_confids"])
    np.testing.assert_allclose(cov, g["rc_coverages"], rtol=0, atol=0)
    np.testing.assert_allclose(sel, g["rc_risks"], rtol=1e-14)
    np.testing.assert_allclose(wts, g["rc_weights"], rtol=0, atol=0)
    assert abs(vm.aurc(g["aurc_risks"], g["aurc_confids"]) - float(g["aurc"])) < 1e-14
    assert abs(vm.eaurc(g["aurc_risks"], g["aurc_confids"]) - float(g["eaurc"])) < 1e-14
    assert abs(vm.roc_auc(g["auroc_y"], g["auroc_score"]) - float(g["auroc"])) < 1e-14
    assert vm.get_auroc_input({"3.nii.gz": {"a": {"max_score": 1.5}}, "25.nii.gz": {"a": {"max_score": 0.5}}}, "a") == ([1, 0], [1.5, 0.5])
