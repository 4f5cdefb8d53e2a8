"""GPU: the per-voxel reductions behind NCC and ACE (values_amd/csrc/evalmetrics.hip) against the reference's values
(tests/golden/evalmetrics_kat.npz, produced by importing evaluation/metrics/{ncc,ace}.py) and the numpy oracle."""
import json

import numpy as np
import pytest
import torch

from tests.helpers import load_npz

pytestmark = pytest.mark.gpu


def test_ncc_matches_reference_fixture_and_oracle():
    from oracle import evalmetrics_oracle as em
    from values_amd.evalmetrics import compute_ncc
    g = load_npz("evalmetrics_kat.npz")
    got = compute_ncc(g["ncc_gt"], g["ncc_pred"])
    # numpy evaluates mean / std / the product in the MAP's dtype: the fixture's predicted map is float32 (what the saved
    # NIfTI maps are), so the reference value carries float32 rounding (~1e-7 relative); the device sums in float64
    assert abs(got - float(g["ncc"])) < 2e-6, got
    assert abs(got - em.compute_ncc(g["ncc_gt"], g["ncc_pred"].astype(np.float64))) < 1e-13, got   # the float64 evaluation
    assert compute_ncc(g["ncc_gt"], g["ncc_pred"]) == got                     # deterministic
    rng = np.random.default_rng(3)
    a, b = rng.random((64, 64, 64)), rng.random((64, 64, 64)).astype(np.float32)          # a full-size map pair
    assert abs(compute_ncc(a, b) - em.compute_ncc(a, b.astype(np.float64))) < 1e-12
    assert abs(compute_ncc(a, a) - (a.size - 1) / a.size) < 1e-12             # std(ddof=1) under a 1/n product


@pytest.mark.parametrize("tag,ign", [("all", None), ("ign2", 2)])
def test_platt_fit_and_ace_match_reference_fixture(tag, ign):
    from oracle import evalmetrics_oracle as em
    from values_amd.evalmetrics import calc_ace, calib_stats, sigmoid_calibration
    g = load_npz("evalmetrics_kat.npz")
    ref, pred, unc = g["ace_ref"], g["ace_pred"], g["ace_unc"]
    a, b = sigmoid_calibration(ref, pred, unc, ignore_value=ign)
    # the same optimum as the installed scikit-learn, to ITS optimiser's tolerance; and as the oracle's Newton, tightly
    assert abs(a - float(g[f"ace_{tag}_a"])) < 2e-3 * abs(a) and abs(b - float(g[f"ace_{tag}_b"])) < 2e-3 * abs(b), (a, b)
    F, y = em.rater_correct(ref, pred, unc, ign)
    ao, bo = em.sigmoid_calibration(F, y)
    assert abs(a - ao) < 1e-8 * abs(ao) and abs(b - bo) < 1e-8 * abs(bo)
    # binning with the REFERENCE's parameters: discrepancies, weights, bin count, ACE
    ra, rb = float(g[f"ace_{tag}_a"]), float(g[f"ace_{tag}_b"])
    d, w, k = calib_stats(ref, pred, unc, ra, rb, ignore_value=ign)
    assert k == int(g[f"ace_{tag}_k"])
    np.testing.assert_allclose(d, g[f"ace_{tag}_disc"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(w, g[f"ace_{tag}_w"], rtol=1e-13)
    assert abs(calc_ace(ref, pred, unc, ra, rb, ignore_value=ign) - float(g[f"ace_{tag}"])) < 1e-13


def test_ace_single_label_quirk_and_2d_swap():
    from values_amd.evalmetrics import calc_ace
    g = load_npz("evalmetrics_kat.npz")
    unc = g["ace_unc"]
    pred = g["ace_pred"]
    allcorrect = np.repeat(pred[None], 2, 0)
    # P = 1 / (1 + exp(-unc * a + b)) with (a, b) = (2, -1) is the fixture's conf1
    got = calc_ace(allcorrect, pred, unc, 2.0, -1.0)
    # two raters = every voxel twice: the same bin fractions as the fixture's single copy.  The fixture's confidences were
    # formed in float32 (numpy keeps a float32 map float32 under Python-float parameters), the device forms them in float64
    assert abs(got - float(g["ace_onelabel"])) < 1e-6, got
    # a 2D map stored as (W, H) is swapped to the prediction's (H, W) (ace.py:24-26)
    u2, p2 = unc[:, :, 0], pred[:, :, 0]
    r2 = g["ace_ref"][:, :, :, 0]
    assert abs(calc_ace(r2, p2, np.swapaxes(u2, 0, 1).copy(), 3.0, 0.5) - calc_ace(r2, p2, u2, 3.0, 0.5)) < 1e-15


def test_evaluation_drivers_on_a_results_directory(tmp_path):
    """failure_detection / ambiguity_modeling / calibration over a small results tree written by values_amd.results"""
    from values_amd import evalmetrics as vm, nifti
    from values_amd.experiment import ExperimentDataloader, ExperimentVersion
    rng = np.random.default_rng(5)
    ev = ExperimentVersion(base_path=tmp_path, naming_scheme_version="fold{fold}", pred_model="Dropout", image_ending=".nii.gz",
                           unc_ending=".nii.gz", unc_types=["pred_entropy"], aggregations=["image_level"], n_reference_segs=2,
                           fold=0)
    metrics, agg = {}, {}
    for split in ("val", "test"):
        d = ev.exp_path / split
        for sub in ("pred_seg", "pred_entropy", "gt_seg"):
            (d / sub).mkdir(parents=True, exist_ok=True)
        for i in range(3):
            iid = f"{30 + i}"
            segs = (rng.random((2, 8, 8, 8)) < 0.4).astype(np.uint8)
            pred = segs[0].copy()
            pred[rng.random(pred.shape) < 0.2] ^= 1
            unc = (0.5 * (segs[0] != segs[1]) + 0.2 * rng.random(pred.shape)).astype(np.float32)
            nifti.save(pred, d / "pred_seg" / f"{iid}_mean.nii.gz")
            nifti.save(unc, d / "pred_entropy" / f"{iid}.nii.gz")
            for r in range(2):
                nifti.save(segs[r], d / "gt_seg" / f"{iid}_{r:02d}.nii.gz")
            metrics[iid] = {"dice": float(rng.random())}
            agg[f"{iid}.nii.gz"] = {"image_level": {"max_score": float(unc.sum())}}
        json.dump(metrics, open(d / "metrics.json", "w"))
        json.dump(agg, open(d / "aggregated_pred_entropy.json", "w"))
    dl = ExperimentDataloader(ev, "test")
    fd = vm.failure_detection(dl)
    assert 0 <= fd["mean"]["pred_entropy"]["image_level"]["metrics"]["aurc"] <= 1
    am = vm.ambiguity_modeling(dl)
    assert am["mean"]["pred_entropy"]["metrics"]["ncc"] > 0.5
    cal = vm.calibration(dl)
    assert (ev.exp_path / "platt_scale_params.json").exists()
    assert 0 <= cal["mean"]["pred_entropy"]["metrics"]["ace"] <= 1
    assert set(json.load(open(dl.dataset_path / "calibration.json"))) == {"mean", "30", "31", "32"}
    assert "auroc" in vm.ood_auroc(dl)["mean"]["pred_entropy"]["image_level"]["metrics"]
