"""GPU: inference -> results directory -> ExperimentDataloader -> aggregate_uncertainties, end to end."""
import json
import os

import numpy as np
import pytest
import torch

from tests.formula import formula_volume

pytestmark = pytest.mark.gpu


def test_end_to_end_results_and_aggregation(tmp_path):
    from oracle import aggregation_oracle as ao
    from tests.test_gpu_unet3d import make_model
    from values_amd import nifti, predict_image_sliding
    from values_amd.experiment import ExperimentDataloader, ExperimentVersion, aggregate_uncertainties
    from values_amd.results import results_dir, save_case
    model = make_model(do_dropout=True)
    d = results_dir(str(tmp_path), "Dropout", "fold0_seed123", "id")
    keep = {}
    for i in range(2):
        img = torch.from_numpy(formula_volume((32, 32, 32), tag=60 + i))
        out = predict_image_sliding([model], img, patch_size=32, n_pred=4, seeds=[i])
        save_case(d, f"case{i}", out["softmax_sum"], out, data=img, num_predictions=out["num_predictions"])
        keep[f"case{i}"] = out
    ev = ExperimentVersion(base_path=tmp_path, naming_scheme_version="fold{fold}_seed{seed}", pred_model="Dropout",
                           image_ending=".nii.gz", unc_ending=".nii.gz",
                           unc_types=["predictive_uncertainty", "epistemic_uncertainty"], aggregations=None,
                           n_reference_segs=0, fold=0, seed=123)
    dl = ExperimentDataloader(ev, "id")
    assert dl.image_ids == ["case0", "case1"]
    m = dl.get_unc_map("case1", "epistemic_uncertainty")
    np.testing.assert_array_equal(m, keep["case1"]["epistemic_uncertainty"].cpu().numpy())
    np.testing.assert_array_equal(dl.get_mean_pred_seg("case0"), keep["case0"]["pred_seg_mean"].cpu().numpy())
    ref = "evaluation.uncertainty_aggregation.aggregate_uncertainties."
    aggs = {"patch_level": {"_target_": ref + "patch_level_aggregation", "patch_size": 10},
            "image_level": {"_target_": ref + "image_level_aggregation"},
            "threshold": {"_target_": ref + "threshold_aggregation", "threshold": 0.01}}
    aggregate_uncertainties(dl, aggs)
    for unc, key in (("predictive_uncertainty", "pred_entropy"), ("epistemic_uncertainty", "epistemic_uncertainty")):
        j = json.load(open(dl.dataset_path / f"aggregated_{unc}.json"))
        for cid in ("case0", "case1"):
            img = keep[cid][key].cpu().numpy()
            r = j[f"{cid}.nii.gz"]
            o = ao.patch_level_aggregation(img, patch_size=10)
            assert r["patch_level"]["max_score"] == pytest.approx(o["max_score"], rel=1e-9)
            assert [list(b) for b in r["patch_level"]["bounding_box"]] == [list(b) for b in o["bounding_box"]]
            assert r["image_level"]["max_score"] == pytest.approx(ao.image_level_aggregation(img)["max_score"], rel=1e-6)  # reference sums in f32
            assert r["threshold"]["max_score"] == pytest.approx(float(ao.threshold_aggregation(img, 0.01)["max_score"]), rel=1e-6)
    assert set(dl.get_aggregated_unc_files_dict()) == {"predictive_uncertainty", "epistemic_uncertainty"}


def test_softmax_model_gets_one_minus_msr_maps(tmp_path):
    """ExperimentDataloader._setup_pred_entropy_softmax (experiment_dataloader.py:51-61): a plain Softmax model has
    no uncertainty maps; pred_entropy := 1 - max softmax is created from pred_prob on first use."""
    from tests.test_gpu_unet3d import make_model
    from values_amd import nifti, predict_uncertainty
    from values_amd.experiment import ExperimentDataloader, ExperimentVersion
    from values_amd.results import results_dir, save_case
    model = make_model(do_dropout=False)
    x = torch.from_numpy(formula_volume((1, 1, 16, 16, 16), tag=70)).float().cuda()
    out = predict_uncertainty([model], x, n_pred=1)
    sm = torch.softmax(out["logits"][0], 1)  # (1, C, ...)
    d = results_dir(str(tmp_path), "Softmax", "fold0_seed123", "id")
    save_case(d, "c0", sm)
    assert not os.path.exists(os.path.join(d, "pred_entropy"))
    ev = ExperimentVersion(base_path=tmp_path, naming_scheme_version="fold{fold}_seed{seed}", pred_model="Softmax",
                           image_ending=".nii.gz", unc_ending=".nii.gz", unc_types=["predictive_uncertainty"],
                           aggregations=None, n_reference_segs=0, fold=0, seed=123)
    dl = ExperimentDataloader(ev, "id")
    got = dl.get_unc_map("c0", "predictive_uncertainty")
    ref = 1 - sm[0].double().cpu().numpy().max(0)
    np.testing.assert_allclose(got, ref, atol=1e-12)
    assert dl.get_mean_pred_seg("c0").shape == (16, 16, 16)  # Softmax models use the _01 segmentation


@pytest.mark.parametrize("C,T,R,shape", [(2, 5, 4, (12, 10, 8)), (3, 4, 3, (9, 7, 5)), (2, 1, 1, (4, 4, 4))])
def test_metrics_match_oracle(C, T, R, shape):
    """calculate_test_metrics / calculate_ged (test_3D.py:250-358) from the two device reductions vs the oracle"""
    from oracle import metrics_oracle as mo
    from tests.formula import formula_tensor
    from values_amd.metrics import calculate_ged, calculate_test_metrics, mask_agreement
    logits = formula_tensor((T, C) + shape, 8100 + C, scale=2.5)
    e = np.exp(logits - logits.max(1, keepdims=True))
    sm = (e / e.sum(1, keepdims=True)).astype(np.float32)
    gt = ((formula_tensor((R,) + shape, 8200 + C) + 1.0) * 0.5 * C).astype(np.int64).clip(0, C - 1)
    mean = sm.mean(0, keepdims=True)
    got = calculate_test_metrics(torch.from_numpy(mean).cuda(), torch.from_numpy(gt).cuda())
    ref = mo.calculate_test_metrics(mean.astype(np.float64), gt)
    assert abs(got["loss"] - ref["loss"]) < 1e-5 and abs(got["dice"] - ref["dice"]) < 1e-7
    got = calculate_ged(torch.from_numpy(sm).cuda(), torch.from_numpy(gt).cuda())
    ref = mo.calculate_ged(sm, gt)
    assert set(got) == set(ref)
    for k in ref:
        assert abs(got[k] - ref[k]) < 1e-6, k
    # the counts themselves, against a direct count
    masks = np.concatenate([sm.argmax(1), gt], 0)
    I = mask_agreement(torch.from_numpy(masks).cuda(), C)
    for i in range(len(masks)):
        for j in range(len(masks)):
            for c in range(C):
                assert I[i, j, c] == int(((masks[i] == c) & (masks[j] == c)).sum())


def test_hard_dice_and_ged_closed_forms_on_device():
    """f2: values_amd.metrics (vx_mask_agreement counts -> ratios) against the hand-counted answers of tests/dice_kat.py --
    the same table the oracle is held to on the CPU.  The row stays parity-unpinned (torchmetrics is absent); these pin the
    documented definition, including its 0 / 0 -> 0 quirk."""
    from tests import dice_kat
    from values_amd.metrics import _micro_dice, calculate_ged, calculate_test_metrics, mask_agreement
    for c in dice_kat.cases():
        sm = torch.from_numpy(dice_kat.onehot(np.stack(c["preds"]), c["C"])).cuda()
        gt = torch.from_numpy(np.stack(c["gts"])).cuda()
        if c["dice"] is not None:
            # calculate_test_metrics takes the MEAN prediction (1, C, ...): one prediction per case here
            assert calculate_test_metrics(sm[:1], gt)["dice"] == pytest.approx(c["dice"], abs=1e-12), c["name"]
        g = calculate_ged(sm, gt, ignore_index=0)
        assert g["ged"] == pytest.approx(c["ged"], abs=1e-12), c["name"]
        if "max_dice_rater" in c:
            for r, v in enumerate(c["max_dice_rater"]):
                assert g["max dice rater {}".format(r)] == pytest.approx(v, abs=1e-7), c["name"]
            assert g["max dice pred"] == pytest.approx(c["max_dice_pred"], abs=1e-7), c["name"]
    p, gl, acc = dice_kat.no_ignore_case()
    I = mask_agreement(torch.from_numpy(np.concatenate([p, gl], 0)).cuda(), 3)
    assert _micro_dice(I, [0], [1], [0, 1, 2]) == pytest.approx(acc, abs=1e-12)


def test_metrics_edge_cases():
    from oracle import metrics_oracle as mo
    from values_amd.metrics import calculate_ged
    shape = (6, 6, 6)
    sm = np.zeros((3, 2) + shape, np.float32); sm[:, 0] = 0.9; sm[:, 1] = 0.1       # predictions: all background
    gt = np.zeros((2,) + shape, np.int64)                                           # raters: all background
    got, ref = calculate_ged(torch.from_numpy(sm).cuda(), torch.from_numpy(gt).cuda()), mo.calculate_ged(sm, gt)
    assert got == pytest.approx(ref) and got["ged"] == pytest.approx(0.0)           # 2*1 - 1 - 1 (all 0/0 -> Dice 0)
    gt[0, :3] = 1
    got, ref = calculate_ged(torch.from_numpy(sm).cuda(), torch.from_numpy(gt).cuda()), mo.calculate_ged(sm, gt)
    assert got == pytest.approx(ref)
    got = calculate_ged(torch.from_numpy(sm).cuda(), torch.from_numpy(gt).cuda(), ged_only=True)
    assert list(got) == ["ged"]


@pytest.mark.parametrize("n,qs", [(1, [0.0, 0.5, 1.0]), (2, [0.25, 0.5, 0.75]), (1000, [0.0, 0.013, 0.5, 0.987, 1.0]),
                                  (3 * 64 ** 3, [0.9731, 0.5, 0.99999])])
def test_quantile_is_bit_identical_to_numpy(n, qs):
    """np.quantile of find_threshold.py:61-66 via radix select: exact, including ties, negatives and interpolation"""
    from tests.formula import hash_uniform
    from values_amd.thresholds import quantile
    x = (hash_uniform(n, 31) * 0.7).astype(np.float32)
    x[::7] = 0.0                      # many exact ties at 0 (uncertainty maps are mostly 0)
    x[1::11] = np.abs(x[1::11])
    xd = torch.from_numpy(x).cuda()
    for q in qs:
        # float64 index and interpolation on the float32 values = numpy 1.24.3 (the reference's pin) for a python-float
        # q; numpy >= 2 rounds q and the interpolation to float32 for float32 data, so compare on the float64 view
        assert quantile(xd, q) == float(np.quantile(x.astype(np.float64), q)), (n, q)
    assert quantile(torch.zeros(50).cuda(), 0.3) == 0.0


def test_find_threshold_files(tmp_path):
    """quantile_analysis.json / threshold_analysis.json as find_threshold.py writes them, consumed by
    threshold_aggregation (aggregate_uncertainties.py:59-60)"""
    from values_amd import nifti
    from values_amd.aggregation import threshold_aggregation
    from tests.formula import formula_tensor
    from values_amd.thresholds import (calculate_foreground_quantile_image, find_threshold, save_foreground_quantiles)
    segs = [(formula_tensor((8, 8, 8), 40 + i) > 0.6).astype(np.uint8) for i in range(3)]
    qs = [calculate_foreground_quantile_image(s) for s in segs]
    assert qs == [1 - np.count_nonzero(s) / s.size for s in segs]
    methods = save_foreground_quantiles({"Dropout": {"v1": qs[:2], "v2": qs[2:]}, "Softmax": {"v1": qs[:1]}}, tmp_path)
    assert methods["Dropout"] == float(np.mean(qs))
    paths = {"Dropout": {"v1": {}}, "Softmax": {"v1": {}}}
    maps = {}
    for unc in ("predictive_uncertainty", "aleatoric_uncertainty", "epistemic_uncertainty"):   # evaluation/configs/datasets/*.yaml
        ps = []
        for i in range(2):
            m = np.abs(formula_tensor((8, 8, 8), 50 + i + len(unc))).astype(np.float32)
            p = tmp_path / f"{unc}_{i}.nii.gz"
            nifti.save(m, p)
            ps.append(p)
            maps.setdefault(unc, []).append(m)
        paths["Dropout"]["v1"][unc] = ps
    paths["Softmax"]["v1"]["predictive_uncertainty"] = paths["Dropout"]["v1"]["predictive_uncertainty"]
    td = find_threshold(paths, tmp_path, tmp_path)
    want = float(np.quantile(np.array(maps["aleatoric_uncertainty"]).astype(np.float64), methods["Dropout"]))
    assert td["Dropout"]["Mean aleatoric threshold"] == want
    on_disk = json.load(open(tmp_path / "threshold_analysis.json"))
    assert on_disk["Mean"]["Mean predictive threshold"] == np.mean([td["Dropout"]["Mean predictive threshold"],
                                                                     td["Softmax"]["Mean predictive threshold"]])
    r = threshold_aggregation(image=maps["predictive_uncertainty"][0], pred_model="Dropout", unc_type="predictive_uncertainty",
                              threshold_path=str(tmp_path / "threshold_analysis.json"))
    assert r["threshold"] == on_disk["Dropout"]["Mean predictive threshold"]
    m = maps["predictive_uncertainty"][0].astype(np.float64)
    assert abs(r["max_score"] - m[m >= r["threshold"]].mean()) < 1e-9


def test_2d_results_directory(tmp_path):
    """save_prediction / save_uncertainty (test_2D.py:116-159): names, colours, ignore map, float32 TIFF content"""
    from tests.formula import formula_tensor
    from values_amd.image_io import read_png, read_tiff_f32
    from values_amd.results2d import TRAINID2COLOR, create_save_dirs, save_prediction, save_uncertainty
    H, W, T = 12, 20, 3
    masks = ((formula_tensor((T, H, W), 71) + 1) * 12).astype(np.uint8).clip(0, 23)
    mean = ((formula_tensor((H, W), 72) + 1) * 12).astype(np.uint8).clip(0, 23)
    ign = formula_tensor((H, W), 73) > 0.8
    d = create_save_dirs(str(tmp_path), "exp", 0, "val")
    assert d["save_pred_dir"].endswith(os.path.join("exp", "test_results", "0", "val", "pred_seg"))
    save_prediction(d["save_pred_dir"], "img7", torch.from_numpy(masks).cuda(), torch.from_numpy(mean).cuda(), ign)
    assert sorted(os.listdir(d["save_pred_dir"])) == ["img7_01.png", "img7_02.png", "img7_03.png", "img7_mean.png"]
    want = np.zeros((H, W, 3), np.uint8)
    lab = mean.copy(); lab[ign] = 255
    for k, v in TRAINID2COLOR.items():
        want[lab == k] = v
    np.testing.assert_array_equal(read_png(os.path.join(d["save_pred_dir"], "img7_mean.png")), want)
    save_prediction(d["save_pred_dir"], "solo", torch.from_numpy(masks[:1]).cuda(), None, None)
    assert os.path.exists(os.path.join(d["save_pred_dir"], "solo_01.png"))
    unc = {"pred_entropy": torch.from_numpy(np.abs(formula_tensor((H, W), 74)).astype(np.float32)).cuda()}
    save_uncertainty(d["save_dir"], "img7", unc)
    np.testing.assert_array_equal(read_tiff_f32(os.path.join(d["save_dir"], "pred_entropy", "img7.tif")),
                                  unc["pred_entropy"].cpu().numpy())
