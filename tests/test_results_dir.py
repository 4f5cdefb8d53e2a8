"""CPU: NIfTI codec, results-directory writer and the ExperimentVersion / ExperimentDataloader path scheme
(no arithmetic on the GPU here; the GPU-backed methods are exercised in tests/test_gpu_results.py)."""
import gzip
import json
import os
import struct

import numpy as np
import pytest

from values_amd import nifti
from values_amd.experiment import ExperimentDataloader, ExperimentVersion
from values_amd.results import log_metrics, results_dir, save_case


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64, np.int16])
def test_nifti_round_trip_and_layout(tmp_path, dtype):
    a = (np.arange(3 * 4 * 5).reshape(3, 4, 5) % 251).astype(dtype)
    p = tmp_path / "a.nii.gz"
    nifti.save(a, str(p))
    b, hdr = nifti.load(str(p))
    assert b.dtype == a.dtype and np.array_equal(a, b)
    raw = gzip.open(p, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack_from("<8h", raw, 40)[:4] == (3, 3, 4, 5)         # dim[1..3] = (X, Y, Z)
    first = np.frombuffer(raw, dtype=np.dtype(dtype).newbyteorder("<"), count=3, offset=352)
    assert np.array_equal(first, a[:3, 0, 0])                             # x varies fastest on disk
    p2 = tmp_path / "b.nii"
    nifti.save(a, str(p2), {"pixdim": [2.0, 3.0, 4.0]})
    c, hdr2 = nifti.load(str(p2))
    assert np.array_equal(a, c) and hdr2["pixdim"] == [2.0, 3.0, 4.0]


def test_nifti_edge_cases(tmp_path):
    nifti.save(np.zeros((2, 2), dtype=bool), str(tmp_path / "m.nii.gz"))
    assert nifti.load(str(tmp_path / "m.nii.gz"))[0].dtype == np.uint8
    with open(tmp_path / "junk.nii", "wb") as f:
        f.write(b"\0" * 400)
    with pytest.raises(ValueError):
        nifti.load(str(tmp_path / "junk.nii"))


def _write_fake_results(root, pred_model="Dropout", n_images=2, T=3):
    rng = np.random.default_rng(0)
    d = results_dir(str(root), pred_model, "fold0_seed123", "id")
    metrics = {}
    for i in range(n_images):
        p = rng.random((T, 2, 6, 5, 4))
        p /= p.sum(1, keepdims=True)
        maps = {k: rng.random((6, 5, 4)).astype(np.float32) for k in
                ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty")}
        gt = rng.integers(0, 2, (2, 6, 5, 4)).astype(np.int32)
        save_case(d, f"img{i}", p, maps, data=rng.random((6, 5, 4)), gt_seg=gt)
        metrics[f"img{i}"] = {"dice": float(i), "loss": 0.5}
    log_metrics(d, metrics)
    return d


def test_results_tree_matches_reference_names(tmp_path):
    d = _write_fake_results(tmp_path)
    assert sorted(os.listdir(d)) == ["aleatoric_uncertainty", "epistemic_uncertainty", "gt_seg", "input", "metrics.json",
                                     "pred_entropy", "pred_prob", "pred_seg"]
    assert sorted(os.listdir(os.path.join(d, "pred_seg")))[:4] == ["img0_01.nii.gz", "img0_02.nii.gz", "img0_03.nii.gz",
                                                                   "img0_mean.nii.gz"]
    assert "img0_mean_02.nii.gz" in os.listdir(os.path.join(d, "pred_prob"))
    assert "img1_03_01.nii.gz" in os.listdir(os.path.join(d, "pred_prob"))
    assert sorted(os.listdir(os.path.join(d, "gt_seg")))[:2] == ["img0_00.nii.gz", "img0_01.nii.gz"]
    m = json.load(open(os.path.join(d, "metrics.json")))
    assert m["mean"] == {"dice": 0.5, "loss": 0.5}
    seg, _ = nifti.load(os.path.join(d, "pred_seg", "img0_mean.nii.gz"))
    assert seg.dtype == np.uint8 and seg.shape == (6, 5, 4)


def test_experiment_version_and_dataloader_paths(tmp_path):
    _write_fake_results(tmp_path)
    ev = ExperimentVersion(base_path=tmp_path, naming_scheme_version="fold{fold}_seed{seed}", pred_model="Dropout",
                           image_ending=".nii.gz", unc_ending=".nii.gz",
                           unc_types=["predictive_uncertainty", "aleatoric_uncertainty", "epistemic_uncertainty"],
                           aggregations=["patch_level"], n_reference_segs=2, fold=0, seed=123)
    assert ev.version_name == "fold0_seed123" and ev.version_params == {"fold": 0, "seed": 123}
    assert ev.exp_path == tmp_path / "Dropout" / "test_results" / "fold0_seed123"
    dl = ExperimentDataloader(ev, "id")
    assert dl.image_ids == ["img0", "img1"]
    assert dl.unc_path_dict["predictive_uncertainty"] == dl.dataset_path / "pred_entropy"
    assert dl.unc_path_dict["epistemic_uncertainty"] == dl.dataset_path / "epistemic_uncertainty"
    assert dl.get_unc_map("img1", "aleatoric_uncertainty").shape == (6, 5, 4)
    assert len(dl.get_pred_segs("img0")) == 4
    assert dl.get_mean_pred_seg("img0").dtype == np.uint8
    refs = dl.get_reference_segs("img0")
    assert refs.shape == (2, 6, 5, 4)
    np.testing.assert_array_equal(dl.get_gt_unc_map("img0"), np.var(refs, axis=0))
    assert dl.get_aggregated_unc_files_dict() == {}


def test_png_and_tiff_codecs_round_trip_and_decode_with_an_independent_reader(tmp_path):
    """2D result files (test_2D.py:145-158): our writers vs our readers, and vs PIL where it is installed"""
    from tests.formula import formula_tensor
    from values_amd.image_io import read_png, read_tiff_f32, write_png, write_tiff_f32
    rgb = ((formula_tensor((37, 53, 3), 91) + 1) * 127.5).astype(np.uint8)
    grey = ((formula_tensor((5, 7), 92) + 1) * 127.5).astype(np.uint8)
    f32 = formula_tensor((31, 17), 93, scale=3.0).astype(np.float32)
    f32[0, 0], f32[1, 1] = 0.0, np.float32(1e-30)
    write_png(tmp_path / "a.png", rgb); write_png(tmp_path / "g.png", grey); write_tiff_f32(tmp_path / "f.tif", f32)
    np.testing.assert_array_equal(read_png(tmp_path / "a.png"), rgb)
    np.testing.assert_array_equal(read_png(tmp_path / "g.png"), grey)
    np.testing.assert_array_equal(read_tiff_f32(tmp_path / "f.tif"), f32)
    PIL = pytest.importorskip("PIL.Image")
    np.testing.assert_array_equal(np.array(PIL.open(tmp_path / "a.png")), rgb)
    np.testing.assert_array_equal(np.array(PIL.open(tmp_path / "g.png")), grey)
    np.testing.assert_array_equal(np.array(PIL.open(tmp_path / "f.tif")), f32)
    # a PNG written by another encoder (filters 1-4) decodes too
    PIL.fromarray(rgb).save(tmp_path / "p.png", optimize=True)
    np.testing.assert_array_equal(read_png(tmp_path / "p.png"), rgb)


def test_trainid_palette_matches_reference_table():
    """cityscapes_labels.py:59-126 (trainId2color is built over reversed(labels): the FIRST label of a train id wins)"""
    from values_amd.results2d import TRAINID2COLOR, UNLABELED
    assert len(TRAINID2COLOR) == 25 and TRAINID2COLOR[UNLABELED] == (0, 0, 0)
    assert TRAINID2COLOR[0] == (128, 64, 128) and TRAINID2COLOR[13] == (0, 0, 142) and TRAINID2COLOR[23] == (84, 86, 22)
