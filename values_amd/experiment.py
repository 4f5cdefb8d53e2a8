"""The results-directory contract of the evaluation stage, MedPy/hydra-free.

`ExperimentVersion` and `ExperimentDataloader` mirror evaluation/experiment_version.py:4-51 and
evaluation/experiment_dataloader.py:11-169: same constructor arguments, attributes, method names and path scheme
    <base_path>/<naming_scheme_pred_model>/test_results/<version_name>/<split>/{pred_seg,pred_prob,pred_entropy,
    aleatoric_uncertainty,epistemic_uncertainty,gt_seg}/<id>...<ending>
so evaluation code written against the reference classes runs unchanged.  Volumes are read with
`values_amd.nifti`; the two arithmetic methods (1 - max softmax, aggregation) run on the GPU.
`aggregate_uncertainties` mirrors evaluation/uncertainty_aggregation/aggregate_uncertainties.py:70-95.
"""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np

from . import nifti
from .io import instantiate


class ExperimentVersion:
    def __init__(self, base_path, naming_scheme_version, pred_model, image_ending, unc_ending, unc_types, aggregations,
                 n_reference_segs, second_cycle_path=None, n_classes=2, naming_scheme_pred_model="{pred_model}",
                 datamodule_config=None, pred_seg_loading=None, gt_unc_map_loading=None, **kwargs):
        self.pred_model = pred_model
        self.naming_scheme_pred_model = naming_scheme_pred_model
        self.naming_scheme_version = naming_scheme_version
        self.version_params = kwargs
        self.version_name = self._build_version_name(naming_scheme_version=naming_scheme_version, **kwargs)
        self.base_path = Path(base_path)
        self.exp_path = (self.base_path / naming_scheme_pred_model.format(pred_model=pred_model, **kwargs)
                         / "test_results" / self.version_name)
        self.second_cycle_path = Path(second_cycle_path) if second_cycle_path is not None else None
        self.image_ending, self.unc_ending = image_ending, unc_ending
        self.n_reference_segs, self.n_classes = n_reference_segs, n_classes
        self.unc_types, self.aggregations = unc_types, aggregations
        self.datamodule_config = datamodule_config
        self.pred_seg_loading, self.gt_unc_map_loading = pred_seg_loading, gt_unc_map_loading

    def _build_version_name(self, naming_scheme_version: str, **kwargs):
        return naming_scheme_version.format(**kwargs)


def set_seed(seed: int) -> None:
    """evaluation/utils/set_seed.py:9-18 without the Lightning call: python / numpy / torch generators (the downstream
    tasks that sample -- Platt-scaling splits, threshold searches -- start from the experiment's seed)"""
    import random
    import numpy as np
    import torch
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


class ExperimentDataloader:
    def __init__(self, exp_version: ExperimentVersion, dataset_split):
        self.exp_version = exp_version
        if "seed" in getattr(exp_version, "version_params", {}):      # experiment_dataloader.py:14
            set_seed(int(exp_version.version_params["seed"]))
        self.dataset_split = dataset_split
        self.dataset_path = exp_version.exp_path / dataset_split if dataset_split else exp_version.exp_path
        self.pred_seg_dir = self.dataset_path / "pred_seg"
        prob = self.dataset_path / "pred_prob"
        self.pred_prob_dir = prob if os.path.exists(prob) else None
        self.image_ids = sorted(self._get_image_ids())
        if exp_version.pred_model == "Softmax":
            self._setup_pred_entropy_softmax()
        self.unc_path_dict = self._setup_unc_path_dict()
        if exp_version.datamodule_config is not None:
            self.dataloader = self.setup_dataloader()
            self.ref_seg_dir = None
        else:
            self.dataloader = None
            self.ref_seg_dir = self.dataset_path / "gt_seg"

    # -- 1 - max softmax for plain Softmax models (experiment_dataloader.py:38-61), on the GPU
    def get_max_softmax_pred(self, image_id: str):
        import torch
        from .uncertainty import calculate_one_minus_msr
        probs = []
        for c in range(self.exp_version.n_classes):
            f = os.path.join(self.pred_prob_dir, f"{image_id}_01_{str(c + 1).zfill(2)}{self.exp_version.unc_ending}")
            probs.append(nifti.load(f)[0])
        return calculate_one_minus_msr(torch.from_numpy(np.array(probs)))["pred_entropy"].numpy()

    def _setup_pred_entropy_softmax(self):
        target = self.dataset_path / "pred_entropy"
        if not os.path.exists(target):
            os.makedirs(target)
            for image_id in self.image_ids:
                nifti.save(self.get_max_softmax_pred(image_id), target / f"{image_id}{self.exp_version.unc_ending}")

    def _setup_unc_path_dict(self):
        return {u: self.dataset_path / ("pred_entropy" if u == "predictive_uncertainty" else u)
                for u in self.exp_version.unc_types}

    def _get_image_ids(self):
        end = self.exp_version.image_ending
        return set("_".join(n.split("_")[:-1]) for n in os.listdir(self.pred_seg_dir) if n.endswith(end))

    def get_pred_seg_paths(self, image_id):
        end = self.exp_version.image_ending
        return [self.pred_seg_dir / n for n in os.listdir(self.pred_seg_dir) if n.startswith(image_id) and n.endswith(end)]

    def get_pred_segs(self, image_id):
        return [nifti.load(p)[0] for p in self.get_pred_seg_paths(image_id)]

    def get_aggregated_unc_files_dict(self):
        return {u: self.dataset_path / f"aggregated_{u}.json" for u in self.unc_path_dict
                if os.path.isfile(self.dataset_path / f"aggregated_{u}.json")}

    def setup_dataloader(self):
        dm = instantiate(dict(self.exp_version.datamodule_config), test_split=self.dataset_split)
        dm.setup("test")
        return dm.test_dataloader()

    def _reference_segs(self, image_id):
        end = self.exp_version.image_ending
        return np.array([nifti.load(self.ref_seg_dir / f"{image_id}_{i:02d}{end}")[0]
                         for i in range(self.exp_version.n_reference_segs)])

    def get_reference_segs(self, image_id):
        if self.dataloader is not None:
            idx = self.dataloader.dataset.image_ids.index(image_id)
            return self.dataloader.dataset.__getitem__(idx)["seg"].squeeze().numpy()
        return self._reference_segs(image_id)

    def get_gt_unc_map(self, image_id):
        if self.exp_version.gt_unc_map_loading is None:
            return np.var(self._reference_segs(image_id), axis=0)  # experiment_dataloader.py:142
        return instantiate(dict(self.exp_version.gt_unc_map_loading), image_id=image_id, dataloader=self.dataloader)

    def get_mean_pred_seg(self, image_id):
        tag = "mean" if self.exp_version.pred_model != "Softmax" else "01"
        p = self.pred_seg_dir / f"{image_id}_{tag}{self.exp_version.image_ending}"
        if self.exp_version.pred_seg_loading is None:
            return nifti.load(p)[0]
        return instantiate(dict(self.exp_version.pred_seg_loading), pred_seg_path=p)

    def get_unc_map(self, image_id, unc_type):
        return nifti.load(self.unc_path_dict[unc_type] / f"{image_id}{self.exp_version.unc_ending}")[0]


def aggregate_uncertainties(exp_dataloader: ExperimentDataloader, aggregations):
    """aggregate_uncertainties.py:70-95: for every uncertainty type, image and aggregation config
    ({"_target_": ..., **params}) -> aggregated_<unc>.json.  `_target_`s naming the reference's functions are
    re-pointed to values_amd.aggregation (GPU)."""
    from .io import TARGET_MAP
    ref = "evaluation.uncertainty_aggregation.aggregate_uncertainties."
    for fn in ("patch_level_aggregation", "image_level_aggregation", "threshold_aggregation"):
        TARGET_MAP.setdefault(ref + fn, "values_amd.aggregation." + fn)
        TARGET_MAP.setdefault("uncertainty_aggregation.aggregate_uncertainties." + fn, "values_amd.aggregation." + fn)
    ending = exp_dataloader.exp_version.unc_ending
    for unc, unc_path in exp_dataloader.unc_path_dict.items():
        all_uncs = {}
        for image_id in exp_dataloader.image_ids:
            key = f"{image_id}{ending}"
            all_uncs[key] = {}
            unc_image, _ = nifti.load(unc_path / key)  # loaded once per image (the reference reloads per aggregation)
            for name, cfg in aggregations.items():
                all_uncs[key][name] = instantiate(dict(cfg), image=unc_image,
                                                  pred_model=exp_dataloader.exp_version.pred_model, unc_type=unc)
        with open(exp_dataloader.dataset_path / f"aggregated_{unc}.json", "w") as f:
            json.dump(all_uncs, f, indent=4)
