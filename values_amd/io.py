"""Checkpoint -> models, without Lightning / hydra / omegaconf installed.

Mirrors `load_models_from_checkpoint` (uncertainty_modeling/test_3D.py:222-247): a Lightning checkpoint is a dict
{"state_dict": {"model.<name>": tensor}, "hyper_parameters": {"model": {"_target_": ..., **kwargs}, ...}}
(written by lightning_experiment.py:55, main.py:83-84).  The first dotted component of every key is stripped, the
model is instantiated from `hyper_parameters["model"]` (with `aleatoric_loss` forwarded when present) and the
state dict loaded -- the reference's `_target_` strings are re-pointed to the HIP-backed classes of this package.
"""
from __future__ import annotations

import importlib
from collections import OrderedDict
from typing import Dict, List

import torch

# reference hydra targets -> MI355X-native classes
TARGET_MAP = {
    "uncertainty_modeling.models.unet3D_module.UNet3D": "values_amd.unet3d.UNet3D",
    "models.unet3D_module.UNet3D": "values_amd.unet3d.UNet3D",
    "uncertainty_modeling.models.ssn_unet3D_module.SsnUNet3D": "values_amd.ssn.SsnUNet3D",
    "models.ssn_unet3D_module.SsnUNet3D": "values_amd.ssn.SsnUNet3D",
}


def instantiate(cfg: Dict, **overrides):
    """Minimal hydra.utils.instantiate for plain dicts: {"_target_": "pkg.mod.Class", **kwargs}."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    target = TARGET_MAP.get(target, target)
    cfg.pop("_partial_", None)
    cfg.update(overrides)
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**cfg)


def load_models_from_checkpoint(checkpoints: List[Dict], device="cuda") -> List[torch.nn.Module]:
    all_models = []
    for checkpoint in checkpoints:
        hparams = checkpoint["hyper_parameters"]
        state_dict = OrderedDict()
        for k, v in checkpoint["state_dict"].items():
            state_dict[".".join(k.split(".")[1:])] = v  # test_3D.py:237-238
        if "aleatoric_loss" in hparams and hparams["aleatoric_loss"] is not None:
            model = instantiate(hparams["model"], aleatoric_loss=hparams["aleatoric_loss"])
        else:
            model = instantiate(hparams["model"])
        model.load_state_dict(state_dict=state_dict)
        all_models.append(model.to(device))
    return all_models


def load_checkpoints(paths: List[str]):
    """torch.load each path (test_3D.py:635-639) -> (list of checkpoint dicts, hparams of the first)."""
    cks = [torch.load(p, map_location="cpu", weights_only=False) for p in paths]
    return cks, cks[0]["hyper_parameters"]
