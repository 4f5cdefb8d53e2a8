"""2D results directory (Tester.create_save_dirs / save_prediction / save_uncertainty, test_2D.py:75-89, 116-159):

    <save_dir>/pred_seg/<image_id>_mean.png, <image_id>_NN.png     colour arg-max masks (NN = 01.. per prediction)
    <save_dir>/<unc_type>/<image_id>.tif                           float32 uncertainty maps

The label -> colour table is the Cityscapes train-id palette extended by the five "_2" shift classes of the
GTA/Cityscapes setup (uncertainty_modeling/data/cityscapes_labels.py:59-102, trainId2color; 255 = unlabeled = black).
Arg-max and colour lookup run on the device (vx_unc_reduce's sample_argmax, vx_colorize_u8); the files are written
with values_amd.image_io.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .image_io import write_png, write_tiff_f32

UNLABELED = 255  # cs_labels.name2trainId["unlabeled"]
TRAINID2COLOR = {
    0: (128, 64, 128), 1: (244, 35, 232), 2: (70, 70, 70), 3: (102, 102, 156), 4: (190, 153, 153), 5: (153, 153, 153),
    6: (250, 170, 30), 7: (220, 220, 0), 8: (107, 142, 35), 9: (152, 251, 152), 10: (70, 130, 180), 11: (220, 20, 60),
    12: (255, 0, 0), 13: (0, 0, 142), 14: (0, 0, 70), 15: (0, 60, 100), 16: (0, 80, 100), 17: (0, 0, 230),
    18: (119, 11, 32), 19: (46, 247, 180), 20: (167, 242, 242), 21: (30, 193, 252), 22: (242, 160, 19), 23: (84, 86, 22),
    255: (0, 0, 0),
}


def _lut() -> np.ndarray:
    lut = np.zeros((256, 3), dtype=np.uint8)
    for k, v in TRAINID2COLOR.items():
        lut[k] = v
    return lut


def colorize(labels: torch.Tensor, ignore_index_map=None) -> torch.Tensor:
    """labels (..., H, W) uint8 on the device -> (..., H, W, 3) uint8 RGB; pixels of the ignore map become unlabeled."""
    _lib.require_gpu()
    lib = _lib.load()
    lab = labels.to(torch.uint8).contiguous()
    dev = lab.device
    lut = torch.from_numpy(_lut()).to(dev)
    ign = None
    if ignore_index_map is not None:
        ign = torch.as_tensor(np.asarray(ignore_index_map) if not isinstance(ignore_index_map, torch.Tensor) else ignore_index_map)
        ign = (ign != 0).to(device=dev, dtype=torch.uint8).expand_as(lab).contiguous()
    out = torch.empty(tuple(lab.shape) + (3,), dtype=torch.uint8, device=dev)
    _lib.check(lib.vx_colorize_u8(lab.data_ptr(), None if ign is None else ign.data_ptr(), lab.numel(), lut.data_ptr(), UNLABELED,
                                  out.data_ptr(), _lib.stream_ptr()), "vx_colorize_u8")
    return out


def create_save_dirs(save_root_dir: str, exp_name: str, version, test_split: str) -> Dict[str, str]:
    save_dir = os.path.join(save_root_dir, exp_name, "test_results", str(version), test_split)
    pred = os.path.join(save_dir, "pred_seg")
    os.makedirs(pred, exist_ok=True)
    return {"save_dir": save_dir, "save_pred_dir": pred, "save_pred_prob_dir": os.path.join(save_dir, "pred_prob")}


def save_prediction(save_pred_dir: str, image_id: str, pred_masks: torch.Tensor, mean_mask: Optional[torch.Tensor],
                    ignore_index_map=None) -> None:
    """pred_masks (Npred, H, W) uint8 = arg-max of each prediction, mean_mask (H, W) = arg-max of the mean prediction.
    File names as test_2D.py:136-141: with several predictions `<id>_mean.png` then `<id>_01.png`...; with one, `<id>_01.png`."""
    n = pred_masks.shape[0]
    if n > 1:
        stack = torch.cat([mean_mask.unsqueeze(0).to(pred_masks.device), pred_masks], 0)
        names = [f"{image_id}_mean"] + [f"{image_id}_{str(i).zfill(2)}" for i in range(1, n + 1)]
    else:
        stack, names = pred_masks, [f"{image_id}_01"]
    rgb = colorize(stack, ignore_index_map).cpu().numpy()
    for img, name in zip(rgb, names):
        write_png(os.path.join(save_pred_dir, f"{name}.png"), img)


def save_uncertainty(save_dir: str, image_id: str, uncertainty_dict: Dict[str, torch.Tensor]) -> None:
    """test_2D.py:151-158: one float32 TIFF per uncertainty type."""
    for unc_type, unc_map in uncertainty_dict.items():
        d = os.path.join(save_dir, unc_type)
        os.makedirs(d, exist_ok=True)
        m = unc_map.detach().cpu().numpy() if isinstance(unc_map, torch.Tensor) else np.asarray(unc_map)
        write_tiff_f32(os.path.join(d, f"{image_id}.tif"), m.astype(np.float32))
