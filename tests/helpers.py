"""Shared helpers for the parity tests (fixture loading, mask unpacking)."""
import os

import numpy as np
import torch

from tests.formula import formula_unet3d_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def unpack_masks(g, t):
    """Dropout keep-masks of pass t as bool arrays (reference NCDHW shapes)."""
    from oracle.unet3d_oracle import DROPOUT_ORDER
    out = {}
    for name in DROPOUT_ORDER:
        shape = tuple(int(v) for v in g[f"maskshape_{name}"])
        n = int(np.prod(shape))
        out[name] = np.unpackbits(g[f"mask_{t}_{name}"])[:n].astype(bool).reshape(shape)
    return out


def formula_sd_torch(seed_tag=0, dtype=torch.float64):
    return {k: torch.from_numpy(v).to(dtype) for k, v in formula_unet3d_state_dict(seed_tag=seed_tag).items()}
