"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's stochastic segmentation
network head and sampling.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates /root/reference/uncertainty_modeling/models/ssn_unet3D_module.py:39-70 (SsnUNet3D.forward: backbone with
last_layer=False, mean / log_cov_diag / cov_factor 1x1x1 heads, view/flatten/transpose of the factor) and the
sampling of predict_cases_ssn (test_3D.py:373-385): `distribution.sample([n_pred])` of
torch.distributions.LowRankMultivariateNormal, whose rsample is
    loc + cov_factor @ eps_W + sqrt(cov_diag) * eps_D          (torch/distributions/lowrank_multivariate_normal.py)
with eps_W ~ N(0, I_rank) per sample and eps_D ~ N(0, I) per element.  The normals are inputs here.

Parity pin: tests/test_oracle_golden.py::test_ssn_oracle_matches_reference vs tests/golden/ssn_16.npz (the imported
reference class with formula weights; its normals captured by the generator, tools/gen_golden.py:gen_ssn).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .unet3d_oracle import unet3d_forward


def ssn_distribution(sd, x, num_classes: int, rank: int, epsilon: float = 1e-5):
    """-> loc (B, C*vox), cov_diag (B, C*vox), cov_factor (B, C*vox, R) as the reference builds them (:44-57)."""
    taps = {}
    unet3d_forward(sd, x, masks=None, taps=taps)          # dropout off: shipped ssn config has do_dropout False
    feat = taps["expand_1_2"]                              # UNet3D.forward(last_layer=False) output
    b = feat.shape[0]
    mean = F.conv3d(feat, sd["mean_conv.weight"], sd["mean_conv.bias"]).reshape(b, -1)
    cov_diag = (F.conv3d(feat, sd["log_cov_diag_conv.weight"], sd["log_cov_diag_conv.bias"]).exp() + epsilon).reshape(b, -1)
    fac = F.conv3d(feat, sd["cov_factor_conv.weight"], sd["cov_factor_conv.bias"])
    fac = fac.reshape(b, rank, num_classes, -1).flatten(2, 3).transpose(1, 2)
    return mean, cov_diag, fac


def lowrank_rsample(loc, cov_diag, cov_factor, eps_w, eps_d):
    """loc/cov_diag (B, E), cov_factor (B, E, R), eps_w (S, B, R), eps_d (S, B, E) -> samples (S, B, E)."""
    loc, cov_diag, cov_factor = (np.asarray(a, dtype=np.float64) for a in (loc, cov_diag, cov_factor))
    eps_w, eps_d = np.asarray(eps_w, dtype=np.float64), np.asarray(eps_d, dtype=np.float64)
    low = np.einsum("ber,sbr->sbe", cov_factor, eps_w)
    return loc[None] + low + np.sqrt(cov_diag)[None] * eps_d
