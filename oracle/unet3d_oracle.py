"""ORACLE (test infrastructure, not product code) -- CPU restatement of the
reference UNet3D forward pass.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.  The product path (values_amd/) never does.

Restates /root/reference/uncertainty_modeling/models/unet3D_module.py:
  * contract block  (:231-237)  Conv3d(k3,p1) -> InstanceNorm3d(affine=False,
    eps=1e-5, biased var) -> LeakyReLU(0.01) -> Dropout(p)
  * expand block    (:263-267)  Conv3d(k3,p1) -> LeakyReLU(0.01) -> Dropout(p)
  * center          (:97-121)   Conv3d,ReLU,Conv3d,ReLU,ConvTranspose3d(k2,s2),ReLU[,Dropout]
  * forward         (:296-373)  4x[contract,contract,pool] / center /
    4x[cat(up, skip), expand, expand, (convT)] / final 1x1x1

Dropout is restated as multiplication by an *injected* keep-mask times
1/(1-p) = 2 (torch.nn.Dropout in training mode, which is how the reference
runs MC-dropout: it never calls .eval()).  The masks are the ones
captured from the reference run by tools/gen_golden.py, in execution order:
  contr_1_1, contr_1_2, ..., contr_4_2, center, expand_4_1, expand_4_2, ...,
  expand_1_2   (17 masks).

Parity pin: tests/test_oracle_golden.py checks this file against
tests/golden/unet3d_{16,32}.npz, which tools/gen_golden.py produced by
importing the reference class itself (float64, as test_3D.py:425 runs it).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

DROPOUT_ORDER = [
    "contr_1_1", "contr_1_2", "contr_2_1", "contr_2_2", "contr_3_1", "contr_3_2",
    "contr_4_1", "contr_4_2", "center",
    "expand_4_1", "expand_4_2", "expand_3_1", "expand_3_2",
    "expand_2_1", "expand_2_2", "expand_1_1", "expand_1_2",
]


def _drop(x, masks, name, p):
    if masks is None or p == 0.0:
        return x
    m = masks[name]
    return x * m.to(x.dtype) * (1.0 / (1.0 - p))


def _contract(x, sd, name, masks, p, instancenorm=True):
    # unet3D_module.py:231-237
    x = F.conv3d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1)
    if instancenorm:
        x = F.instance_norm(x, eps=1e-5)
    x = F.leaky_relu(x, 0.01)
    return _drop(x, masks, name, p)


def _expand(x, sd, name, masks, p):
    # unet3D_module.py:263-267
    x = F.conv3d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1)
    x = F.leaky_relu(x, 0.01)
    return _drop(x, masks, name, p)


def unet3d_forward(sd, x, masks=None, p=0.5, instancenorm=True, aleatoric_loss=False,
                   num_classes=None, taps=None):
    """sd: dict name -> torch tensor (reference state-dict names), x: (N,Cin,D,H,W).
    masks: None (dropout off) or dict name -> bool tensor shaped like that layer's output.
    taps: optional dict that receives intermediate activations (for bisecting)."""
    def tap(k, v):
        if taps is not None:
            taps[k] = v
        return v

    c11 = tap("contr_1_1", _contract(x, sd, "contr_1_1", masks, p, instancenorm))
    c1 = tap("contr_1_2", _contract(c11, sd, "contr_1_2", masks, p, instancenorm))
    pool = F.max_pool3d(c1, 2, 2)
    c21 = tap("contr_2_1", _contract(pool, sd, "contr_2_1", masks, p, instancenorm))
    c2 = tap("contr_2_2", _contract(c21, sd, "contr_2_2", masks, p, instancenorm))
    pool = F.max_pool3d(c2, 2, 2)
    c31 = tap("contr_3_1", _contract(pool, sd, "contr_3_1", masks, p, instancenorm))
    c3 = tap("contr_3_2", _contract(c31, sd, "contr_3_2", masks, p, instancenorm))
    pool = F.max_pool3d(c3, 2, 2)
    c41 = tap("contr_4_1", _contract(pool, sd, "contr_4_1", masks, p, instancenorm))
    c4 = tap("contr_4_2", _contract(c41, sd, "contr_4_2", masks, p, instancenorm))
    pool = F.max_pool3d(c4, 2, 2)

    # center, unet3D_module.py:97-121
    h = F.relu(F.conv3d(pool, sd["center.0.weight"], sd["center.0.bias"], padding=1))
    h = F.relu(F.conv3d(h, sd["center.2.weight"], sd["center.2.bias"], padding=1))
    h = F.relu(F.conv_transpose3d(h, sd["center.4.weight"], sd["center.4.bias"], stride=2))
    center = tap("center", _drop(h, masks, "center", p))

    # decoder, unet3D_module.py:329-358 (center_crop is the identity for sizes divisible by 16)
    cat = torch.cat([center, c4], 1)
    e = tap("expand_4_1", _expand(cat, sd, "expand_4_1", masks, p))
    e = tap("expand_4_2", _expand(e, sd, "expand_4_2", masks, p))
    up = F.conv_transpose3d(e, sd["upscale4.weight"], sd["upscale4.bias"], stride=2)
    cat = torch.cat([up, c3], 1)
    e = tap("expand_3_1", _expand(cat, sd, "expand_3_1", masks, p))
    e = tap("expand_3_2", _expand(e, sd, "expand_3_2", masks, p))
    up = F.conv_transpose3d(e, sd["upscale3.weight"], sd["upscale3.bias"], stride=2)
    cat = torch.cat([up, c2], 1)
    e = tap("expand_2_1", _expand(cat, sd, "expand_2_1", masks, p))
    e = tap("expand_2_2", _expand(e, sd, "expand_2_2", masks, p))
    up = F.conv_transpose3d(e, sd["upscale2.weight"], sd["upscale2.bias"], stride=2)
    cat = torch.cat([up, c1], 1)
    e = tap("expand_1_1", _expand(cat, sd, "expand_1_1", masks, p))
    e = tap("expand_1_2", _expand(e, sd, "expand_1_2", masks, p))

    if aleatoric_loss:
        out = F.conv3d(e, sd["final_aleatoric.weight"], sd["final_aleatoric.bias"])
        mu, s = out.split(num_classes, 1)
        return mu, s
    return F.conv3d(e, sd["final.weight"], sd["final.bias"])


def conv3d_k3_naive(x, w, b):
    """Independent numpy direct 3x3x3 correlation (pad 1) for tiny shapes;
    used by the oracle self-test to make sure F.conv3d is what we think it is."""
    import numpy as np

    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    n, cin, d, h, wd = x.shape
    cout = w.shape[0]
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1), (1, 1)))
    out = np.zeros((n, cout, d, h, wd))
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                patch = xp[:, :, kz:kz + d, ky:ky + h, kx:kx + wd]
                out += np.einsum("ncdhw,oc->nodhw", patch, w[:, :, kz, ky, kx])
    return out + np.asarray(b, dtype=np.float64).reshape(1, -1, 1, 1, 1)
