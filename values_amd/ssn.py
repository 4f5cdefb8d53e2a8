"""Stochastic segmentation network on MI355X: the reference's SsnUNet3D surface (ssn_unet3D_module.py:7-70).

Same constructor keywords and state-dict keys as the reference class (`final.*` of the parent with
num_classes * (2 + rank) channels is kept as a parameter container although its forward never uses it, plus
`mean_conv`, `log_cov_diag_conv`, `cov_factor_conv`).  The three 1x1x1 heads act on the same decoder features, so
they run as ONE head of (2 + rank) * C channels at the end of vx_unet3d_forward; `forward` returns a
`LowRankNormal` whose `sample([n])` is vx_ssn_sample (torch.distributions.LowRankMultivariateNormal.rsample's
formula), shaped like the reference's `distribution.sample([n_pred])`: (n, batch, C * D*H*W).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .unet3d import UNet3D


class LowRankNormal:
    """What predict_cases_ssn needs from the distribution object (test_3D.py:375-385): sample(), mean."""

    def __init__(self, head: torch.Tensor, num_classes: int, rank: int, epsilon: float, seed: int):
        self._head = head            # (B, (2 + R) * C, D, H, W) float32: mean | log_cov_diag | cov_factor
        self.num_classes, self.rank, self.epsilon = num_classes, rank, epsilon
        self._seed = seed
        self._draws = 0

    @property
    def batch_shape(self):
        return torch.Size([self._head.shape[0]])

    @property
    def event_shape(self):
        return torch.Size([self.num_classes * self._head[0, 0].numel()])

    @property
    def mean(self) -> torch.Tensor:
        b = self._head.shape[0]
        return self._head[:, :self.num_classes].reshape(b, -1)

    def sample_volumes(self, n: int, eps_w: Optional[torch.Tensor] = None, eps_d: Optional[torch.Tensor] = None,
                       seed: Optional[int] = None) -> torch.Tensor:
        """(B, n, C, D, H, W) logit samples -- the layout vx_unc_reduce reads.  eps_w (n, B, R) / eps_d
        (n, B, C * vox) inject the standard normals (parity); otherwise they are generated on the device."""
        _lib.require_gpu()
        lib = _lib.load()
        h = self._head
        B, C, R = h.shape[0], self.num_classes, self.rank
        sp = tuple(h.shape[2:])
        nvox = int(h[0, 0].numel())
        out = torch.empty((B, n, C) + sp, dtype=torch.float32, device=h.device)
        hold = []
        pw = pd = None
        if eps_w is not None:
            ew = eps_w.to(device=h.device, dtype=torch.float32).contiguous()
            hold.append(ew); pw = ew.data_ptr()
        if eps_d is not None:
            ed = eps_d.to(device=h.device, dtype=torch.float32).contiguous()
            hold.append(ed); pd = ed.data_ptr()
        if seed is None:
            seed = (self._seed * 7919 + self._draws) & 0xFFFFFFFF
            self._draws += 1
        _lib.check(lib.vx_ssn_sample(h.data_ptr(), pw, pd, int(seed) & 0xFFFFFFFF, B, n, C, R, nvox, float(self.epsilon),
                                     out.data_ptr(), _lib.stream_ptr()), "vx_ssn_sample")
        self._hold = hold
        return out

    def sample(self, sample_shape=(1,), **kw) -> torch.Tensor:
        n = int(sample_shape[0]) if len(sample_shape) else 1
        v = self.sample_volumes(n, **kw)                      # (B, n, C, *sp)
        return v.transpose(0, 1).reshape(n, v.shape[0], -1)   # (n, B, C * vox), as the reference returns it

    rsample = sample


class SsnUNet3D(UNet3D):
    def __init__(self, num_classes: int, in_channels: int = 1, initial_filter_size: int = 8, kernel_size: int = 3,
                 do_instancenorm: bool = True, do_dropout: bool = False, rank: int = 10, epsilon: float = 1e-5):
        super().__init__(num_classes * 2 + num_classes * rank, in_channels, initial_filter_size, kernel_size,
                         do_instancenorm, do_dropout, aleatoric_loss=False)
        self.num_classes = num_classes
        self.epsilon = epsilon
        self.rank = rank
        self.mean_conv = nn.Conv3d(initial_filter_size, num_classes, kernel_size=1)
        self.log_cov_diag_conv = nn.Conv3d(initial_filter_size, num_classes, kernel_size=1)
        self.cov_factor_conv = nn.Conv3d(initial_filter_size, num_classes * rank, kernel_size=1)

    def _head_params(self, sd):
        names = ("mean_conv", "log_cov_diag_conv", "cov_factor_conv")
        w = torch.cat([sd[n + ".weight"].reshape(sd[n + ".weight"].shape[0], -1) for n in names], 0).contiguous()
        b = torch.cat([sd[n + ".bias"] for n in names], 0).contiguous()
        return w, b

    def forward(self, x: torch.Tensor, enable_concat: bool = True, mean_only: bool = False, **kw):
        if not enable_concat:
            raise NotImplementedError("values_amd.SsnUNet3D: enable_concat=False is training-only")
        head = self._run(x, **kw).float()
        dist = LowRankNormal(head, self.num_classes, 0 if mean_only else self.rank, self.epsilon, self.seed + self._calls)
        return dist
