import sys, torch, numpy as np, torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from values_amd import _lib
import test_gpu_kernels as T
lib = _lib.load()
n, d, h, w = 1, 8, 8, 32
def run(exact_c, exact_w):
    g = torch.Generator().manual_seed(1)
    coarse = torch.randn((n, 16, d // 2, h // 2, w // 2), generator=g)
    uw = torch.randn((16, 8, 2, 2, 2), generator=g) * 0.25
    if exact_c: coarse = coarse.half().float()
    if exact_w: uw = uw.half().float()
    ub = torch.randn(8, generator=g) * 0.3
    raw = torch.zeros((n, 8, d, h, w))
    wt = torch.zeros((8, 16, 3, 3, 3)); 
    for c in range(8): wt[c, c, 1, 1, 1] = 1.0     # identity on the up half: out = up
    b = torch.zeros(8)
    up_ref = F.conv_transpose3d(coarse.double(), uw.double(), ub.double(), stride=2)
    xd = T.cl(raw).to("cuda")
    cd = T.cl(coarse).to("cuda")
    got, _, _, _ = T._xp8_conv(xd, 16, wt, b, n, d, h, w, act=0, xblk=0, in_pitch=8, up=(cd, uw, ub))
    e = (got.double() - up_ref).abs()
    print("exact_c", exact_c, "exact_w", exact_w, "max err", e.max().item(), "mean", e.mean().item(), "kernel", lib.vx_last_kernel_name().decode())
    return e
for ec in (1, 0):
    for ew in (1, 0):
        e = run(ec, ew)
print("err by z parity", [e[..., z::2, :, :].max().item() for z in (0, 1)], "y", [e[..., :, y::2, :].max().item() for y in (0, 1)], "x", [e[..., x::2].max().item() for x in (0, 1)])
