import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, torch.nn.functional as F
from tests.test_gpu_kernels import run_conv
from values_amd.formula import formula_tensor
cin, cout, shape = 16, 32, (1,4,4,16)
n,d,h,w = shape
x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 101))
wt = torch.from_numpy(formula_tensor((cout, cin, 3, 3, 3), 102, scale=(1.0 / (27 * cin)) ** 0.5))
b = torch.from_numpy(formula_tensor((cout,), 103, scale=0.2))
ref = F.conv3d(x.float().double(), wt.float().double(), b.float().double(), padding=1)
got, st, _ = run_conv(x, wt, b, stats=False)
e = (got.double() - ref).abs()
bad = (e > 1e-3).nonzero()
print(len(bad), "bad elements")
refn = ref.numpy(); gn = got.double().numpy()
for idx in bad[:12].tolist():
    nn, c, z, y, xx = idx
    g = gn[nn, c, z, y, xx]
    hits = np.argwhere(np.abs(refn - g) < 1e-5)
    print(idx, "got", round(g,5), "ref", round(refn[nn,c,z,y,xx],5), "matches ref at", hits[:3].tolist(), "bias", round(float(b[c]),4))
got2, _, _ = run_conv(x, wt*0, b, stats=False)
print("zero weights: max dev from bias", (got2.double() - b.float().double().view(1,-1,1,1,1)).abs().max().item())
got3, _, _ = run_conv(x, wt, b*0, stats=False)
ref3 = F.conv3d(x.float().double(), wt.float().double(), None, padding=1)
print("zero bias: max err", (got3.double() - ref3).abs().max().item())
