import sys, time, os, torch
sys.path.insert(0, "/root/repo")
from values_amd import UNet3D, predict_uncertainty, HostPipeline
dev = torch.device("cuda", 0)
torch.manual_seed(123)
model = UNet3D(num_classes=2, do_dropout=True).to(dev)
x = torch.randn((32, 1, 64, 64, 64)).to(dev)
for i in range(8):
    predict_uncertainty([model], x, n_pred=10, seeds=[i])
torch.cuda.synchronize()
xh = x.cpu().pin_memory()
hp = HostPipeline([model], n_pred=10)
for i in range(3):
    hp.submit(xh, seeds=[i])
hp.flush(); torch.cuda.synchronize()
t0 = time.perf_counter(); ts = []
for i in range(40):
    hp.submit(xh, seeds=[100 + i]); ts.append((time.perf_counter() - t0) * 1e3)
hp.flush(); torch.cuda.synchronize()
print("submit return times (ms):", " ".join(f"{b - a:.1f}" for a, b in zip([0] + ts[:-1], ts)), " total %.1f" % ((time.perf_counter() - t0) * 1e3))
