import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from values_amd import UNet3D, predict_uncertainty, HostPipeline
dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = UNet3D(num_classes=2, do_dropout=True).to(dev)
x = torch.randn((32, 1, 64, 64, 64), device=dev)
xh = x.cpu().pin_memory()
for i in range(4):
    predict_uncertainty([m], x, n_pred=10, seeds=[i])
torch.cuda.synchronize()
ts = []
for i in range(6):
    t0 = time.perf_counter()
    o = predict_uncertainty([m], x, n_pred=10, seeds=[i])
    ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
print("host time of one predict_uncertainty call (GPU idle at start): " + " ".join(f"{t:.2f}" for t in ts) + " ms")
for ns in (1, 2):
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        o = predict_uncertainty([m], x, n_pred=10, seeds=[i], n_streams=ns)
        ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    print(f"n_streams={ns}: " + " ".join(f"{t:.2f}" for t in ts) + " ms")
hp = HostPipeline([m], n_pred=10)
for i in range(4): hp.submit(xh, seeds=[i])
ts = []
for i in range(8):
    t0 = time.perf_counter()
    hp.submit(xh, seeds=[i])
    ts.append((time.perf_counter() - t0) * 1e3)
print("HostPipeline.submit host time back to back: " + " ".join(f"{t:.2f}" for t in ts) + " ms")
