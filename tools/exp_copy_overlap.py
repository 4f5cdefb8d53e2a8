import sys, time, torch
sys.path.insert(0, "/root/repo")
from values_amd import UNet3D, predict_uncertainty
dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = UNet3D(num_classes=2, do_dropout=True).to(dev)
x = torch.randn((32, 1, 64, 64, 64), device=dev)
big = torch.empty(176 * 1024 * 1024 // 4, device=dev)
h = torch.empty(big.shape, dtype=big.dtype).pin_memory()
cs = torch.cuda.Stream()
def compute():
    return predict_uncertainty([m], x, n_pred=10, seeds=[1])
def copy():
    with torch.cuda.stream(cs):
        h.copy_(big, non_blocking=True)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("compute alone %.2f ms" % timed(compute))
print("D2H alone     %.2f ms" % timed(copy))
def both():
    o = compute(); copy(); return o
print("both          %.2f ms" % timed(both))
h2 = torch.empty(big.shape, dtype=big.dtype).pin_memory()
def up():
    with torch.cuda.stream(cs):
        big.copy_(h2, non_blocking=True)
print("H2D alone     %.2f ms" % timed(up))
def both2():
    o = compute(); up(); return o
print("compute + H2D %.2f ms" % timed(both2))
