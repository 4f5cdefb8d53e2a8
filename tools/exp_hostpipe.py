"""Experiment behind values_amd.HostPipeline: which part of a host-inclusive step costs what."""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from values_amd import UNet3D, predict_uncertainty, HostPipeline
dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = UNet3D(num_classes=2, do_dropout=True).to(dev)
x = torch.randn((32, 1, 64, 64, 64), device=dev)
xh = x.cpu().pin_memory()
KEYS = HostPipeline.KEYS
def timed(fn, n=15):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("device only                 %.2f ms" % timed(lambda i: predict_uncertainty([m], x, n_pred=10, seeds=[i])))
down, up = torch.cuda.Stream(), torch.cuda.Stream()
bufs = [None] * 3
keep = [None] * 3
evs = [None] * 3
def variant(upload, wait_event, one_buffer):
    def f(i):
        s = i % 3
        if wait_event and evs[s] is not None:
            evs[s].synchronize()
        if upload:
            with torch.cuda.stream(up):
                xd = xh.to(dev, non_blocking=True)
            torch.cuda.current_stream().wait_stream(up)
        else:
            xd = x
        out = predict_uncertainty([m], xd, n_pred=10, seeds=[i])
        if bufs[s] is None:
            if one_buffer:
                n = sum(out[k].numel() * out[k].element_size() for k in KEYS)
                bufs[s] = torch.empty(n, dtype=torch.uint8).pin_memory()
            else:
                bufs[s] = {k: torch.empty(out[k].shape, dtype=out[k].dtype).pin_memory() for k in KEYS}
        down.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(down):
            if one_buffer:
                off = 0
                for k in KEYS:
                    nb = out[k].numel() * out[k].element_size()
                    bufs[s][off:off + nb].view(out[k].dtype).view(out[k].shape).copy_(out[k], non_blocking=True)
                    off += nb
            else:
                for k in KEYS:
                    bufs[s][k].copy_(out[k], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(down); evs[s] = ev
        keep[s] = (out, xd)
    return f
for upload in (False, True):
    for wait_event in (False, True):
        bufs[:] = [None] * 3; evs[:] = [None] * 3
        print(f"download, upload={upload!s:5} event wait={wait_event!s:5} %.2f ms" % timed(variant(upload, wait_event, False)))
hp = HostPipeline([m], n_pred=10)
print("HostPipeline                %.2f ms" % timed(lambda i: hp.submit(xh, seeds=[i]))); hp.flush()
for n in (15, 60):
    hp = HostPipeline([m], n_pred=10)
    print(f"HostPipeline {n} steps          %.2f ms" % timed(lambda i: hp.submit(xh, seeds=[i]), n=n)); hp.flush()
import time as _t
hp = HostPipeline([m], n_pred=10)
for i in range(3): hp.submit(xh, seeds=[i])
hp.flush(); torch.cuda.synchronize()
t0 = _t.perf_counter(); ts = []
for i in range(30):
    hp.submit(xh, seeds=[i]); ts.append((_t.perf_counter() - t0) * 1e3)
hp.flush(); torch.cuda.synchronize()
print("submit return times (ms):", " ".join(f"{b - a:.1f}" for a, b in zip([0] + ts[:-1], ts)), " total %.1f" % ((_t.perf_counter() - t0) * 1e3))
