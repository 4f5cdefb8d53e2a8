#!/usr/bin/env python3
"""Same-process A/B of two builds of the library on single 2D conv launches (HRNet shapes; tools/ab_layers.py is the 3D twin).

    python tools/ab_conv2d.py [--base values_amd/libvalues_amd_base.so] [--new values_amd/libvalues_amd.so] [specs ...]

spec = cin:cout:ks:stride:H:W[:pre]   (N = --N images; pre = 1: the folded BatchNorm + ReLU prologue on the input)
Input channels are laid out as the model does: pitch = round16(cin), zero tail."""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--base", default=os.path.join(ROOT, "values_amd", "libvalues_amd_base.so"))
ap.add_argument("--new", default=os.path.join(ROOT, "values_amd", "libvalues_amd.so"))
ap.add_argument("--N", type=int, default=32)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("specs", nargs="*")
args = ap.parse_args()


def open_lib(path):
    lib = C.CDLL(path, mode=os.RTLD_LOCAL)
    for name, (res, at) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = at
    return lib


libs = {"base": open_lib(args.base), "new": open_lib(args.new)}
dev = torch.device("cuda", 0)
N = args.N
keep = []


def make(lib, spec):
    f = list(map(int, spec.split(":")))
    cin, cout, ks, stride, H, W = f[:6]
    pre = f[6] if len(f) > 6 else 0
    g = torch.Generator(device="cpu").manual_seed(7)
    cp = (cin + 15) // 16 * 16
    pit = (cin + 3) // 4 * 4                                              # activations travel at pitch round4(C)
    x = torch.zeros((N, H, W, pit)); x[..., :cin] = torch.randn((N, H, W, cin), generator=g); x = x.to(dev)
    w = (torch.randn((cout, cin, ks, ks), generator=g) * 0.05).to(dev)      # packed with the REAL channel count, as the model does
    wp = torch.empty(lib.vx_conv2d_packed_floats(cin, cout, ks), dtype=torch.float32, device=dev)
    assert lib.vx_pack_conv2d(_lib.ptr(w), _lib.ptr(wp), cin, cout, ks, _lib.stream_ptr()) == 0
    oh = (H + 2 * (ks // 2) - ks) // stride + 1; ow = (W + 2 * (ks // 2) - ks) // stride + 1
    pitch = (cout + 15) // 16 * 16
    out = torch.zeros((N, oh, ow, pitch), device=dev)
    a = _lib.Conv2dArgs()
    a.w_family = lib.vx_conv2d_family(cin, cout, ks)
    a.in_ = x.data_ptr(); a.in_pitch = pit; a.w_packed = wp.data_ptr(); a.bias = None
    a.out = out.data_ptr(); a.out_pitch = pitch; a.out_coff = 0
    a.N, a.H, a.W, a.Cin, a.Cout, a.KS, a.S = N, H, W, cp, cout, ks, stride
    part = torch.empty((N * lib.vx_conv2d_tiles(H, W, ks, stride), cout, 2), device=dev)
    a.stats_partial = part.data_ptr()
    keep.extend([x, w, wp, out, part])
    if pre:
        sc = torch.ones((1, pit), device=dev); sh = torch.zeros((1, pit), device=dev)
        a.in_scale, a.in_shift, a.in_relu, a.in_cpitch, a.in_group_images = sc.data_ptr(), sh.data_ptr(), 1, pit, 0
        keep.extend([sc, sh])
    return a


def run(lib, a, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rc = lib.vx_conv2d(C.byref(a), _lib.stream_ptr())
        assert rc == 0, lib.vx_last_error_string()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


specs = args.specs or ["18:18:3:1:128:256:1", "36:36:3:1:64:128:1", "72:72:3:1:32:64:1", "144:144:3:1:16:32:1", "18:36:3:2:128:256",
                       "270:270:1:1:128:256", "64:64:3:1:128:256"]
for spec in specs:
    A = {k: make(l, spec) for k, l in libs.items()}
    for k in libs:
        run(libs[k], A[k], args.reps)
    t = {k: [] for k in libs}
    for r in range(args.rounds):
        for k in (("base", "new") if r % 2 == 0 else ("new", "base")):
            t[k].append(run(libs[k], A[k], args.reps))
    mb, mn = statistics.median(t["base"]), statistics.median(t["new"])
    f = list(map(int, spec.split(":")))
    fl = 2.0 * f[2] * f[2] * f[0] * f[1] * N * (f[4] // f[3]) * (f[5] // f[3])
    kn = libs["new"].vx_last_kernel_name().decode()
    print(f"{spec:24s} base {mb*1e3:8.1f} us  new {mn*1e3:8.1f} us  new/base {mn / mb:.4f}  new {fl / mn / 1e9:7.1f} TF useful  {kn}", flush=True)
