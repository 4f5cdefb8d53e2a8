#!/usr/bin/env python3
"""Same-process A/B of two builds of the library on single conv launches (the guide's rule: interleave in ONE process,
n >= 10 -- separate runs of bench.py drift by +-5 % with the clock on some boxes).

    python tools/ab_layers.py [--base values_amd/libvalues_amd_base.so] [--new values_amd/libvalues_amd.so] [specs ...]

spec = cin:cout:edge:act:drop:head[:up[:pre[:pool]]] as tools/stamp_s16.py.  Both libraries are loaded RTLD_LOCAL (their
symbols do not interpose), each packs its own weights; rounds of `--reps` launches alternate A, B, A, B ...; prints the
median per build and the ratio."""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--base", default=os.path.join(ROOT, "values_amd", "libvalues_amd_base.so"))
ap.add_argument("--new", default=os.path.join(ROOT, "values_amd", "libvalues_amd.so"))
ap.add_argument("--N", type=int, default=320)
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
    cin, cout, edge, act, drop, head = f[:6]
    up = f[6] if len(f) > 6 else 0
    pre = f[7] if len(f) > 7 else 0
    pool = f[8] if len(f) > 8 else 0
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn((N, edge, edge, edge, 8 if up else cin), generator=g).to(dev)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.05).to(dev); b = torch.zeros(cout, device=dev)
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev)
    assert lib.vx_pack_conv3d_k3(_lib.ptr(w), _lib.ptr(wp), cin, cout, _lib.stream_ptr()) == 0
    out = torch.empty((N, edge, edge, edge, cout), device=dev)
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, cout)
    a.in_ = x.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = b.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = (8 if up else cin), cout, 0
    keep.extend([x, w, b, wp, out])
    if up:
        coarse = torch.randn((N, edge // 2, edge // 2, edge // 2, 16), generator=g).to(dev)
        uw = (torch.randn((16, 8, 2, 2, 2), generator=g) * 0.2).to(dev); ub = torch.zeros(8, device=dev)
        uwp = torch.empty(lib.vx_convT_k2s2_packed_floats(16, 8), dtype=torch.float32, device=dev)
        assert lib.vx_pack_convT_k2s2(_lib.ptr(uw), _lib.ptr(uwp), 16, 8, _lib.stream_ptr()) == 0
        a.up_in, a.up_w, a.up_b, a.up_pitch = coarse.data_ptr(), uwp.data_ptr(), ub.data_ptr(), 16
        keep.extend([coarse, uw, ub, uwp])
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = N, edge, edge, edge, cin, cout
    if pre:
        rep = 10 if (cin == 8 and N % 10 == 0) else 1
        if rep > 1:
            x2 = torch.randn((N // rep, edge, edge, edge, cin), generator=g).to(dev); a.in_ = x2.data_ptr(); keep.append(x2)
        mean = torch.zeros((N // rep, cin), device=dev); rstd = torch.ones((N // rep, cin), device=dev)   # (one row of Cin statistics per source sample)
        a.in_mean, a.in_rstd, a.in_drop_mode, a.in_drop_seed, a.in_drop_layer, a.in_repeat = mean.data_ptr(), rstd.data_ptr(), 1, 7, 1, rep
        keep.extend([mean, rstd])
    a.act, a.drop_mode, a.drop_seed, a.drop_layer = act, drop, 1, 2
    if pool:
        praw = torch.empty((N, edge // 2, edge // 2, edge // 2, 8), device=dev)
        pfl = torch.empty((N, edge // 2, edge // 2, edge // 2, 2), dtype=torch.int32, device=dev)
        a.pool_out, a.pool_flags = praw.data_ptr(), pfl.data_ptr()
        a.drop_mode, a.drop_seed, a.drop_layer = 1, 1, 1
        keep.extend([praw, pfl])
    if not act:
        st = torch.zeros((N, lib.vx_conv3d_k3_tiles(edge, edge, edge), cout, 2), device=dev)
        a.stats_partial = st.data_ptr(); keep.append(st)
    if head:
        hw = torch.randn((2, cout), generator=g).to(dev); hb = torch.zeros(2, device=dev)
        ho = torch.empty((N, 2, edge, edge, edge), device=dev)
        a.out = None
        a.head_out, a.head_w, a.head_b, a.head_C = ho.data_ptr(), hw.data_ptr(), hb.data_ptr(), 2
        keep.extend([hw, hb, ho])
    return a


def run(lib, a, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rc = lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr())
        assert rc == 0, lib.vx_last_error_string()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


specs = args.specs or ["8:8:64:0:0:0:0:1:1", "16:8:64:1:1:0:1:1", "8:8:64:1:1:1"]
for spec in specs:
    A = {k: make(l, spec) for k, l in libs.items()}
    for k in libs:
        run(libs[k], A[k], args.reps)            # warm-up (clocks, code objects)
    t = {k: [] for k in libs}
    for r in range(args.rounds):
        for k in (("base", "new") if r % 2 == 0 else ("new", "base")):
            t[k].append(run(libs[k], A[k], args.reps))
    mb, mn = statistics.median(t["base"]), statistics.median(t["new"])
    kn = libs["new"].vx_last_kernel_name().decode()
    print(f"{spec:24s} base {mb:.4f} ms  new {mn:.4f} ms  new/base {mn / mb:.4f}   (min {min(t['base']):.4f} / {min(t['new']):.4f})  {kn}", flush=True)
