#!/usr/bin/env python3
"""Same-process A/B of the z-column kernel of the Cout = 16 layers (conv3d_zc16.hip) against the tile kernel on single launches:
the knob vx_config.s16_no_zc16 switches the dispatch, the packed weights (family 6) serve both.  Rounds of `--reps` launches
alternate A, B, A, B; prints the median per kernel and the ratio.

    python tools/ab_zc16.py [--N 320] [--edge 32] [--reps 20] [--rounds 8]
"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=320)
ap.add_argument("--edge", type=int, default=32)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--only", default="")
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
N, E = args.N, args.edge
g = torch.Generator(device="cpu").manual_seed(1)

# name, cin, act, drop, stats, pre, pool (zc16 only), out_split
SPECS = [("contr_2_1  8->16 stats", 8, 0, 0, 1, 0, 0, 0),
         ("contr_2_2 16->16 stats + prologue", 16, 0, 0, 1, 1, 0, 0),
         ("contr_2_2 16->16 stats + prologue + pool [zc16 only]", 16, 0, 0, 1, 1, 1, 0),
         ("16->16 stats (no prologue)", 16, 0, 0, 1, 0, 0, 0),
         ("expand_2_2 16->16 lrelu + drop + out_split", 16, 1, 1, 0, 0, 0, 1)]


def cfg_set(**kw):
    c = _lib.Config()
    _lib.check(lib.vx_get_config(C.byref(c)), "get")
    for k, v in kw.items():
        setattr(c, k, v)
    _lib.check(lib.vx_set_config(C.byref(c)), "set")


for name, cin, act, drop, stats, pre, pool, osplit in SPECS:
    if args.only and args.only not in name:
        continue
    x = torch.randn((N, E, E, E, cin), generator=g).to(dev)
    w = (torch.randn((16, cin, 3, 3, 3), generator=g) * (1.0 / (27 * cin)) ** 0.5).to(dev)
    b = (torch.randn((16,), generator=g) * 0.1).to(dev)
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, 16), dtype=torch.float32, device=dev)
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(w), _lib.ptr(wp), cin, 16, _lib.stream_ptr()), "pack")
    out = torch.empty((N, E, E, E, 16), dtype=torch.float32, device=dev)
    mean = torch.zeros((N, cin), device=dev); rstd = torch.ones((N, cin), device=dev)
    st = torch.zeros((N, lib.vx_conv3d_k3_tiles(E, E, E), 16, 2), dtype=torch.float32, device=dev)
    praw = torch.empty((N, E, E // 2, E // 2, 16), dtype=torch.float32, device=dev)
    pfl = torch.empty((N, E, E // 2, E // 2, 4), dtype=torch.int32, device=dev)

    def args_for(zc):
        a = _lib.ConvArgs()
        a.w_family = lib.vx_conv3d_k3_family(cin, 16)
        a.in_ = x.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = b.data_ptr(); a.out = out.data_ptr()
        a.in_pitch, a.out_pitch, a.out_coff = cin, 16, 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = N, E, E, E, cin, 16
        a.act, a.drop_mode, a.drop_seed, a.drop_layer = act, drop, 1, 2
        a.out_split = osplit
        if stats:
            a.stats_partial = st.data_ptr()
        if pre:
            a.in_mean, a.in_rstd, a.in_repeat = mean.data_ptr(), rstd.data_ptr(), 1
            a.in_drop_mode, a.in_drop_seed, a.in_drop_layer = 1, 3, 4
        if pool and zc:
            a.pool_out, a.pool_flags = praw.data_ptr(), pfl.data_ptr()
            a.drop_mode, a.drop_seed, a.drop_layer = 1, 1, 5
        return a
    s = _lib.stream_ptr()
    times = {"tile": [], "zc16": []}
    names = {}
    for rnd in range(args.rounds + 1):
        for which in ("tile", "zc16"):
            cfg_set(s16_no_zc16=1 if which == "tile" else 0)
            a = args_for(which == "zc16")
            _lib.check(lib.vx_conv3d_k3(C.byref(a), s), "warm")
            names[which] = lib.vx_last_kernel_name().decode()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                lib.vx_conv3d_k3(C.byref(a), s)
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[which].append(e0.elapsed_time(e1) / args.reps)
    cfg_set(s16_no_zc16=0)
    mt, mz = statistics.median(times["tile"]), statistics.median(times["zc16"])
    flops = 2.0 * 27 * cin * 16 * N * E ** 3
    print(f"{name:58s} tile {mt:.4f} ms ({flops / mt / 1e9:.0f} TF)  zc16 {mz:.4f} ms ({flops / mz / 1e9:.0f} TF)  zc16/tile {mz / mt:.3f}   [{names['tile']} | {names['zc16']}]",
          flush=True)
