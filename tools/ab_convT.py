#!/usr/bin/env python3
"""Same-process A/B of the deep transposed convolutions (center.4: 128 -> 64 at 4^3, upscale4: 64 -> 32 at 8^3): the split-fp16
kernel (vx_config.conv_fp32 = 0) against the native-fp32 matrix kernel (conv_fp32 = 1 selects it for this launch only).

    python tools/ab_convT.py [--N 320] [--reps 20] [--rounds 8]
"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=320)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=8)
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)


def cfg_set(**kw):
    c = _lib.Config()
    _lib.check(lib.vx_get_config(C.byref(c)), "get")
    for k, v in kw.items():
        setattr(c, k, v)
    _lib.check(lib.vx_set_config(C.byref(c)), "set")


for name, cin, cout, e, act, drop in (("center.4 128->64 4^3 relu", 128, 64, 4, _lib.VX_ACT_RELU, 0),
                                      ("upscale4 64->32 8^3", 64, 32, 8, 0, 0)):
    N = args.N
    x = torch.randn((N, e, e, e, cin), generator=g).to(dev)
    w = (torch.randn((cin, cout, 2, 2, 2), generator=g) * (1.0 / cin) ** 0.5).to(dev)
    b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    wp = torch.empty(lib.vx_convT_k2s2_packed_floats(cin, cout), dtype=torch.float32, device=dev)
    _lib.check(lib.vx_pack_convT_k2s2(_lib.ptr(w), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "packT")
    outs = {}
    times = {0: [], 1: []}
    names = {}
    s = _lib.stream_ptr()
    for rnd in range(args.rounds + 1):
        for fp32 in (1, 0):
            cfg_set(conv_fp32=fp32)
            out = outs.setdefault(fp32, torch.empty((N, 2 * e, 2 * e, 2 * e, cout), dtype=torch.float32, device=dev))
            a = _lib.ConvTArgs()
            a.in_ = x.data_ptr(); a.in_pitch = cin; a.w_packed = wp.data_ptr(); a.bias = b.data_ptr()
            a.out = out.data_ptr(); a.out_pitch = cout; a.out_coff = 0
            a.N, a.D, a.H, a.W, a.Cin, a.Cout = N, e, e, e, cin, cout
            a.act = act
            _lib.check(lib.vx_convT_k2s2(C.byref(a), s), "convT")
            names[fp32] = lib.vx_last_kernel_name().decode()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                lib.vx_convT_k2s2(C.byref(a), s)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[fp32].append(e0.elapsed_time(e1) / args.reps)
    cfg_set(conv_fp32=0)
    m1, m0 = statistics.median(times[1]), statistics.median(times[0])
    byt = 4.0 * N * e ** 3 * (cin + 8 * cout)
    err = (outs[0] - outs[1]).abs().max().item()
    print(f"{name:28s} fp32 {m1:.4f} ms ({byt / m1 / 1e9:.2f} TB/s)  split-fp16 {m0:.4f} ms ({byt / m0 / 1e9:.2f} TB/s)  ratio {m0 / m1:.3f}  max|diff| {err:.2e}  [{names[1]} | {names[0]}]", flush=True)
