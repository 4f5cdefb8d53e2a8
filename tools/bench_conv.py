#!/usr/bin/env python3
"""Micro-benchmark of single layers through the C ABI (for kernel tuning on the GPU box).

    python tools/bench_conv.py [--n 80] [--reps 10] [layer ...]
layers: name:cin:cout:edge, default = the heavy layers of the 64^3 network.
Prints ms per launch, algorithmic TFLOP/s and checks the result against torch (fp32 conv on device is NOT
available without MIOpen kernels for every shape, so the check uses a CPU float64 conv on a small crop)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from values_amd import _lib  # noqa: E402

DEFAULT = ["expand_1_1:16:8:64", "contr_1_2:8:8:64", "contr_2_1:8:16:32", "contr_2_2:16:16:32", "expand_2_1:32:16:32",
           "contr_3_2:32:32:16", "expand_3_1:64:32:16", "expand_4_1:128:64:8"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=80)
    ap.add_argument("--reps", type=int, default=300)  # short loops run at ramping clocks and under-report ~10 %
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--drop", type=int, default=1)
    ap.add_argument("--stats", type=int, default=0)
    ap.add_argument("layers", nargs="*", default=DEFAULT)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    for spec in args.layers:
        name, cin, cout, edge = spec.split(":")
        cin, cout, edge = int(cin), int(cout), int(edge)
        N = args.n
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn((N, edge, edge, edge, cin), generator=g).to(dev)
        w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * (1.0 / (27 * cin)) ** 0.5).to(dev)
        b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
        wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev)
        _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(w), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "pack")
        out = torch.empty((N, edge, edge, edge, cout), dtype=torch.float32, device=dev)
        a = _lib.ConvArgs()
        a.w_family = lib.vx_conv3d_k3_family(cin, cout)
        a.in_ = x.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = b.data_ptr(); a.out = out.data_ptr()
        a.in_pitch, a.out_pitch, a.out_coff = cin, cout, 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = N, edge, edge, edge, cin, cout
        a.act = args.act
        a.drop_mode = args.drop
        a.drop_seed, a.drop_layer = 1, 2
        st = None
        if args.stats:
            st = torch.zeros((N, lib.vx_conv3d_k3_tiles(edge, edge, edge), cout, 2), dtype=torch.float32, device=dev)
            a.stats_partial = st.data_ptr()
        s = _lib.stream_ptr()
        for _ in range(max(2, args.reps // 4)):
            _lib.check(lib.vx_conv3d_k3(C.byref(a), s), "conv")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            _lib.check(lib.vx_conv3d_k3(C.byref(a), s), "conv")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        fl = 2.0 * 27 * cin * cout * edge ** 3 * N
        # correctness on sample 0, no dropout
        a.drop_mode = 0
        _lib.check(lib.vx_conv3d_k3(C.byref(a), s), "conv")
        torch.cuda.synchronize()
        ref = F.conv3d(x[:1].permute(0, 4, 1, 2, 3).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
        if args.act == 1:
            ref = F.leaky_relu(ref, 0.01)
        err = (out[:1].permute(0, 4, 1, 2, 3).double().cpu() - ref).abs().max().item()
        print(f"{name:12s} {cin:3d}->{cout:3d} @{edge:2d}^3 N={N}: {ms:8.4f} ms  {fl / ms / 1e9:7.2f} TFLOP/s  max|d|={err:.1e}",
              flush=True)


if __name__ == "__main__":
    main()
