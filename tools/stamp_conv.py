#!/usr/bin/env python3
"""Diagnostic: phase shares of the conv kernel's work-item loop (s_memtime stamps, VX_CONV_STAMPS build).
Build:  make -C values_amd/csrc OUT=../libvalues_amd_stamps.so EXTRA=-DVX_CONV_STAMPS  (separate objects dir)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "values_amd", "libvalues_amd_stamps.so")
lib = _lib.load()
dev = torch.device("cuda", 0)
for spec in sys.argv[1:] or ["16:8:64", "16:16:32", "8:8:64", "64:32:16"]:
    cin, cout, edge = map(int, spec.split(":")); N = 80
    x = torch.randn((N, edge, edge, edge, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05; b = torch.zeros(cout, device=dev)
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev)
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(w), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "pack")
    out = torch.empty((N, edge, edge, edge, cout), device=dev)
    dbg = torch.zeros((4096, 8, 8), dtype=torch.int64, device=dev)
    os.environ["VX_CONV_DBG_PTR"] = str(dbg.data_ptr())
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, cout)
    a.in_ = x.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = b.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = cin, cout, 0
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = N, edge, edge, edge, cin, cout
    a.act, a.drop_mode, a.drop_seed, a.drop_layer = 1, 1, 1, 2
    reps = int(os.environ.get("STAMP_REPS", "300"))   # long enough for steady clocks
    for _ in range(reps // 3):
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "conv")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "conv")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    d = dbg.cpu().double()
    used = d[:, :, 6] > 0
    it = d[:, :, 6][used].mean().item()
    names = ["barrier1", "commit", "barrier2", "prefetch", "compute", "epilogue"]
    tot = sum(d[:, :, i][used].mean().item() for i in range(6))
    print(f"{cin}->{cout}@{edge}: {ms:.4f} ms/launch, {int(used.sum())} waves, {it:.1f} items/wave, {tot/it:.0f} ticks/item; "
          f"loop ticks / launch time = {tot / (ms * 1e3):.0f} ticks/us (s_memtime rate if the loop spans the launch)")
    for i, n in enumerate(names):
        v = d[:, :, i][used].mean().item()
        print(f"   {n:10s} {v/it:9.1f} per item  {100*v/tot:5.1f} %")
