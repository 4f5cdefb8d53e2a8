#!/usr/bin/env python3
"""Diagnostic: where a wave of the split-fp16 x-pair conv kernel spends its cycles, per item, for the early (0..3) and
late (4..7) waves (s_memtime stamps, VX_CONV_STAMPS build; nothing of this is in the product library).
Build:  mkdir -p /tmp/stamps && cp values_amd/csrc/*.hip values_amd/csrc/*.h values_amd/csrc/*.cpp values_amd/csrc/Makefile /tmp/stamps ...
        (tools/build_stamps.sh does it)
Usage:  python tools/stamp_s16.py [cin:cout:edge:act:drop:head[:up[:pre[:pool[:compose[:upsplit]]]]] ...]   with VX_S16_DBG / VX_XP_ABL (diagnostic build only) for phase ablation
        (up = 1: the fused up-convolution of conv3d_xp8.hip, the skip half from a plain 8-channel tensor)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd import _lib
if not os.environ.get("STAMP_PRODUCT"):      # STAMP_PRODUCT=1: time the product library (no stamps; for rocprofv3 --pmc)
    _lib.LIB_PATH = os.path.join(ROOT, "values_amd", "libvalues_amd_stamps.so")
lib = _lib.load()
dev = torch.device("cuda", 0)
N = int(os.environ.get("STAMP_N", "160"))
for spec in sys.argv[1:] or ["8:8:64:0:0:0", "8:8:64:1:1:1", "16:8:64:1:1:0"]:
    f = list(map(int, spec.split(":")))
    cin, cout, edge, act, drop, head = f[:6]
    up = f[6] if len(f) > 6 else 0
    pre = f[7] if len(f) > 7 else 0        # 1: normalise-on-load prologue (hash dropout of the producing block), in_repeat = 10 for Cin = 8
    pool = f[8] if len(f) > 8 else 0       # 1: pooled output (contr_1_2's epilogue)
    x = torch.randn((N, edge, edge, edge, 8 if up else cin), device=dev)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05; b = torch.zeros(cout, device=dev)
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev)
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(w), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "pack")
    out = torch.empty((N, edge, edge, edge, cout), device=dev)
    dbg = torch.zeros((4096, 16, 8), dtype=torch.int64, device=dev)   # conv3d_xp8w.hip: up to 16 waves
    os.environ["VX_CONV_DBG_PTR"] = str(dbg.data_ptr())
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, cout)
    a.in_ = x.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = b.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = (8 if up else cin), cout, 0
    if up:
        coarse = torch.randn((N, edge // 2, edge // 2, edge // 2, 16), device=dev)
        uw = torch.randn((16, 8, 2, 2, 2), device=dev) * 0.2; ub = torch.zeros(8, device=dev)
        uwp = torch.empty(lib.vx_convT_k2s2_packed_floats(16, 8), dtype=torch.float32, device=dev)
        _lib.check(lib.vx_pack_convT_k2s2(_lib.ptr(uw), _lib.ptr(uwp), 16, 8, _lib.stream_ptr()), "packT")
        a.up_in, a.up_w, a.up_b, a.up_pitch = coarse.data_ptr(), uwp.data_ptr(), ub.data_ptr(), 16
        if len(f) > 9 and f[9]:      # field 10: the up-convolution composed into the weights (round 4, vx_conv3d_args.up_fused)
            uf = torch.empty(lib.vx_conv3d_upfused_packed_floats(), dtype=torch.float32, device=dev)
            _lib.check(lib.vx_pack_conv3d_upfused(_lib.ptr(w), _lib.ptr(b), _lib.ptr(uw), _lib.ptr(ub), _lib.ptr(uf), _lib.stream_ptr()), "packU")
            a.up_fused = uf.data_ptr()
        if len(f) > 10 and f[10]:    # field 11: the coarse tensor arrives as fp16 pairs (the bits do not matter for timing)
            a.up_split = 1
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = N, edge, edge, edge, cin, cout
    if pre:
        rep = 10 if (cin == 8 and N % 10 == 0) else 1
        if rep > 1:
            x = torch.randn((N // rep, edge, edge, edge, cin), device=dev); a.in_ = x.data_ptr()
        mean = torch.zeros((N // rep, cin), device=dev); rstd = torch.ones((N // rep, cin), device=dev)   # (one row of Cin statistics per source sample)
        a.in_mean, a.in_rstd, a.in_drop_mode, a.in_drop_seed, a.in_drop_layer, a.in_repeat = mean.data_ptr(), rstd.data_ptr(), 1, 7, 1, rep
        if pre == 2:      # the once-per-volume pre-split tensor (vx_prenorm_split), as the MC-dropout forward feeds contr_1_2
            _lib.check(lib.vx_prenorm_split(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), N // rep, edge ** 3, 2.0, _lib.stream_ptr()), "presplit")
            a.in_split = 1
    a.act, a.drop_mode, a.drop_seed, a.drop_layer = act, drop, 1, 2
    if pool:
        praw = torch.empty((N, edge // 2, edge // 2, edge // 2, 8), device=dev)
        pfl = torch.empty((N, edge // 2, edge // 2, edge // 2, 2), dtype=torch.int32, device=dev)
        a.pool_out, a.pool_flags = praw.data_ptr(), pfl.data_ptr()
        a.drop_mode, a.drop_seed, a.drop_layer = 1, 1, 1
    st = None
    if not act:
        st = torch.zeros((N, lib.vx_conv3d_k3_tiles(edge, edge, edge), cout, 2), device=dev)
        a.stats_partial = st.data_ptr()
    if head:
        hw = torch.randn((2, cout), device=dev); hb = torch.zeros(2, device=dev)
        ho = torch.empty((N, 2, edge, edge, edge), device=dev)
        a.out = None
        a.head_out, a.head_w, a.head_b, a.head_C = ho.data_ptr(), hw.data_ptr(), hb.data_ptr(), 2
    reps = int(os.environ.get("STAMP_REPS", "200"))
    for _ in range(reps // 2):
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "conv")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "conv")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    d = dbg.cpu().double()
    names = ["barrier wait", "multiply", "epilogue", "load wait", "convert+lds", "issue loads"]
    print(f"{spec}: {ms:.4f} ms/launch at {N} samples (stamped build) abl={os.environ.get('VX_XP_ABL', '0')} {lib.vx_last_kernel_name().decode()}")
    if os.environ.get("STAMP_TERSE"):
        continue
    kname = lib.vx_last_kernel_name().decode()
    print("  kernel", kname)
    if "xp8w" not in kname and "zc16" not in kname and "deep" not in kname:
        d = d.reshape(-1, 8, 8)[: 4096]      # 8-wave kernels index [workgroup][8 waves]
    groups = [("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))]
    if "xp8w" in kname or "zc16" in kname or "deep" in kname:
        groups.append(("producers", slice(8, 16)))
    for half, sl in groups:
        dd = d[:, sl]
        used = dd[:, :, 6] > 0
        it = dd[:, :, 6][used].mean().item()
        tot = sum(dd[:, :, i][used].mean().item() for i in range(6))
        print(f"  {half}: {it:.1f} items/wave, {tot/it:.0f} ticks/item (s_memtime = 100 MHz: x ~21 for shader cycles)")
        for i, n in enumerate(names):
            v = dd[:, :, i][used].mean().item()
            print(f"     {n:12s} {v/it:8.1f} ticks/item  {100*v/tot:5.1f} %")
