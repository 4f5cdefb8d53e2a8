#!/usr/bin/env python3
"""Micro-benchmark of the norm/LeakyReLU/dropout(/pool) kernel at the two 64^3 configurations of the network:
  fan-out  : contr_1_1 -> 8 channels, one source volume feeds T = 10 samples, plain output      (write-only stream)
  pool+skip: contr_1_2 -> skip half of the x-blocked concat buffer + 2x2x2 max-pool              (read + write)
Prints ms per launch and achieved TB/s (algorithmic bytes)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from values_amd import _lib  # noqa: E402


def run(label, a, rep, reps=200):
    lib = _lib.load()
    s = _lib.stream_ptr()
    for _ in range(reps // 4):
        _lib.check(lib.vx_norm_act_drop_pool_bcast(C.byref(a), rep, s), "norm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.vx_norm_act_drop_pool_bcast(C.byref(a), rep, s), "norm")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    V, T, S, Cc = 16, 10, 64, 8
    N = V * T
    vox = S ** 3
    for drop in (_lib.VX_DROP_HASH, _lib.VX_DROP_NONE):
        src = torch.randn((V, S, S, S, Cc), device=dev)
        out = torch.empty((N, S, S, S, Cc), device=dev)
        mean = torch.zeros((V, Cc), device=dev); rstd = torch.ones((V, Cc), device=dev)
        a = _lib.NormArgs()
        a.x = src.data_ptr(); a.x_pitch = Cc; a.mean = mean.data_ptr(); a.rstd = rstd.data_ptr()
        a.out = out.data_ptr(); a.out_pitch = Cc; a.out_coff = 0
        a.N, a.D, a.H, a.W, a.C = N, S, S, S, Cc
        a.act, a.drop_mode, a.drop_seed, a.drop_layer = _lib.VX_ACT_LRELU, drop, 1, 0
        ms = run("fanout", a, T)
        print(f"fan-out   drop={drop}: {ms:.4f} ms  {N * vox * Cc * 4 / ms / 1e9:.2f} TB/s written")
        raw = torch.randn((N, S, S, S, Cc), device=dev)
        cat = torch.empty((N, S, S, S // 4, 2, 4, Cc), device=dev)
        pool = torch.empty((N, S // 2, S // 2, S // 2, Cc), device=dev)
        meanN = torch.zeros((N, Cc), device=dev); rstdN = torch.ones((N, Cc), device=dev)
        b = _lib.NormArgs()
        b.x = raw.data_ptr(); b.x_pitch = Cc; b.mean = meanN.data_ptr(); b.rstd = rstdN.data_ptr()
        b.out = cat.data_ptr(); b.out_xblk = 4; b.out_half = 1
        b.pool_out = pool.data_ptr(); b.pool_pitch = Cc
        b.N, b.D, b.H, b.W, b.C = N, S, S, S, Cc
        b.act, b.drop_mode, b.drop_seed, b.drop_layer = _lib.VX_ACT_LRELU, drop, 1, 1
        ms = run("pool", b, 1)
        print(f"pool+skip drop={drop}: {ms:.4f} ms  {N * vox * Cc * 4 * 2.125 / ms / 1e9:.2f} TB/s moved")


if __name__ == "__main__":
    main()
