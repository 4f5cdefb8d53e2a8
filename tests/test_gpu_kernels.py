"""GPU: every kernel of libvalues_amd.so, called through the C ABI, against the oracle
(oracle/*.py: float64 CPU restatement of the reference) on seeded inputs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import load_npz
from values_amd import _lib
from tests.formula import formula_tensor

pytestmark = pytest.mark.gpu

KEYS = ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty")


def dev():
    return torch.device("cuda", 0)


def cl(x):  # NCDHW -> channels-last contiguous
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def to_xblk(up, skip, xb):
    """(N,C,D,H,W) up & skip -> concat buffer [N][D][H][W/xb][2][xb][C] (flat last dims)"""
    n, c, d, h, w = up.shape
    u = cl(up).reshape(n, d, h, w // xb, 1, xb, c)
    k = cl(skip).reshape(n, d, h, w // xb, 1, xb, c)
    return torch.cat([u, k], 4).contiguous()


def from_xblk(buf, half, n, c, d, h, w, xb):
    return ncdhw(buf.reshape(n, d, h, w // xb, 2, xb, c)[:, :, :, :, half].reshape(n, d, h, w, c))


@pytest.fixture(params=["split16", "fp32"])
def conv_mode(request, vxcfg):
    """the default split-fp16 schedule (conv3d_s16.hip) and the native-fp32 kernels (VX_CONV_FP32=1)"""
    if request.param == "fp32":
        vxcfg.setenv("VX_CONV_FP32", "1")
    else:
        vxcfg.delenv("VX_CONV_FP32", raising=False)
    return request.param


def run_conv(x, w, b, act=0, drop_mode=0, mask=None, seed=0, layer=0, stats=False, in_pitch=None, out_pitch=None,
             out_coff=0, xblk=0):
    """x (N,Cin,D,H,W) f32 cpu, w torch layout -> (out NCDHW cpu, stats or None)"""
    lib = _lib.load()
    N, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    if xblk:
        xd = to_xblk(x[:, :Cin // 2].float(), x[:, Cin // 2:].float(), xblk).to(dev())
    else:
        xd = cl(x.float()).to(dev())
    if in_pitch and in_pitch > Cin:
        pad = torch.full((N, D, H, W, in_pitch - Cin), 7.0, device=dev())  # garbage channels must be ignored
        xd = torch.cat([xd, pad], -1).contiguous()
    in_pitch = in_pitch or Cin
    out_pitch = out_pitch or Cout
    wd = w.float().contiguous().to(dev())
    bd = b.float().contiguous().to(dev())
    n = lib.vx_conv3d_k3_packed_floats(Cin, Cout)
    wp = torch.empty(n, dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wd), _lib.ptr(wp), Cin, Cout, _lib.stream_ptr()), "pack")
    out = torch.full((N, D, H, W, out_pitch), -77.0, dtype=torch.float32, device=dev())
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(Cin, Cout)
    a.in_ = xd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = in_pitch, out_pitch, out_coff
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = N, D, H, W, Cin, Cout
    a.act, a.drop_mode, a.drop_seed, a.drop_layer = act, drop_mode, seed, layer
    a.in_xblk = xblk
    md = None
    if mask is not None:
        md = cl(mask).to(torch.uint8).to(dev())
        a.drop_mask = md.data_ptr()
    st = None
    if stats:
        nt = lib.vx_conv3d_k3_tiles_for(D, H, W, Cout)
        assert nt <= lib.vx_conv3d_k3_tiles(D, H, W)
        st = torch.zeros((N, nt, Cout, 2), dtype=torch.float32, device=dev())
        a.stats_partial = st.data_ptr()
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    torch.cuda.synchronize()
    res = ncdhw(out[..., out_coff:out_coff + Cout]).cpu()
    untouched = out[..., :out_coff].cpu(), out[..., out_coff + Cout:].cpu()
    return res, (st.cpu() if st is not None else None), untouched


@pytest.mark.parametrize("cin,cout,shape", [
    (8, 8, (1, 16, 16, 16)), (16, 8, (2, 8, 16, 32)), (8, 16, (1, 16, 16, 16)), (16, 16, (1, 8, 8, 16)),
    (16, 32, (1, 8, 8, 8)), (32, 32, (2, 8, 8, 8)), (32, 64, (1, 4, 4, 4)), (64, 64, (1, 4, 4, 4)),
    (64, 128, (2, 2, 2, 2)), (128, 128, (1, 2, 2, 2)), (128, 64, (1, 4, 4, 4)), (64, 32, (1, 8, 8, 8)),
    (24, 8, (1, 4, 8, 16)),   # Cin multiple of 8 only -> CB=8 path
    (8, 8, (1, 6, 10, 20)),   # ragged: not a multiple of the workgroup tile
    (16, 16, (1, 3, 5, 9)),   # ragged small
    (8, 8, (2, 8, 8, 64)), (16, 8, (1, 4, 4, 32)), (16, 8, (1, 5, 7, 38)),  # x-pair packing, full and ragged tiles
    (8, 8, (1, 3, 3, 7)), (16, 8, (2, 2, 2, 2)),                            # x-pair, odd width / tiny
    (8, 8, (1, 5, 37, 70)), (16, 8, (1, 4, 33, 40)), (16, 16, (1, 6, 35, 18)), (32, 32, (1, 4, 40, 16)),  # H >= 32: 16x8x4 tiles, ragged
])
def test_conv3d_k3_matches_oracle(cin, cout, shape, conv_mode):
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 101))
    wt = torch.from_numpy(formula_tensor((cout, cin, 3, 3, 3), 102, scale=(1.0 / (27 * cin)) ** 0.5))
    b = torch.from_numpy(formula_tensor((cout,), 103, scale=0.2))
    ref = F.conv3d(x.float().double(), wt.float().double(), b.float().double(), padding=1)
    got, st, _ = run_conv(x, wt, b, stats=True)
    err = (got.double() - ref).abs().max().item()
    assert err < 2e-5, err
    # statistics partials sum to the per-(n,c) sum / sumsq of the output
    s = st.double().sum(1)
    np.testing.assert_allclose(s[..., 0].numpy(), ref.sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[..., 1].numpy(), (ref * ref).sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("cin,shape,ncls,xb", [(8, (3, 8, 8, 32), 2, 0), (16, (2, 5, 7, 38), 4, 0), (16, (2, 4, 8, 16), 2, 4),
                                                (8, (1, 3, 3, 7), 3, 0)])
def test_conv3d_k3_fused_head_matches_separate_kernels(cin, shape, ncls, xb, conv_mode):
    """expand_1_2 + final (unet3D_module.py:365) in one launch: equal to vx_conv3d_k3 followed by
    vx_conv1x1_ncdhw (bit-identical on the 4x4x1 kernel), including the slot scatter and the TTA un-flip (test_3D.py:445-447)"""
    lib = _lib.load()
    _fused_head_case(lib, cin, shape, ncls, xb, conv_mode)


def _fused_head_case(lib, cin, shape, ncls, xb, conv_mode):
    assert lib.vx_conv3d_k3_head_fusable(cin, 8)
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 161))
    wt = torch.from_numpy(formula_tensor((8, cin, 3, 3, 3), 162, scale=(1.0 / (27 * cin)) ** 0.5))
    b = torch.from_numpy(formula_tensor((8,), 163, scale=0.2))
    hw = torch.from_numpy(formula_tensor((ncls, 8, 1, 1, 1), 164, scale=0.3)).float().contiguous().to(dev())
    hb = torch.from_numpy(formula_tensor((ncls,), 165, scale=0.2)).float().to(dev())
    mask = torch.from_numpy(formula_tensor((n, 8, d, h, w), 166)) > 0
    if xb:
        xd = to_xblk(x[:, :cin // 2].float(), x[:, cin // 2:].float(), xb).to(dev())
    else:
        xd = cl(x.float()).to(dev())
    wd, bd = wt.float().contiguous().to(dev()), b.float().to(dev())
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, 8), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wd), _lib.ptr(wp), cin, 8, _lib.stream_ptr()), "pack")
    md = cl(mask).to(torch.uint8).to(dev())
    dst = torch.tensor([(n - 1 - i) * 2 for i in range(n)], dtype=torch.int32, device=dev())   # scattered slots
    flip = torch.tensor([(3 * i + 5) % 8 for i in range(n)], dtype=torch.int32, device=dev())
    slots = 2 * n
    feat = torch.empty((n, d, h, w, 8), dtype=torch.float32, device=dev())
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, 8)
    a.in_ = xd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = feat.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = cin, 8, 0
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, 8
    a.act, a.drop_mode, a.drop_mask, a.in_xblk = _lib.VX_ACT_LRELU, _lib.VX_DROP_MASK, md.data_ptr(), xb
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "conv")
    ref = torch.full((slots, ncls, d, h, w), -5.0, dtype=torch.float32, device=dev())
    _lib.check(lib.vx_conv1x1_ncdhw(feat.data_ptr(), 8, hw.data_ptr(), hb.data_ptr(), ref.data_ptr(), n, d, h, w, 8, ncls,
                                    dst.data_ptr(), flip.data_ptr(), _lib.stream_ptr()), "conv1x1")
    got = torch.full((slots, ncls, d, h, w), -5.0, dtype=torch.float32, device=dev())
    a.out = None
    a.head_out, a.head_w, a.head_b, a.head_C = got.data_ptr(), hw.data_ptr(), hb.data_ptr(), ncls
    a.head_dst, a.head_flip = dst.data_ptr(), flip.data_ptr()
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "conv+head")
    torch.cuda.synchronize()
    if conv_mode == "fp32":
        assert torch.equal(got, ref)     # same fmaf chain as conv1x1.hip
    else:
        assert (got - ref).abs().max().item() < 1e-5   # two 4-channel partial chains + one add
    # and against the float64 composition of the two reference modules
    r64 = F.leaky_relu(F.conv3d(x.float().double(), wt.float().double(), b.float().double(), padding=1), 0.01) * mask * 2.0
    r64 = F.conv3d(r64, hw.cpu().double(), hb.cpu().double())
    for i in range(n):
        f = int(flip[i])
        dims = [1 + k for k in range(3) if f >> k & 1]
        want = torch.flip(r64[i], dims) if dims else r64[i]
        assert (got[int(dst[i])].cpu().double() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("cin,cout,shape,xb", [
    (8, 8, (3, 8, 32, 64), 0), (8, 8, (2, 5, 37, 70), 0),        # one chunk per tile: double-buffered, staggered (DB = 2)
    (16, 8, (2, 8, 32, 64), 4), (16, 8, (3, 4, 33, 40), 0),       # two chunks per tile (DB = 3), x-blocked concat input
    (16, 16, (2, 6, 35, 34), 0), (8, 16, (2, 8, 32, 32), 0),      # plain large-tile instances (EPI only)
])
@pytest.mark.parametrize("mode", ["plain_stats", "lrelu_hash", "lrelu_hash_head"])
def test_conv3d_k3_specialised_instances_equal_generic(cin, cout, shape, xb, mode, vxcfg):
    """The large-tile instances of conv3d_s16.hip with compile-time epilogues (EPI 0 / 1 / 2) and the double-buffered,
    staggered item loop (DB 2 / 3) must give the bits of the generic kernel (two barriers per item, run-time
    epilogue: vx_config.s16_generic) -- output, statistics partials and fused head, hash dropout included."""
    if mode == "lrelu_hash_head" and cout != 8:
        pytest.skip("the head rides in x-pair epilogues only")
    lib = _lib.load()
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 171))
    wt = torch.from_numpy(formula_tensor((cout, cin, 3, 3, 3), 172, scale=(1.0 / (27 * cin)) ** 0.5))
    b = torch.from_numpy(formula_tensor((cout,), 173, scale=0.2))
    xd = (to_xblk(x[:, :cin // 2].float(), x[:, cin // 2:].float(), xb) if xb else cl(x.float())).to(dev())
    wd, bd = wt.float().contiguous().to(dev()), b.float().to(dev())
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wd), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "pack")
    ncls = 2
    hw = torch.from_numpy(formula_tensor((ncls, 8, 1, 1, 1), 174, scale=0.3)).float().contiguous().to(dev())
    hb = torch.from_numpy(formula_tensor((ncls,), 175, scale=0.2)).float().to(dev())
    flip = torch.tensor([(3 * i + 1) % 8 for i in range(n)], dtype=torch.int32, device=dev())
    nt = lib.vx_conv3d_k3_tiles_for(d, h, w, cout)

    def run():
        out = torch.full((n, d, h, w, cout), -3.0, dtype=torch.float32, device=dev())
        st = torch.zeros((n, nt, cout, 2), dtype=torch.float32, device=dev())
        head = torch.full((n, ncls, d, h, w), -5.0, dtype=torch.float32, device=dev())
        a = _lib.ConvArgs()
        a.w_family = lib.vx_conv3d_k3_family(cin, cout)
        a.in_ = xd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
        a.in_pitch, a.out_pitch, a.out_coff = cin, cout, 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, cout
        a.in_xblk = xb
        if mode == "plain_stats":
            a.stats_partial = st.data_ptr()
        else:
            a.act, a.drop_mode, a.drop_seed, a.drop_layer = _lib.VX_ACT_LRELU, _lib.VX_DROP_HASH, 77, 5
        if mode == "lrelu_hash_head":
            a.out = None
            a.head_out, a.head_w, a.head_b, a.head_C = head.data_ptr(), hw.data_ptr(), hb.data_ptr(), ncls
            a.head_flip = flip.data_ptr()
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "conv")
        torch.cuda.synchronize()
        return out, st, head

    vxcfg.delenv("VX_CONV_FP32", raising=False)
    got = run()
    vxcfg.set(s16_generic=1)
    ref = run()
    for g, r in zip(got, ref):
        assert torch.equal(g, r)
    if mode == "plain_stats":   # and the statistics are the sums of what was stored
        tot = got[1].double().sum(1)
        o = got[0].double()
        assert torch.allclose(tot[..., 0], o.sum((1, 2, 3)), rtol=1e-5, atol=1e-3)
        assert torch.allclose(tot[..., 1], (o * o).sum((1, 2, 3)), rtol=1e-5, atol=1e-3)


def test_conv3d_k3_epilogue_act_mask_pitch(conv_mode):
    x = torch.from_numpy(formula_tensor((1, 16, 8, 8, 16), 111))
    wt = torch.from_numpy(formula_tensor((8, 16, 3, 3, 3), 112, scale=0.05))
    b = torch.from_numpy(formula_tensor((8,), 113, scale=0.2))
    mask = torch.from_numpy(formula_tensor((1, 8, 8, 8, 16), 114)) > 0
    ref = F.leaky_relu(F.conv3d(x.float().double(), wt.float().double(), b.float().double(), padding=1), 0.01)
    ref = ref * mask * 2.0
    got, _, (lo, hi) = run_conv(x, wt, b, act=_lib.VX_ACT_LRELU, drop_mode=_lib.VX_DROP_MASK, mask=mask, in_pitch=24,
                                out_pitch=24, out_coff=8)
    assert (got.double() - ref).abs().max().item() < 2e-5
    assert (lo == -77.0).all() and (hi == -77.0).all()  # neighbours in the concat buffer untouched
    ref_relu = F.relu(F.conv3d(x.float().double(), wt.float().double(), b.float().double(), padding=1))
    got, _, _ = run_conv(x, wt, b, act=_lib.VX_ACT_RELU)
    assert (got.double() - ref_relu).abs().max().item() < 2e-5


@pytest.mark.parametrize("c,cout,shape,xb", [(8, 8, (2, 8, 8, 32), 4), (16, 16, (1, 4, 8, 16), 4), (32, 32, (1, 4, 4, 8), 4),
                                             (64, 64, (1, 6, 6, 6), 2), (8, 8, (1, 3, 5, 7), 1), (16, 16, (1, 4, 4, 12), 4),
                                             (8, 8, (1, 4, 34, 36), 4), (16, 16, (1, 3, 33, 20), 4)])
def test_conv3d_k3_reads_xblocked_concat(c, cout, shape, xb, conv_mode):
    """decoder conv on cat([up, skip]) (unet3D_module.py:332-334) reading the two halves from the x-blocked buffer"""
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, 2 * c, d, h, w), 115))
    wt = torch.from_numpy(formula_tensor((cout, 2 * c, 3, 3, 3), 116, scale=(1.0 / (27 * 2 * c)) ** 0.5))
    b = torch.from_numpy(formula_tensor((cout,), 117, scale=0.2))
    ref = F.conv3d(x.float().double(), wt.float().double(), b.float().double(), padding=1)
    got, _, _ = run_conv(x, wt, b, xblk=xb)
    assert (got.double() - ref).abs().max().item() < 2e-5


def test_conv3d_k3_hash_dropout_statistics():
    x = torch.from_numpy(formula_tensor((2, 8, 16, 16, 16), 121))
    wt = torch.from_numpy(formula_tensor((8, 8, 3, 3, 3), 122, scale=0.07))
    b = torch.ones(8, dtype=torch.float64) * 3.0  # keep outputs away from 0
    base, _, _ = run_conv(x, wt, b)
    d1, _, _ = run_conv(x, wt, b, drop_mode=_lib.VX_DROP_HASH, seed=5, layer=3)
    d1b, _, _ = run_conv(x, wt, b, drop_mode=_lib.VX_DROP_HASH, seed=5, layer=3)
    d2, _, _ = run_conv(x, wt, b, drop_mode=_lib.VX_DROP_HASH, seed=6, layer=3)
    d3, _, _ = run_conv(x, wt, b, drop_mode=_lib.VX_DROP_HASH, seed=5, layer=4)
    assert torch.equal(d1, d1b)  # deterministic
    keep = d1 != 0
    assert torch.equal(d1[keep], (2 * base)[keep])  # kept values are exactly 2x
    rate = keep.float().mean().item()
    assert abs(rate - 0.5) < 0.01, rate
    # per-channel and per-sample rates, and independence across seed / layer / sample
    assert (keep.float().mean((0, 2, 3, 4)) - 0.5).abs().max() < 0.02
    for other in (d2, d3):
        agree = ((other != 0) == keep).float().mean().item()
        assert abs(agree - 0.5) < 0.01, agree
    agree = (keep[0] == keep[1]).float().mean().item()
    assert abs(agree - 0.5) < 0.01, agree
    # neighbouring voxels uncorrelated
    a, bb = keep[..., :-1].float() - 0.5, keep[..., 1:].float() - 0.5
    assert abs((a * bb).mean().item()) < 0.005


@pytest.mark.parametrize("cout,shape,flipcode", [(8, (2, 16, 16, 32), 0), (8, (1, 6, 10, 40), 0), (16, (1, 4, 8, 16), 5),
                                                 (8, (3, 8, 8, 8), 7), (8, (1, 16, 16, 16), 2)])
def test_conv3d_c1_matches_oracle(cout, shape, flipcode):
    lib = _lib.load()
    v, d, h, w = shape
    repeat = 2
    x = torch.from_numpy(formula_tensor((v, 1, d, h, w), 131))
    wt = torch.from_numpy(formula_tensor((cout, 1, 3, 3, 3), 132, scale=0.2))
    b = torch.from_numpy(formula_tensor((cout,), 133, scale=0.2))
    N = v * repeat
    xd, wd, bd = x.float().to(dev()), wt.float().to(dev()), b.float().to(dev())
    out = torch.empty((N, d, h, w, cout), dtype=torch.float32, device=dev())
    nt = lib.vx_conv3d_k3_c1_tiles(d, h, w)
    st = torch.zeros((N, nt, cout, 2), dtype=torch.float32, device=dev())
    flips = torch.tensor([flipcode if n % 2 else 0 for n in range(N)], dtype=torch.int32, device=dev())
    _lib.check(lib.vx_conv3d_k3_c1(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(out), cout, N, d, h, w, cout, repeat,
                                   None, _lib.ptr(flips), _lib.ptr(st), _lib.stream_ptr()), "c1")
    torch.cuda.synchronize()
    got = ncdhw(out).cpu().double()
    for n in range(N):
        xi = x[n // repeat:n // repeat + 1].float().double()
        code = flipcode if n % 2 else 0
        dims = [2 + k for k in range(3) if code >> k & 1]
        if dims:
            xi = torch.flip(xi, dims)
        ref = F.conv3d(xi, wt.float().double(), b.float().double(), padding=1)
        assert (got[n:n + 1] - ref).abs().max().item() < 1e-5
        np.testing.assert_allclose(st[n].double().sum(0)[:, 0].cpu().numpy(), ref.sum((0, 2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("cin,cout,shape", [(16, 8, (2, 4, 8, 16)), (128, 64, (1, 2, 2, 2)), (64, 32, (1, 4, 4, 4)),
                                            (32, 16, (1, 3, 5, 7)), (16, 16, (1, 2, 3, 5)), (32, 24, (1, 2, 2, 3)),
                                            (48, 8, (1, 2, 2, 2)), (64, 64, (2, 2, 3, 2))])
def test_convT_matches_oracle(cin, cout, shape):
    lib = _lib.load()
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 141))
    wt = torch.from_numpy(formula_tensor((cin, cout, 2, 2, 2), 142, scale=(1.0 / cin) ** 0.5))
    b = torch.from_numpy(formula_tensor((cout,), 143, scale=0.2))
    mask = torch.from_numpy(formula_tensor((n, cout, 2 * d, 2 * h, 2 * w), 144)) > 0
    xd, wd, bd = cl(x.float()).to(dev()), wt.float().contiguous().to(dev()), b.float().to(dev())
    wp = torch.empty(lib.vx_convT_k2s2_packed_floats(cin, cout), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_convT_k2s2(_lib.ptr(wd), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "packT")
    for act, use_mask in ((0, False), (_lib.VX_ACT_RELU, True)):
        pitch = 2 * cout
        out = torch.full((n, 2 * d, 2 * h, 2 * w, pitch), -77.0, dtype=torch.float32, device=dev())
        a = _lib.ConvTArgs()
        a.in_ = xd.data_ptr(); a.in_pitch = cin; a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr()
        a.out = out.data_ptr(); a.out_pitch = pitch; a.out_coff = 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, cout
        a.act = act
        md = cl(mask).to(torch.uint8).to(dev())
        if use_mask:
            a.drop_mode = _lib.VX_DROP_MASK
            a.drop_mask = md.data_ptr()
        _lib.check(lib.vx_convT_k2s2(C.byref(a), _lib.stream_ptr()), "convT")
        torch.cuda.synchronize()
        ref = F.conv_transpose3d(x.float().double(), wt.float().double(), b.float().double(), stride=2)
        if act:
            ref = F.relu(ref)
        if use_mask:
            ref = ref * mask * 2.0
        got = ncdhw(out[..., :cout]).cpu().double()
        assert (got - ref).abs().max().item() < 1e-5
        assert (out[..., cout:] == -77.0).all()
        # the same into the up half of an x-blocked concat buffer
        xb = 4 if (2 * w) % 4 == 0 else 2
        cat = torch.full((n, 2 * d, 2 * h, (2 * w) // xb, 2, xb, cout), -77.0, dtype=torch.float32, device=dev())
        a.out = cat.data_ptr(); a.out_xblk = xb; a.out_half = 0
        _lib.check(lib.vx_convT_k2s2(C.byref(a), _lib.stream_ptr()), "convT")
        torch.cuda.synchronize()
        got = from_xblk(cat, 0, n, cout, 2 * d, 2 * h, 2 * w, xb).cpu().double()
        assert (got - ref).abs().max().item() < 1e-5
        assert (cat[:, :, :, :, 1] == -77.0).all()
        a.out = out.data_ptr(); a.out_xblk = 0


def test_convT_split_fp16_reports_weights_and_outputs_past_the_fp16_range(vxcfg):
    """vx_convT_k2s2 on split-fp16 products (Cin in {64, 128}, round 5): a weight past 65504 is clamped (finite arithmetic) and
    reported through the launch's range word as infinity; an output past 32768 reports its magnitude; in-range launches leave the
    word alone; conv_fp32 = 1 runs the native-fp32 kernel on the same packed weights (include/values_amd.h, vx_convT_args)."""
    lib = _lib.load()
    n, d, h, w, cin, cout = 2, 4, 4, 4, 64, 32
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 151)).float()
    wt = torch.from_numpy(formula_tensor((cin, cout, 2, 2, 2), 152, scale=(1.0 / cin) ** 0.5)).float().contiguous()
    b = torch.zeros((cout,), dtype=torch.float32)

    def run(wts, xin):
        wd = wts.contiguous().to(dev())
        wp = torch.empty(lib.vx_convT_k2s2_packed_floats(cin, cout), dtype=torch.float32, device=dev())
        _lib.check(lib.vx_pack_convT_k2s2(_lib.ptr(wd), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "packT")
        out = torch.empty((n, 2 * d, 2 * h, 2 * w, cout), dtype=torch.float32, device=dev())
        flag = torch.zeros((1,), dtype=torch.int32, device=dev())
        xd = cl(xin).to(dev())
        a = _lib.ConvTArgs()
        a.in_ = xd.data_ptr(); a.in_pitch = cin; a.w_packed = wp.data_ptr(); a.bias = b.to(dev()).data_ptr()
        bias_keep = b.to(dev()); a.bias = bias_keep.data_ptr()
        a.out = out.data_ptr(); a.out_pitch = cout; a.out_coff = 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, cout
        a.range_flag = flag.data_ptr()
        _lib.check(lib.vx_convT_k2s2(C.byref(a), _lib.stream_ptr()), "convT")
        torch.cuda.synchronize()
        return out, flag.view(torch.float32).item(), lib.vx_last_kernel_name().decode()

    out, fl, kn = run(wt, x)
    assert kn.startswith("convT_k2s2_s16_kernel<64,") and fl == 0.0, (kn, fl)
    ref = F.conv_transpose3d(x.double(), wt.double(), None, stride=2)
    assert (ncdhw(out).cpu().double() - ref).abs().max().item() < 1e-5
    big = wt.clone(); big[3, 5, 0, 1, 1] = 1.0e5                   # one weight past the fp16 range
    _, fl, _ = run(big, x)
    assert fl == float("inf"), fl
    _, fl, _ = run(wt, x * 3.0e4)                                   # outputs past 32768, weights and inputs in range
    assert 32768.0 <= fl < float("inf"), fl
    vxcfg.set(conv_fp32=1)
    out32, fl, kn = run(big, x)
    assert kn.startswith("convT_k2s2_mfma_kernel<64,"), kn
    ref = F.conv_transpose3d(x.double(), big.double(), None, stride=2)
    assert ((ncdhw(out32).cpu().double() - ref).abs() / (1.0 + ref.abs())).max().item() < 1e-5


@pytest.mark.parametrize("c,shape,pool", [(8, (2, 8, 8, 16), True), (16, (1, 4, 6, 10), True), (8, (1, 3, 5, 7), False),
                                          (64, (2, 2, 2, 2), True), (256, (1, 2, 4, 6), True), (4, (1, 2, 2, 2), False)])
def test_instnorm_lrelu_drop_pool_matches_oracle(c, shape, pool):
    lib = _lib.load()
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, c, d, h, w), 151, scale=3.0)) + 1.5
    mask = torch.from_numpy(formula_tensor((n, c, d, h, w), 152)) > 0
    xf = x.float()
    # statistics through the real finalize kernel: one "tile" per sample holding exact sums
    part = torch.stack([xf.double().sum((2, 3, 4)), (xf.double() ** 2).sum((2, 3, 4))], -1).float()  # (n,c,2)
    part = part.reshape(n, 1, c, 2).contiguous().to(dev())
    mean = torch.empty((n, c), dtype=torch.float32, device=dev())
    rstd = torch.empty((n, c), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_instnorm_finalize(_lib.ptr(part), n, 1, c, d * h * w, 1e-5, _lib.ptr(mean), _lib.ptr(rstd),
                                        _lib.stream_ptr()), "finalize")
    xd = cl(xf).to(dev())
    pitch = 2 * c
    out = torch.full((n, d, h, w, pitch), -77.0, dtype=torch.float32, device=dev())
    pooled = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=torch.float32, device=dev()) if pool else None
    md = cl(mask).to(torch.uint8).to(dev())
    a = _lib.NormArgs()
    a.x = xd.data_ptr(); a.x_pitch = c; a.mean = mean.data_ptr(); a.rstd = rstd.data_ptr()
    a.out = out.data_ptr(); a.out_pitch = pitch; a.out_coff = c
    if pool:
        a.pool_out = pooled.data_ptr(); a.pool_pitch = c
    a.N, a.D, a.H, a.W, a.C = n, d, h, w, c
    a.act = _lib.VX_ACT_LRELU; a.drop_mode = _lib.VX_DROP_MASK; a.drop_mask = md.data_ptr()
    _lib.check(lib.vx_norm_act_drop_pool(C.byref(a), _lib.stream_ptr()), "norm")
    torch.cuda.synchronize()
    ref = F.leaky_relu(F.instance_norm(xf.double(), eps=1e-5), 0.01) * mask * 2.0
    got = ncdhw(out[..., c:]).cpu().double()
    assert (got - ref).abs().max().item() < 2e-5
    assert (out[..., :c] == -77.0).all()
    if pool:
        refp = F.max_pool3d(ref, 2, 2)
        gotp = ncdhw(pooled).cpu().double()
        assert (gotp - refp).abs().max().item() < 2e-5
    # the same into the skip half of an x-blocked concat buffer, input broadcast over 3 output samples
    xb = 4 if w % 4 == 0 else (2 if w % 2 == 0 else 1)
    rep = 3
    cat = torch.full((n * rep, d, h, w // xb, 2, xb, c), -77.0, dtype=torch.float32, device=dev())
    mrep = mask.repeat_interleave(rep, 0)
    md2 = cl(mrep).to(torch.uint8).to(dev())
    a.out = cat.data_ptr(); a.out_xblk = xb; a.out_half = 1; a.N = n * rep; a.drop_mask = md2.data_ptr()
    if pool:
        pooled2 = torch.empty((n * rep, d // 2, h // 2, w // 2, c), dtype=torch.float32, device=dev())
        a.pool_out = pooled2.data_ptr()
    _lib.check(lib.vx_norm_act_drop_pool_bcast(C.byref(a), rep, _lib.stream_ptr()), "norm bcast")
    torch.cuda.synchronize()
    got = from_xblk(cat, 1, n * rep, c, d, h, w, xb).cpu().double()
    assert (got - ref.repeat_interleave(rep, 0)).abs().max().item() < 2e-5
    assert (cat[:, :, :, :, 0] == -77.0).all()


def test_conv1x1_slots_and_unflip():
    lib = _lib.load()
    n, f, c, d, h, w = 4, 8, 3, 4, 6, 8
    x = torch.from_numpy(formula_tensor((n, f, d, h, w), 161))
    wt = torch.from_numpy(formula_tensor((c, f), 162))
    b = torch.from_numpy(formula_tensor((c,), 163))
    flips = [0, 1, 6, 7]
    dst = [3, 0, 5, 2]
    out = torch.full((6, c, d, h, w), -77.0, dtype=torch.float32, device=dev())
    xd = cl(x.float()).to(dev())
    wd, bd = wt.float().to(dev()), b.float().to(dev())  # keep every device tensor alive across the launch
    dst_d = torch.tensor(dst, dtype=torch.int32, device=dev())
    flip_d = torch.tensor(flips, dtype=torch.int32, device=dev())
    _lib.check(lib.vx_conv1x1_ncdhw(_lib.ptr(xd), f, _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(out), n, d, h, w, f, c,
                                    _lib.ptr(dst_d), _lib.ptr(flip_d), _lib.stream_ptr()), "1x1")
    torch.cuda.synchronize()
    ref = torch.einsum("nfdhw,cf->ncdhw", x.float().double(), wt.float().double()) + b.float().double().view(1, -1, 1, 1, 1)
    for i in range(n):
        dims = [1 + k for k in range(3) if flips[i] >> k & 1]
        r = torch.flip(ref[i], dims) if dims else ref[i]
        assert (out[dst[i]].cpu().double() - r).abs().max().item() < 1e-5
    assert (out[1] == -77.0).all() and (out[4] == -77.0).all()


# ------------------------------------------------------------------------------------------- reduction
def test_unc_reduce_from_logits_with_more_than_eight_classes():
    """uncertainty_maps(from_logits=True) for wide heads (19 Cityscapes classes): planar softmax + probability
    reduction on the device, same maps as the fused logit kernel gives for narrow heads (test_3D.py:486-518)."""
    from values_amd.uncertainty import uncertainty_maps
    for C_, T_ in ((19, 5), (9, 3), (8, 4)):
        x = torch.from_numpy(formula_tensor((2, T_, C_, 5, 7, 3), 300 + C_, scale=3.0)).float()
        m = uncertainty_maps(x.cuda(), from_logits=True, want_sample_argmax=True)
        p = torch.softmax(x.double(), 2)
        mean = p.mean(1)
        pe = -(mean * torch.log(mean)).sum(1)
        ee = -(p * torch.log(p)).sum(2).mean(1)
        assert (m["pred_entropy"].cpu().double() - pe).abs().max().item() < 2e-5
        assert (m["expected_entropy"].cpu().double() - ee).abs().max().item() < 2e-5
        assert (m["mutual_information"].cpu().double() - (pe - ee)).abs().max().item() < 2e-5
        assert (m["mean_softmax"].cpu().double() - mean).abs().max().item() < 1e-6
        assert torch.equal(m["argmax"].cpu().long(), mean.argmax(1))
        assert torch.equal(m["sample_argmax"].cpu().long(), p.argmax(2))
    with pytest.raises(ValueError):
        uncertainty_maps(torch.zeros((1, 2, 9, 4), dtype=torch.float64).cuda(), from_logits=True)


@pytest.mark.parametrize("case", ["hand", "r3d", "r2d", "ex"])
def test_unc_reduce_matches_reference_fixtures(case):
    from values_amd import calculate_uncertainty
    g = load_npz("unc_kat.npz")
    x = torch.from_numpy(g[f"{case}_in"])
    for device in ("cpu", "cuda"):
        r = calculate_uncertainty(x.to(device))
        for k in KEYS:
            assert r[k].dtype == torch.float32 and r[k].device.type == device
            assert tuple(r[k].shape) == g[f"{case}_{k}"].shape
            np.testing.assert_allclose(r[k].cpu().numpy(), g[f"{case}_{k}"], atol=2e-6, rtol=0)
            assert not torch.isnan(r[k]).any()
    r = calculate_uncertainty(x.cuda(), ssn=True)
    np.testing.assert_allclose(r["aleatoric_uncertainty"].cpu().numpy(), g[f"{case}_epistemic_uncertainty"], atol=2e-6)
    np.testing.assert_allclose(r["epistemic_uncertainty"].cpu().numpy(), g[f"{case}_aleatoric_uncertainty"], atol=2e-6)


def test_one_minus_msr_matches_reference_fixture():
    from values_amd import calculate_one_minus_msr
    g = load_npz("unc_kat.npz")
    r = calculate_one_minus_msr(torch.from_numpy(g["msr_in"]).cuda())
    np.testing.assert_array_equal(r["pred_entropy"].cpu().numpy(), g["msr_pred_entropy"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(10, 2, 16, 16, 16), (3, 5, 7, 9), (1, 2, 4, 4, 4), (4, 8, 33)])
def test_unc_reduce_logits_and_probs_vs_oracle(dtype, shape):
    from oracle import uncertainty_oracle as uo
    from values_amd import uncertainty_maps
    logits = formula_tensor(shape, 171, scale=4.0)
    logits = logits.astype(np.float32 if dtype == torch.float32 else np.float64)
    sm = uo.softmax(logits.astype(np.float64), axis=1)
    ref = uo.calculate_uncertainty(sm)
    mean, mean_seg, pred_seg = uo.mean_and_argmax(sm)
    for from_logits, xin in ((True, logits), (False, sm.astype(logits.dtype))):
        m = uncertainty_maps(torch.from_numpy(xin).unsqueeze(0).cuda(), from_logits=from_logits, want_sample_argmax=True)
        np.testing.assert_allclose(m["pred_entropy"][0].cpu().numpy(), ref["pred_entropy"], atol=5e-6)
        np.testing.assert_allclose(m["expected_entropy"][0].cpu().numpy(), ref["aleatoric_uncertainty"], atol=5e-6)
        np.testing.assert_allclose(m["mutual_information"][0].cpu().numpy(), ref["epistemic_uncertainty"], atol=5e-6)
        np.testing.assert_allclose(m["mean_softmax"][0].cpu().numpy(), mean, atol=2e-6)
        srt = np.sort(mean, axis=0)
        clear = (srt[-1] - srt[-2]) > 1e-5
        assert (m["argmax"][0].cpu().numpy() == mean_seg)[clear].all()
        ssrt = np.sort(sm, axis=1)
        sclear = (ssrt[:, -1] - ssrt[:, -2]) > 1e-5
        assert (m["sample_argmax"][0].cpu().numpy() == pred_seg)[sclear].all()


def test_unc_reduce_edge_cases():
    from values_amd import uncertainty_maps
    lib = _lib.load()
    # empty volume: no launch, no error
    x = torch.zeros((1, 3, 2, 0), device="cuda")
    m = uncertainty_maps(x)
    assert m["pred_entropy"].shape == (1, 0)
    # batched volumes == per-volume calls (bit exact)
    x = torch.from_numpy(formula_tensor((3, 5, 2, 8, 8, 8), 181, scale=3.0)).float().cuda()
    mb = uncertainty_maps(x, from_logits=True)
    for b in range(3):
        m1 = uncertainty_maps(x[b:b + 1], from_logits=True)
        for k in ("pred_entropy", "expected_entropy", "mutual_information", "mean_softmax", "argmax"):
            assert torch.equal(mb[k][b], m1[k][0])
    # argument errors come back as codes + message, not crashes
    rc = lib.vx_unc_reduce(None, 0, 0, 1, 1, 2, 8, None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.vx_last_error_string()
    y = torch.zeros(2 * 9 * 8, device="cuda")
    o = torch.zeros(8, device="cuda")
    rc = lib.vx_unc_reduce(_lib.ptr(y), 0, 1, 1, 2, 9, 8, None, _lib.ptr(o), _lib.ptr(o), _lib.ptr(o), None, None, None)
    assert rc == -2  # from_logits with C > 8
    # T = 1: expected entropy == predictive entropy, MI == 0
    p = torch.softmax(torch.from_numpy(formula_tensor((1, 1, 4, 64), 182, scale=2.0)).float().cuda(), 2)
    m = uncertainty_maps(p)
    assert m["mutual_information"].abs().max().item() < 1e-6
    # maximum uint8 class count
    big = torch.softmax(torch.from_numpy(formula_tensor((1, 2, 255, 16), 183, scale=2.0)).float().cuda(), 2)
    m = uncertainty_maps(big)
    assert torch.equal(m["argmax"][0].long().cpu(), big.mean(1)[0].argmax(0).cpu())


# ------------------------------------------------------------------------------------------- aggregation
def test_aggregations_match_reference_fixture():
    import json
    import os
    from tests.helpers import GOLDEN
    from values_amd import aggregation as agg
    with open(os.path.join(GOLDEN, "agg_kat.json")) as f:
        g = json.load(f)
    for size in (24, 64):
        img = np.abs(formula_tensor((size,) * 3, tag=g[f"vol{size}_tag"], scale=0.7)).astype(np.float32)
        c = size // 3
        img[c:c + 6, c + 2:c + 9, c + 1:c + 7] += 0.5
        r = g[f"vol{size}"]
        for key, kw in (("patch10", dict(patch_size=10)), ("patch10_mean", dict(patch_size=10, mean=True)),
                        ("patch_5_7_9", dict(patch_size=[5, 7, 9]))):
            mine = agg.patch_level_aggregation(img, **kw)
            assert mine["max_score"] == pytest.approx(r[key]["max_score"], rel=1e-6)
            assert [list(b) for b in mine["bounding_box"]] == [list(b) for b in r[key]["bounding_box"]]
        assert agg.image_level_aggregation(img)["max_score"] == pytest.approx(r["image"]["max_score"], rel=1e-6)
        assert agg.image_level_aggregation(img, mean=True) == pytest.approx(r["image_mean"], rel=1e-6)
        for thr in (0.3, 0.6, 5.0):
            for mean in (True, False):
                mine = agg.threshold_aggregation(img, threshold=thr, mean=mean)
                assert float(mine["max_score"]) == pytest.approx(r[f"thr_{thr}_{int(mean)}"]["max_score"], rel=1e-6)
                assert mine["threshold"] == thr
    img2 = np.abs(formula_tensor((40, 56), tag=33, scale=1.0)).astype(np.float32)
    mine = agg.patch_level_aggregation(img2, patch_size=10)
    assert mine["max_score"] == pytest.approx(g["img2d"]["patch10"]["max_score"], rel=1e-6)
    assert [list(b) for b in mine["bounding_box"]] == [list(b) for b in g["img2d"]["patch10"]["bounding_box"]]
    with pytest.raises(Exception):
        agg.threshold_aggregation(img2)


def test_softmax_variance_matches_numpy():
    """north_star's softmax-variance map (no reference counterpart): mean over classes of the variance over samples"""
    from values_amd.uncertainty import softmax_variance
    logits = formula_tensor((2, 6, 3, 5, 7, 9), 301, scale=2.0).astype(np.float32)
    e = np.exp(logits.astype(np.float64) - logits.max(2, keepdims=True))
    p = e / e.sum(2, keepdims=True)
    want = p.var(axis=1).mean(axis=1)
    got = softmax_variance(torch.from_numpy(logits).cuda(), from_logits=True).cpu().numpy()
    assert np.abs(got - want).max() < 1e-6
    got = softmax_variance(torch.from_numpy(p.astype(np.float32)).cuda(), from_logits=False).cpu().numpy()
    assert np.abs(got - want).max() < 1e-6


# ---------------------------------------------------------------------------------------------------------------
# conv3d_xp8.hip: the z-column kernel of the full-resolution Cout = 8 layers (rolling LDS window, per-column statistics,
# optional normalise-on-load prologue).  The dispatch takes it when W % 32 == 0, H % 8 == 0, D % 4 == 0, D >= 8.
def _hash_mask(seed, layer, n, c, d, h, w):
    """keep-mask of the hash bit generator in the reference's (N, C, D, H, W) layout"""
    lib = _lib.load()
    m = torch.empty((n, d, h, w, c), dtype=torch.uint8, device=dev())
    _lib.check(lib.vx_drop_hash_mask(seed, layer, n, d * h * w * c, _lib.ptr(m), _lib.stream_ptr()), "mask")
    return ncdhw(m).cpu().double()


def _xp8_conv(x_cl, cin, w, b, n, d, h, wd, *, act=0, drop=0, seed=0, layer=0, stats=False, xblk=0, head=None,
              pre=None, out_xblk=0, up=None, in_pitch=None, presplit=False, compose=False, up_split=False):
    """one vx_conv3d_k3 launch on a channels-last (or x-blocked) device input; returns (out NCDHW cpu, stats, head)"""
    lib = _lib.load()
    assert lib.vx_conv3d_k3_prologue_ok(d, h, wd, cin, 8) == 1
    wdv, bd = w.float().contiguous().to(dev()), b.float().contiguous().to(dev())
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, 8), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wdv), _lib.ptr(wp), cin, 8, _lib.stream_ptr()), "pack")
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, 8)
    out = torch.full((n, d, h, wd, 16 if out_xblk else 8), -77.0, dtype=torch.float32, device=dev())
    a.in_ = x_cl.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = in_pitch or cin, 8, 0
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, wd, cin, 8
    a.act, a.drop_mode, a.drop_seed, a.drop_layer, a.in_xblk = act, drop, seed, layer, xblk
    if up is not None:   # (coarse channels-last device tensor, torch ConvTranspose3d weight (16, 8, 2, 2, 2), bias)
        coarse, uw, ub = up
        uwd, ubd = uw.float().contiguous().to(dev()), ub.float().contiguous().to(dev())
        uwp = torch.empty(lib.vx_convT_k2s2_packed_floats(16, 8), dtype=torch.float32, device=dev())
        _lib.check(lib.vx_pack_convT_k2s2(_lib.ptr(uwd), _lib.ptr(uwp), 16, 8, _lib.stream_ptr()), "packT")
        a.up_in, a.up_w, a.up_b, a.up_pitch = coarse.data_ptr(), uwp.data_ptr(), ubd.data_ptr(), coarse.shape[-1]
        if up_split:        # the coarse tensor handed over as fp16 (hi, lo) pairs (vx_conv3d_args.out_split of the producing conv):
            # vx_prenorm_split with mean 0, rstd 1 would apply a LeakyReLU -- build the pairs here instead
            cs = coarse.clone()
            v = cs[..., :16].reshape(-1, 4)
            hi = v.half()
            lo = ((v - hi.float()) * 2048.0).half()
            cs[..., :16] = torch.cat([hi, lo], 1).view(torch.float32).reshape(cs[..., :16].shape)
            a.up_in, a.up_split = cs.data_ptr(), 1
            uwp = (uwp, cs)
        if compose:         # the up-convolution composed into the conv's weights (vx_pack_conv3d_upfused)
            uf = torch.empty(lib.vx_conv3d_upfused_packed_floats(), dtype=torch.float32, device=dev())
            _lib.check(lib.vx_pack_conv3d_upfused(_lib.ptr(wdv), _lib.ptr(bd), _lib.ptr(uwd), _lib.ptr(ubd), _lib.ptr(uf),
                                                  _lib.stream_ptr()), "vx_pack_conv3d_upfused")
            a.up_fused = uf.data_ptr()
            uwp = (uwp, uf)
    a.out_xblk, a.out_half = out_xblk, 1
    keep = [wdv, bd, wp, out]
    st = None
    if stats:
        nt = lib.vx_conv3d_k3_tiles_for(d, h, wd, 8)
        st = torch.full((n, nt, 8, 2), 5.0, dtype=torch.float32, device=dev())     # the kernel must overwrite every entry
        a.stats_partial = st.data_ptr()
    ho = None
    if head is not None:
        hw, hb, dst, flip, slots = head
        ho = torch.full((slots, hw.shape[0], d, h, wd), -5.0, dtype=torch.float32, device=dev())
        a.out = None
        a.head_out, a.head_w, a.head_b, a.head_C = ho.data_ptr(), hw.data_ptr(), hb.data_ptr(), hw.shape[0]
        a.head_dst, a.head_flip = dst.data_ptr(), flip.data_ptr()
    if pre is not None:
        mean, rstd, rep, pmode, pseed, player = pre
        a.in_mean, a.in_rstd, a.in_repeat = mean.data_ptr(), rstd.data_ptr(), rep
        a.in_drop_mode, a.in_drop_seed, a.in_drop_layer = pmode, pseed, player
        keep += [mean, rstd]
        if presplit:   # the shared raw tensor goes through vx_prenorm_split first (in place, on a copy), the conv then only masks
            x_cl = x_cl.clone()
            n_in, nvox = x_cl.shape[0], d * h * wd
            _lib.check(lib.vx_prenorm_split(_lib.ptr(x_cl), _lib.ptr(mean), _lib.ptr(rstd), n_in, nvox,
                                            2.0 if pmode == _lib.VX_DROP_HASH else 1.0, _lib.stream_ptr()), "vx_prenorm_split")
            a.in_ = x_cl.data_ptr()
            a.in_split = 1
            keep.append(x_cl)
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    a.range_flag = flag.data_ptr()
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    torch.cuda.synchronize()
    if out_xblk:
        res = from_xblk(out, 1, n, 8, d, h, wd, out_xblk).cpu()
        assert (from_xblk(out, 0, n, 8, d, h, wd, out_xblk) == -77.0).all()      # the other half is not touched
    else:
        res = ncdhw(out).cpu()
    return res, (st.cpu() if st is not None else None), (ho.cpu() if ho is not None else None), flag.view(torch.float32).item()


@pytest.mark.parametrize("cin,shape,xblk", [(8, (3, 8, 16, 64), 0), (8, (2, 16, 8, 32), 0), (8, (5, 12, 24, 96), 0),
                                            (16, (3, 8, 16, 64), 0), (16, (2, 12, 8, 32), 4), (16, (5, 16, 24, 64), 4)])
def test_conv3d_xp8_epilogues_match_oracle(cin, shape, xblk):
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 301))
    wt = torch.from_numpy(formula_tensor((8, cin, 3, 3, 3), 302, scale=(1.0 / (27 * cin)) ** 0.5))
    b = torch.from_numpy(formula_tensor((8,), 303, scale=0.2))
    xd = (to_xblk(x[:, :8].float(), x[:, 8:].float(), xblk) if xblk else cl(x.float())).to(dev())
    ref = F.conv3d(x.float().double(), wt.float().double(), b.float().double(), padding=1)
    # plain + statistics (an InstanceNorm follows): every column's partials, the padding entries zero
    got, st, _, mx = _xp8_conv(xd, cin, wt, b, n, d, h, w, stats=True, xblk=xblk)
    assert (got.double() - ref).abs().max().item() < 2e-5
    s = st.double().sum(1)
    np.testing.assert_allclose(s[..., 0].numpy(), ref.sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[..., 1].numpy(), (ref * ref).sum((2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3)
    assert mx == 0.0                                          # range guard: ordinary magnitudes are not reported ...
    gotb, _, _, mxb = _xp8_conv(xd, cin, wt, b + 5e4, n, d, h, w, xblk=xblk)
    assert abs(mxb - gotb.abs().max().item()) < 1e-2 * 5e4 and mxb > 4e4      # ... values near the fp16 limit are (either kernel)
    # LeakyReLU + hash dropout: the bit generator's mask exported and applied to the oracle
    got, _, _, _ = _xp8_conv(xd, cin, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=91, layer=6, xblk=xblk)
    keep = _hash_mask(91, 6, n, 8, d, h, w)
    assert 0.45 < keep.mean().item() < 0.55
    assert (got.double() - F.leaky_relu(ref, 0.01) * keep * 2).abs().max().item() < 4e-5
    # ReLU, no dropout; and the concat-buffer output form (skip half written as dense blocks)
    got, _, _, _ = _xp8_conv(xd, cin, wt, b, n, d, h, w, act=_lib.VX_ACT_RELU, xblk=xblk)
    assert (got.double() - F.relu(ref)).abs().max().item() < 2e-5
    got, _, _, _ = _xp8_conv(xd, cin, wt, b, n, d, h, w, xblk=xblk, out_xblk=4)
    assert (got.double() - ref).abs().max().item() < 2e-5
    if cin == 8:   # fused 1x1x1 head with scattered slots and un-flips (expand_1_2)
        hw = torch.from_numpy(formula_tensor((3, 8), 304, scale=0.3)).float().contiguous().to(dev())
        hb = torch.from_numpy(formula_tensor((3,), 305, scale=0.2)).float().to(dev())
        dst = torch.tensor([(n - 1 - i) * 2 for i in range(n)], dtype=torch.int32, device=dev())
        flip = torch.tensor([(3 * i + 5) % 8 for i in range(n)], dtype=torch.int32, device=dev())
        _, _, ho, _ = _xp8_conv(xd, cin, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=92, layer=16,
                                head=(hw, hb, dst, flip, 2 * n))
        feat = F.leaky_relu(ref, 0.01) * _hash_mask(92, 16, n, 8, d, h, w) * 2
        logits = torch.einsum("kc,ncdhw->nkdhw", hw.cpu().double(), feat) + hb.cpu().double().view(1, 3, 1, 1, 1)
        for i in range(n):
            dims = [ax + 1 for ax in range(3) if (int(flip[i]) >> ax) & 1]
            want = torch.flip(logits[i], dims) if dims else logits[i]
            assert (ho[int(dst[i])].double() - want).abs().max().item() < 4e-5, i
        assert (ho[1::2] == -5.0).all()                       # unused slots untouched


@pytest.mark.parametrize("cin,shape,rep,xblk,pmode", [(8, (2, 8, 16, 64), 3, 0, 1), (8, (3, 12, 8, 32), 1, 0, 1),
                                                      (8, (2, 8, 8, 32), 2, 0, 0), (16, (2, 8, 16, 32), 1, 4, 1),
                                                      (16, (1, 12, 8, 64), 1, 0, 1)])
def test_conv3d_xp8_prologue_matches_oracle(cin, shape, rep, xblk, pmode):
    """the normalise-on-load prologue (unet3D_module.py:231-237 applied to the conv's input while it is staged): n_in raw
    tensors, each read by `rep` output samples with their own dropout pattern; for a concat input only the skip half"""
    n_in, d, h, w = shape
    n = n_in * rep
    raw = torch.from_numpy(formula_tensor((n_in, 8, d, h, w), 311, scale=2.0)) + 0.7           # the raw contract-block conv output
    up = torch.from_numpy(formula_tensor((n, 8, d, h, w), 312)) if cin == 16 else None           # up half of a concat (no prologue)
    wt = torch.from_numpy(formula_tensor((8, cin, 3, 3, 3), 313, scale=(1.0 / (27 * cin)) ** 0.5))
    b = torch.from_numpy(formula_tensor((8,), 314, scale=0.2))
    rf = raw.float()
    mean = rf.double().mean((2, 3, 4))
    rstd = 1.0 / torch.sqrt(rf.double().var((2, 3, 4), unbiased=False) + 1e-5)
    meand, rstdd = mean.float().contiguous().to(dev()), rstd.float().contiguous().to(dev())
    keep = _hash_mask(55, 1, n, 8, d, h, w) if pmode else torch.ones((n, 8, d, h, w), dtype=torch.float64)
    norm = F.leaky_relu((rf.double() - meand.cpu().double().view(n_in, 8, 1, 1, 1)) * rstdd.cpu().double().view(n_in, 8, 1, 1, 1), 0.01)
    skip = norm.repeat_interleave(rep, 0) * keep * (2.0 if pmode else 1.0)
    xin = torch.cat([up.float().double(), skip], 1) if cin == 16 else skip
    ref = F.conv3d(xin, wt.float().double(), b.float().double(), padding=1)
    if cin == 16:
        assert rep == 1
        xd = (to_xblk(up.float(), rf, xblk) if xblk else cl(torch.cat([up.float(), rf], 1))).to(dev())
    else:
        xd = cl(rf).to(dev())
    got, st, _, _ = _xp8_conv(xd, cin, wt, b, n, d, h, w, stats=(cin == 8), xblk=xblk,
                              act=_lib.VX_ACT_NONE if cin == 8 else _lib.VX_ACT_LRELU,
                              pre=(meand, rstdd, rep, _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE, 55, 1))
    want = ref if cin == 8 else F.leaky_relu(ref, 0.01)
    err = (got.double() - want).abs().max().item()
    assert err < 4e-5, err
    if st is not None:
        s = st.double().sum(1)
        np.testing.assert_allclose(s[..., 0].numpy(), ref.sum((2, 3, 4)).numpy(), rtol=1e-4, atol=2e-3)
    if cin == 8:
        # the same conv on the PRE-SPLIT tensor (vx_prenorm_split once per input sample, the staging waves only apply each
        # output sample's dropout bits): the same normalised values, the same fp16 pairs, the same products -> the same bits
        got2, st2, _, _ = _xp8_conv(xd, cin, wt, b, n, d, h, w, stats=True, act=_lib.VX_ACT_NONE, presplit=True,
                                    pre=(meand, rstdd, rep, _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE, 55, 1))
        assert _lib.load().vx_last_kernel_name().decode().startswith("conv3d_xp8w_kernel<1,0,2,0,")
        assert torch.equal(got2, got) and torch.equal(st2, st)


@pytest.mark.parametrize("shape,xblk,pitch,pmode", [((2, 8, 16, 64), 4, 16, None), ((3, 12, 8, 32), 0, 16, None),
                                                    ((1, 16, 24, 96), 4, 20, 1), ((2, 8, 8, 32), 0, 16, 0)])
def test_conv3d_xp8_fused_upconvolution_matches_oracle(shape, xblk, pitch, pmode, vxcfg):
    """upscale -> torch.cat([up, skip], 1) -> expand (unet3D_module.py:332-356) in ONE launch: the transposed
    convolution of the coarse tensor is evaluated while the conv stages its tiles; the up half is neither read nor
    written.  Skip half from a concat buffer (whose up half holds garbage) or from a plain 8-channel tensor; with and
    without the normalise-on-load prologue on the skip half; coarse pitch > 16."""
    lib = _lib.load()
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_upfuse_ok(d, h, w, 16, 8) == 1 and lib.vx_conv3d_k3_upfuse_ok(d, h, w, 8, 8) == 0
    coarse = torch.from_numpy(formula_tensor((n, 16, d // 2, h // 2, w // 2), 331, scale=1.5)).float()
    uw = torch.from_numpy(formula_tensor((16, 8, 2, 2, 2), 332, scale=0.25)).float()
    ub = torch.from_numpy(formula_tensor((8,), 333, scale=0.3)).float()
    raw = (torch.from_numpy(formula_tensor((n, 8, d, h, w), 334, scale=2.0)) + 0.4).float()
    wt = torch.from_numpy(formula_tensor((8, 16, 3, 3, 3), 335, scale=(1.0 / (27 * 16)) ** 0.5))
    b = torch.from_numpy(formula_tensor((8,), 336, scale=0.2))
    up_ref = F.conv_transpose3d(coarse.double(), uw.double(), ub.double(), stride=2)
    pre = None
    skip = raw.double()
    if pmode is not None:
        mean = raw.double().mean((2, 3, 4)).float().contiguous().to(dev())
        rstd = (1.0 / torch.sqrt(raw.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
        keep = _hash_mask(77, 1, n, 8, d, h, w) if pmode else torch.ones((n, 8, d, h, w), dtype=torch.float64)
        skip = F.leaky_relu((raw.double() - mean.cpu().double().view(n, 8, 1, 1, 1)) * rstd.cpu().double().view(n, 8, 1, 1, 1), 0.01)
        skip = skip * keep * (2.0 if pmode else 1.0)
        pre = (mean, rstd, 1, _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE, 77, 1)
    ref = F.conv3d(torch.cat([up_ref, skip], 1), wt.float().double(), b.float().double(), padding=1)
    cd = cl(coarse)
    if pitch > 16:
        cd = torch.cat([cd, torch.full(cd.shape[:-1] + (pitch - 16,), 9.0)], -1).contiguous()
    cd = cd.to(dev())
    garbage = torch.full((n, 8, d, h, w), 1e30)
    xd = (to_xblk(garbage, raw, xblk) if xblk else cl(raw)).to(dev())
    got, _, _, mx = _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=93, layer=15,
                              xblk=xblk, in_pitch=8 if not xblk else None, pre=pre, up=(cd, uw, ub))
    assert lib.vx_last_kernel_name().decode().startswith("conv3d_xp8w_kernel<2,1,%d,1," % (0 if pre is None else 1))
    want = F.leaky_relu(ref, 0.01) * _hash_mask(93, 15, n, 8, d, h, w) * 2
    err = (got.double() - want).abs().max().item()
    assert err < 4e-5, err
    assert mx == 0.0
    got, _, _, _ = _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, xblk=xblk, in_pitch=8 if not xblk else None,
                             pre=pre, up=(cd, uw, ub))                       # no dropout (eval-mode members)
    assert (got.double() - F.leaky_relu(ref, 0.01)).abs().max().item() < 4e-5
    # range guard: up values near the fp16 limit are reported although they are never stored
    _, _, _, mxb = _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, xblk=xblk, in_pitch=8 if not xblk else None,
                             pre=pre, up=(cd, uw, ub + 4.5e4))
    assert mxb > 4e4
    # Round 4: the up-convolution COMPOSED into the conv's weights (vx_pack_conv3d_upfused -> vx_conv3d_args.up_fused): per
    # output parity class a 2 x 2 x 3-tap convolution over the 16 coarse channels, the up bias through a table of the 27
    # border classes.  Same function, the same oracle, the same tolerance; with and without dropout, the coarse tensor as
    # plain floats and as the fp16 pairs expand_2_2's epilogue hands over; and the knob that switches it off
    got, _, _, mx = _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=93, layer=15,
                              xblk=xblk, in_pitch=8 if not xblk else None, pre=pre, up=(cd, uw, ub), compose=True)
    assert lib.vx_last_kernel_name().decode().startswith("conv3d_xp8w_kernel<2,1,%d,2," % (0 if pre is None else 1))
    err = (got.double() - want).abs().max().item()
    assert err < 4e-5, err
    got, _, _, _ = _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, xblk=xblk, in_pitch=8 if not xblk else None,
                             pre=pre, up=(cd, uw, ub), compose=True, up_split=True)
    assert (got.double() - F.leaky_relu(ref, 0.01)).abs().max().item() < 4e-5
    got1, _, _, _ = _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, xblk=xblk, in_pitch=8 if not xblk else None,
                              pre=pre, up=(cd, uw, ub), up_split=True)           # per-step evaluation on the pre-split tensor
    assert (got1.double() - F.leaky_relu(ref, 0.01)).abs().max().item() < 4e-5
    vxcfg.set(s16_no_upcompose=1)
    _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, xblk=xblk, in_pitch=8 if not xblk else None, pre=pre,
              up=(cd, uw, ub), compose=True)
    assert lib.vx_last_kernel_name().decode().startswith("conv3d_xp8w_kernel<2,3,%d,1," % (0 if pre is None else 1))
    vxcfg.set(s16_no_upcompose=0)
    # refused, not silently ignored, where the column kernel does not run
    vxcfg.set(s16_no_upfuse=1)
    with pytest.raises(_lib.VxError):
        _xp8_conv(xd, 16, wt, b, n, d, h, w, act=_lib.VX_ACT_LRELU, xblk=xblk, in_pitch=8 if not xblk else None, pre=pre,
                  up=(cd, uw, ub))


def test_conv3d_xp8_agrees_with_general_kernel_and_refuses_what_it_cannot_do(vxcfg):
    lib = _lib.load()
    n, d, h, w = 2, 8, 8, 64
    x = torch.from_numpy(formula_tensor((n, 8, d, h, w), 321))
    wt = torch.from_numpy(formula_tensor((8, 8, 3, 3, 3), 322, scale=0.07))
    b = torch.from_numpy(formula_tensor((8,), 323, scale=0.2))
    a1, s1, _ = run_conv(x, wt, b, stats=True)
    vxcfg.set(s16_no_xp8=1)
    assert lib.vx_conv3d_k3_prologue_ok(d, h, w, 8, 8) == 0
    a0, s0, _ = run_conv(x, wt, b, stats=True)
    assert (a1 - a0).abs().max().item() < 2e-6                 # same products, the bias enters at another point of the sum
    assert (s1.double().sum(1) - s0.double().sum(1)).abs().max().item() < 1e-2
    # a prologue on a layer the column kernel does not take is an error, not a silently un-normalised input
    mean = torch.zeros((n, 8), device=dev()); rstd = torch.ones((n, 8), device=dev())
    with pytest.raises(_lib.VxError):
        _xp8_conv.__wrapped__ if hasattr(_xp8_conv, "__wrapped__") else None
        a = _lib.ConvArgs()
        xd = cl(x.float()).to(dev())
        wp = torch.empty(lib.vx_conv3d_k3_packed_floats(8, 8), dtype=torch.float32, device=dev())
        out = torch.empty((n, d, h, w, 8), device=dev())
        a.w_family = lib.vx_conv3d_k3_family(8, 8)
        a.in_ = xd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = mean.data_ptr(); a.out = out.data_ptr()
        a.in_pitch, a.out_pitch = 8, 8
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, 8, 8
        a.in_mean, a.in_rstd = mean.data_ptr(), rstd.data_ptr()
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")


def test_weights_packed_under_another_configuration_are_refused(vxcfg):
    """every launch carries the kernel family its weights were packed for (w_family): a raw C-ABI caller that packs
    under one vx_config and runs under another gets VX_E_DTYPE, not numbers from the wrong layout"""
    lib = _lib.load()
    x = torch.zeros((1, 8, 8, 16, 8), device=dev())
    wt = torch.zeros((8, 8, 3, 3, 3), device=dev())
    b = torch.zeros(8, device=dev())
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(8, 8), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wt), _lib.ptr(wp), 8, 8, _lib.stream_ptr()), "pack")
    out = torch.empty_like(x)
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(8, 8)
    a.in_ = x.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = b.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch = 8, 8
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = 1, 8, 8, 16, 8, 8
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "same configuration")
    vxcfg.set(conv_fp32=1)
    assert lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()) == -3      # VX_E_DTYPE
    assert b"family" in lib.vx_last_error_string()


@pytest.mark.parametrize("shape,pmode,rep", [((2, 8, 16, 64), 1, 1), ((1, 16, 8, 32), 0, 1), ((2, 8, 8, 32), 1, 3)])
def test_conv3d_xp8_pooled_output_matches_oracle(shape, pmode, rep, vxcfg):
    """contract block tail in two launches (unet3D_module.py:231-237, 303-310): the conv leaves its raw output, its
    statistics AND, per 2 x 2 x 2 window, the maximum of the raw values the block's dropout keeps plus an any-dropped bit;
    vx_pool_finish applies InstanceNorm + LeakyReLU + the dropout's 2 to those -- equal to
    MaxPool3d(Dropout(LeakyReLU(InstanceNorm(conv)))) computed from the full tensor (the functions in between are
    monotone).  With the normalise-on-load prologue in front (rep > 1: MC-dropout samples sharing the input) and without."""
    lib = _lib.load()
    n_in, d, h, w = shape
    n = n_in * rep
    assert lib.vx_conv3d_k3_poolfuse_ok(d, h, w, 8, 8) == 1 and lib.vx_conv3d_k3_poolfuse_ok(d, h, w, 16, 8) == 0
    x = torch.from_numpy(formula_tensor((n_in, 8, d, h, w), 341, scale=1.5)).float()
    wt = torch.from_numpy(formula_tensor((8, 8, 3, 3, 3), 342, scale=(1.0 / (27 * 8)) ** 0.5))
    b = torch.from_numpy(formula_tensor((8,), 343, scale=0.2))
    pre = None
    xin = x.double().repeat_interleave(rep, 0)
    if rep > 1:     # the input is a raw tensor normalised on load with the previous block's dropout (layer 0)
        mean0 = x.double().mean((2, 3, 4)).float().contiguous().to(dev())
        rstd0 = (1.0 / torch.sqrt(x.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
        keep0 = _hash_mask(21, 0, n, 8, d, h, w)
        xin = F.leaky_relu((x.double() - mean0.cpu().double().view(n_in, 8, 1, 1, 1)) * rstd0.cpu().double().view(n_in, 8, 1, 1, 1), 0.01)
        xin = xin.repeat_interleave(rep, 0) * keep0 * 2
        pre = (mean0, rstd0, rep, _lib.VX_DROP_HASH, 21, 0)
    ref = F.conv3d(xin, wt.float().double(), b.float().double(), padding=1)
    # --- launch 1: conv + statistics + window maxima
    wdv, bd = wt.float().contiguous().to(dev()), b.float().contiguous().to(dev())
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(8, 8), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wdv), _lib.ptr(wp), 8, 8, _lib.stream_ptr()), "pack")
    xd = cl(x).to(dev())
    out = torch.full((n, d, h, w, 8), -77.0, dtype=torch.float32, device=dev())
    nt = lib.vx_conv3d_k3_tiles_for(d, h, w, 8)
    st = torch.full((n, nt, 8, 2), 5.0, dtype=torch.float32, device=dev())
    praw = torch.full((n, d // 2, h // 2, w // 2, 8), 123.0, dtype=torch.float32, device=dev())
    pfl = torch.full((n, d // 2, h // 2, w // 2, 2), -1, dtype=torch.int32, device=dev())
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(8, 8)
    a.in_ = xd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = 8, 8, 0
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, 8, 8
    a.stats_partial = st.data_ptr()
    a.pool_out, a.pool_flags = praw.data_ptr(), pfl.data_ptr()
    a.drop_mode, a.drop_seed, a.drop_layer = (_lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE), 57, 1
    if pre is not None:
        a.in_mean, a.in_rstd, a.in_repeat = pre[0].data_ptr(), pre[1].data_ptr(), pre[2]
        a.in_drop_mode, a.in_drop_seed, a.in_drop_layer = pre[3], pre[4], pre[5]
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    assert lib.vx_last_kernel_name().decode().startswith("conv3d_xp8w_kernel<1,4,%d,0," % (0 if pre is None else 1))
    got = ncdhw(out).cpu()
    assert (got.double() - ref).abs().max().item() < 4e-5
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[..., 0].numpy(), ref.sum((2, 3, 4)).numpy(), rtol=1e-4, atol=2e-3)
    # --- launch 2: statistics of the conv's own output, then the pooled tensor
    mean = got.double().mean((2, 3, 4)).float().contiguous().to(dev())
    rstd = (1.0 / torch.sqrt(got.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
    pooled = torch.full((n, d // 2, h // 2, w // 2, 12), -9.0, dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pool_finish(_lib.ptr(praw), _lib.ptr(pfl), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(pooled), 12, n,
                                  (d // 2) * (h // 2) * (w // 2), 1 if pmode else 0, _lib.stream_ptr()), "vx_pool_finish")
    torch.cuda.synchronize()
    keep = _hash_mask(57, 1, n, 8, d, h, w) if pmode else torch.ones((n, 8, d, h, w), dtype=torch.float64)
    # the float32 arithmetic of the product on the conv's float32 output: (x - mean) * rstd, LeakyReLU, x 2, mask, max
    t = (got - mean.cpu().view(n, 8, 1, 1, 1)) * rstd.cpu().view(n, 8, 1, 1, 1)
    t = torch.maximum(t, 0.01 * t) * (2.0 if pmode else 1.0) * keep.float()
    want = F.max_pool3d(t, 2, 2)
    assert torch.equal(ncdhw(pooled[..., :8]).cpu(), want)            # bit for bit: every function in between is monotone
    assert (pooled[..., 8:] == -9.0).all()
    if pmode:
        frac = (pfl.cpu() & 0xF).float().ne(0).float().mean().item()
        assert frac > 0.9                                            # 8 elements per window and channel: a drop almost surely
    # --- Round 4: the NEXT block's first conv (8 -> 16 on the pooled grid) finishing the window maxima itself while it stages
    # its tiles (vx_conv3d_args.in_pool_flags) = vx_pool_finish + the plain conv, bit for bit: output and statistics
    assert lib.vx_conv3d_k3_poolfin_ok(8, 16) == 1 and lib.vx_conv3d_k3_poolfin_ok(8, 8) == 0 and lib.vx_conv3d_k3_poolfin_ok(16, 16) == 0
    dp, hp, wq = d // 2, h // 2, w // 2
    w2 = torch.from_numpy(formula_tensor((16, 8, 3, 3, 3), 344, scale=(1.0 / (27 * 8)) ** 0.5)).float().contiguous().to(dev())
    b2 = torch.from_numpy(formula_tensor((16,), 345, scale=0.2)).float().to(dev())
    wp2 = torch.empty(lib.vx_conv3d_k3_packed_floats(8, 16), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(w2), _lib.ptr(wp2), 8, 16, _lib.stream_ptr()), "pack")
    nt2 = lib.vx_conv3d_k3_tiles_for(dp, hp, wq, 16)
    dense = pooled[..., :8].contiguous()

    def second(fused):
        o2 = torch.full((n, dp, hp, wq, 16), -3.0, dtype=torch.float32, device=dev())
        s2 = torch.zeros((n, nt2, 16, 2), dtype=torch.float32, device=dev())
        a2 = _lib.ConvArgs()
        a2.w_family = lib.vx_conv3d_k3_family(8, 16)
        a2.in_ = (praw if fused else dense).data_ptr(); a2.w_packed = wp2.data_ptr(); a2.bias = b2.data_ptr(); a2.out = o2.data_ptr()
        a2.in_pitch, a2.out_pitch, a2.out_coff = 8, 16, 0
        a2.N, a2.D, a2.H, a2.W, a2.Cin, a2.Cout = n, dp, hp, wq, 8, 16
        a2.stats_partial = s2.data_ptr()
        if fused:
            a2.in_mean, a2.in_rstd, a2.in_pool_flags = mean.data_ptr(), rstd.data_ptr(), pfl.data_ptr()
            a2.in_drop_mode = _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE
        _lib.check(lib.vx_conv3d_k3(C.byref(a2), _lib.stream_ptr()), "vx_conv3d_k3 (second)")
        torch.cuda.synchronize()
        return o2, s2
    o_sep, s_sep = second(False)
    o_fus, s_fus = second(True)
    assert torch.equal(o_sep, o_fus) and torch.equal(s_sep, s_fus)
    assert torch.isfinite(o_fus).all()
    # refused where no instance carries the pool-finish code (round-4 advice): two row tiles per wave (8 -> 32), and an
    # epilogue with activation + dropout (its compile-time instance ignores the flag words)
    assert lib.vx_conv3d_k3_poolfin_ok(8, 32) == 0 and lib.vx_conv3d_k3_poolfin_ok(8, 24) == 1
    a3 = _lib.ConvArgs()
    o3 = torch.zeros((n, dp, hp, wq, 16), dtype=torch.float32, device=dev())
    a3.w_family = lib.vx_conv3d_k3_family(8, 16)
    a3.in_ = praw.data_ptr(); a3.w_packed = wp2.data_ptr(); a3.bias = b2.data_ptr(); a3.out = o3.data_ptr()
    a3.in_pitch, a3.out_pitch, a3.out_coff = 8, 16, 0
    a3.N, a3.D, a3.H, a3.W, a3.Cin, a3.Cout = n, dp, hp, wq, 8, 16
    a3.in_mean, a3.in_rstd, a3.in_pool_flags = mean.data_ptr(), rstd.data_ptr(), pfl.data_ptr()
    a3.act, a3.drop_mode, a3.drop_seed, a3.drop_layer = _lib.VX_ACT_LRELU, _lib.VX_DROP_HASH, 3, 2
    with pytest.raises(_lib.VxError):
        _lib.check(lib.vx_conv3d_k3(C.byref(a3), _lib.stream_ptr()), "vx_conv3d_k3 (pool-finish + activation)")
    w32 = torch.zeros(lib.vx_conv3d_k3_packed_floats(8, 32), dtype=torch.float32, device=dev())
    o32 = torch.zeros((n, dp, hp, wq, 32), dtype=torch.float32, device=dev())
    b32 = torch.zeros(32, dtype=torch.float32, device=dev())
    a3.act, a3.drop_mode = _lib.VX_ACT_NONE, _lib.VX_DROP_NONE
    a3.w_family = lib.vx_conv3d_k3_family(8, 32)
    a3.w_packed, a3.bias, a3.out, a3.out_pitch, a3.Cout = w32.data_ptr(), b32.data_ptr(), o32.data_ptr(), 32, 32
    with pytest.raises(_lib.VxError):
        _lib.check(lib.vx_conv3d_k3(C.byref(a3), _lib.stream_ptr()), "vx_conv3d_k3 (pool-finish, 8 -> 32)")
    # refused where the z-column kernel does not run
    vxcfg.set(s16_no_xp8=1)
    with pytest.raises(_lib.VxError):
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")


@pytest.mark.parametrize("cin,cout,shape,pmode", [(16, 16, (2, 8, 16, 32), 1), (32, 32, (2, 6, 10, 12), 1), (64, 64, (3, 4, 4, 4), 1),
                                                  (16, 16, (1, 5, 33, 20), 0), (8, 16, (2, 4, 8, 16), 1), (128, 128, (2, 2, 2, 2), 1)])
def test_conv3d_tile_kernel_prologue_matches_oracle(cin, cout, shape, pmode, vxcfg):
    """normalise-on-load in the general split-fp16 tile kernel (the second conv of the contract blocks below full
    resolution, unet3D_module.py:231-237): raw input + statistics + the producing block's dropout -> the conv of the
    normalised, activated, dropped tensor; ragged tiles, several channel chunks, with statistics of its own."""
    lib = _lib.load()
    vxcfg.set(s16_no_zc16=1)       # (the first shape would otherwise take the z-column kernel of round 5: its own test below)
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_prologue_ok(d, h, w, cin, cout) == 1
    raw = (torch.from_numpy(formula_tensor((n, cin, d, h, w), 351, scale=2.0)) + 0.3).float()
    wt = torch.from_numpy(formula_tensor((cout, cin, 3, 3, 3), 352, scale=(1.0 / (27 * cin)) ** 0.5))
    b = torch.from_numpy(formula_tensor((cout,), 353, scale=0.2))
    mean = raw.double().mean((2, 3, 4)).float().contiguous().to(dev())
    rstd = (1.0 / torch.sqrt(raw.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
    keep = _hash_mask(61, 4, n, cin, d, h, w) if pmode else torch.ones((n, cin, d, h, w), dtype=torch.float64)
    xin = F.leaky_relu((raw.double() - mean.cpu().double().view(n, cin, 1, 1, 1)) * rstd.cpu().double().view(n, cin, 1, 1, 1), 0.01)
    xin = xin * keep * (2.0 if pmode else 1.0)
    ref = F.conv3d(xin, wt.float().double(), b.float().double(), padding=1)
    wdv, bd = wt.float().contiguous().to(dev()), b.float().contiguous().to(dev())
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wdv), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "pack")
    xd = cl(raw).to(dev())
    out = torch.full((n, d, h, w, cout), -77.0, dtype=torch.float32, device=dev())
    nt = lib.vx_conv3d_k3_tiles_for(d, h, w, cout)
    st = torch.zeros((n, nt, cout, 2), dtype=torch.float32, device=dev())
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, cout)
    a.in_ = xd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = cin, cout, 0
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, cout
    a.stats_partial = st.data_ptr()
    a.in_mean, a.in_rstd, a.in_repeat = mean.data_ptr(), rstd.data_ptr(), 1
    a.in_drop_mode, a.in_drop_seed, a.in_drop_layer = (_lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE), 61, 4
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    torch.cuda.synchronize()
    assert lib.vx_last_kernel_name().decode().startswith("conv3d_k3_s16_kernel")
    got = ncdhw(out).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err < 4e-5, err
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[..., 0].numpy(), ref.sum((2, 3, 4)).numpy(), rtol=1e-4, atol=2e-3)
    # a padded input pitch is refused (the element index of the dropout bits is derived from the dense layout)
    a.in_pitch = cin + 4
    with pytest.raises(_lib.VxError):
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: the role-split z-column kernel of the Cout = 16 layers (conv3d_zc16.hip)
def _zc16_launch(x_cl, cin, wp, bd, n, d, h, w, *, act=0, drop=0, seed=0, layer=0, stats=False, pre=None, out_xblk=0,
                 pool=False, out_split=False, poolfin=None, out_planar=False, in_planar=False):
    """one vx_conv3d_k3 launch (16 output channels) on a dense channels-last device input.
    Returns (out device tensor, stats or None, (pool_raw, pool_flags) or None, kernel name)"""
    lib = _lib.load()
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, 16)
    out = torch.full((n, d, h, w, 32 if out_xblk else 16), -77.0, dtype=torch.float32, device=dev())
    a.in_ = x_cl.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = cin, (32 if out_xblk else 16), 0
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, 16
    a.act, a.drop_mode, a.drop_seed, a.drop_layer = act, drop, seed, layer
    a.out_xblk, a.out_half, a.out_split = out_xblk, 1, 1 if out_split else 0
    a.out_planar, a.in_planar = (1 if out_planar else 0), (1 if in_planar else 0)
    st = None
    if stats:
        nt = lib.vx_conv3d_k3_tiles_for(d, h, w, 16)
        st = torch.full((n, nt, 16, 2), 5.0, dtype=torch.float32, device=dev())
        a.stats_partial = st.data_ptr()
    pl = None
    if pool:
        pl = (torch.full((n, d, h // 2, w // 2, 16), 123.0, dtype=torch.float32, device=dev()),
              torch.full((n, d, h // 2, w // 2, 4), -1, dtype=torch.int32, device=dev()))
        a.pool_out, a.pool_flags = pl[0].data_ptr(), pl[1].data_ptr()
    if pre is not None:
        a.in_mean, a.in_rstd, a.in_repeat = pre[0].data_ptr(), pre[1].data_ptr(), 1
        a.in_drop_mode, a.in_drop_seed, a.in_drop_layer = pre[2], pre[3], pre[4]
    if poolfin is not None:
        a.in_mean, a.in_rstd, a.in_pool_flags, a.in_drop_mode = poolfin[0].data_ptr(), poolfin[1].data_ptr(), poolfin[2].data_ptr(), poolfin[3]
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    torch.cuda.synchronize()
    return out, st, pl, lib.vx_last_kernel_name().decode()


def _pack16(cin, seed):
    lib = _lib.load()
    wt = torch.from_numpy(formula_tensor((16, cin, 3, 3, 3), seed, scale=(1.0 / (27 * cin)) ** 0.5)).float().contiguous()
    b = torch.from_numpy(formula_tensor((16,), seed + 1, scale=0.2)).float().contiguous()
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, 16), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wt.to(dev())), _lib.ptr(wp), cin, 16, _lib.stream_ptr()), "pack")
    return wt, b, wp, b.to(dev())


ZC16_SHAPES = [(2, 8, 16, 32), (1, 4, 8, 64), (3, 6, 24, 32)]


@pytest.mark.parametrize("shape", ZC16_SHAPES)
@pytest.mark.parametrize("cin", [16, 8])
def test_conv3d_zc16_plain_and_activation_epilogues_match_oracle(cin, shape, vxcfg):
    """conv3d_zc16.hip (round 5; unet3D_module.py:231-243, 263-267 at the levels below full resolution): bias + statistics,
    LeakyReLU / ReLU, LeakyReLU + hash dropout (+ the pre-split hand-over, + the x-blocked concat output) against the float64
    oracle, columns at every border class (1-3 columns in y, 1-2 in x, 2-4 items in z), several samples per workgroup; and the
    knob that switches the kernel off gives the tile kernel's numbers."""
    lib = _lib.load()
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_family(cin, 16) == 6
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 701, scale=1.5)).float()
    wt, b, wp, bd = _pack16(cin, 702)
    xd = cl(x).to(dev())
    ref = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    # --- bias + statistics (EPI 0)
    out, st, _, kn = _zc16_launch(xd, cin, wp, bd, n, d, h, w, stats=True)
    assert kn == "conv3d_zc16_kernel<%d,0,0,0,0>" % cin, kn
    got = ncdhw(out).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err < 4e-5, err
    ssum = st.double().sum(1).cpu()
    np.testing.assert_allclose(ssum[..., 0].numpy(), got.double().sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(ssum[..., 1].numpy(), (got.double() ** 2).sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    # --- run-time activation (EPI 3)
    for act, fn in ((_lib.VX_ACT_LRELU, lambda t: F.leaky_relu(t, 0.01)), (_lib.VX_ACT_RELU, F.relu)):
        o2, _, _, kn = _zc16_launch(xd, cin, wp, bd, n, d, h, w, act=act)
        assert kn == "conv3d_zc16_kernel<%d,3,0,0,0>" % cin, kn
        assert (ncdhw(o2).cpu().double() - fn(ref)).abs().max().item() < 4e-5
    # --- LeakyReLU + hash dropout (EPI 1), plain / pre-split / x-blocked
    keep = _hash_mask(77, 13, n, 16, d, h, w)
    want = F.leaky_relu(ref, 0.01) * keep * 2.0
    o3, _, _, kn = _zc16_launch(xd, cin, wp, bd, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=77, layer=13)
    assert kn == "conv3d_zc16_kernel<%d,1,0,0,0>" % cin, kn
    assert (ncdhw(o3).cpu().double() - want).abs().max().item() < 4e-5
    o4, _, _, _ = _zc16_launch(xd, cin, wp, bd, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=77, layer=13, out_split=True)
    hl = o4.cpu().contiguous().view(torch.float16).view(n, d, h, w, 4, 2, 4).float()      # [quad][hi | lo][4]
    back = (hl[..., 0, :] + hl[..., 1, :] / 2048.0).reshape(n, d, h, w, 16)
    assert (back - o3.cpu()).abs().max().item() <= 1e-6 * max(1.0, o3.abs().max().item())
    if w % 4 == 0:
        o5, _, _, _ = _zc16_launch(xd, cin, wp, bd, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=77, layer=13, out_xblk=4)
        blk = o5.cpu().view(n, d, h, w // 4, 2, 4, 16)
        assert torch.equal(blk[:, :, :, :, 1].reshape(n, d, h, w, 16), o3.cpu())      # the skip half holds the output
        assert (blk[:, :, :, :, 0] == -77.0).all()                                       # the up half is untouched
    # --- the knob: the tile kernel on the same launch
    vxcfg.set(s16_no_zc16=1)
    o6, st6, _, kn = _zc16_launch(xd, cin, wp, bd, n, d, h, w, stats=True)
    assert kn.startswith("conv3d_k3_s16_kernel"), kn
    assert (ncdhw(o6).cpu().double() - ref).abs().max().item() < 4e-5
    assert (o6 - out).abs().max().item() < 2e-5


@pytest.mark.parametrize("shape", ZC16_SHAPES + [(5, 4, 8, 32), (1, 12, 16, 64), (1, 4, 8, 96)])      # (the last: an interior column tile in x)
def test_conv3d_zc16_planar_presplit_handover_is_the_float_handover(shape, vxcfg):
    """Round 6 (verdict item 4; unet3D_module.py:263-267: expand_2_1 -> expand_2_2 with no normalisation in between): the producer's
    activation epilogue stores its 16 channels as fp16 (hi, lo) planes in the consumer's LDS row order (out_planar: instances
    <16,5,..> / <16,6,..>), the consumer stages that tensor by LDS-DMA (in_planar: <16,1,4,0,0> / <16,3,4,0,0>).  The planar tensor is
    the split of the float one, the consumer's output the SAME BITS as over the float hand-over, and the chain matches the float64
    oracle; two column tiles in x / three in y, an odd number of samples, hash dropout and none."""
    lib = _lib.load()
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_planar_ok(d, h, w, 16, 16) == 1 and lib.vx_conv3d_k3_planar_ok(d, h, 16, 16, 16) == 0
    x = torch.from_numpy(formula_tensor((n, 16, d, h, w), 831, scale=1.5)).float()
    w1, b1, wp1, bd1 = _pack16(16, 832)
    w2, b2, wp2, bd2 = _pack16(16, 834)
    xd = cl(x).to(dev())
    for drop in (_lib.VX_DROP_HASH, 0):
        kw = dict(act=_lib.VX_ACT_LRELU, drop=drop, seed=21)
        y1, _, _, k1 = _zc16_launch(xd, 16, wp1, bd1, n, d, h, w, layer=11, **kw)
        z1, _, _, k2 = _zc16_launch(y1, 16, wp2, bd2, n, d, h, w, layer=12, **kw)
        y2, _, _, k3 = _zc16_launch(xd, 16, wp1, bd1, n, d, h, w, layer=11, out_planar=True, **kw)
        z2, _, _, k4 = _zc16_launch(y2, 16, wp2, bd2, n, d, h, w, layer=12, in_planar=True, **kw)
        e = 1 if drop else 3
        assert k1.startswith(f"conv3d_zc16_kernel<16,{e},0,0,0>") and k3.startswith(f"conv3d_zc16_kernel<16,{5 if drop else 6},0,0,0>"), (k1, k3)
        assert k4.startswith(f"conv3d_zc16_kernel<16,{e},4,0,0>"), k4
        # the planar tensor: [n][d][h][octet][hi | lo][w][8 halves] -> channels-last float
        pl = y2.contiguous().view(torch.float16).view(n, d, h, 2, 2, w, 8).float()
        back = (pl[:, :, :, :, 0] + pl[:, :, :, :, 1] / 2048.0).permute(0, 1, 2, 4, 3, 5).reshape(n, d, h, w, 16)
        assert (back - y1).abs().max().item() <= 1e-6 * max(1.0, y1.abs().max().item())
        assert torch.equal(z2, z1), (z2 - z1).abs().max().item()
        ref1 = F.leaky_relu(F.conv3d(x.double(), w1.double(), b1.double(), padding=1), 0.01)
        if drop:
            ref1 = ref1 * _hash_mask(21, 11, n, 16, d, h, w) * 2.0
        ref2 = F.leaky_relu(F.conv3d(ref1, w2.double(), b2.double(), padding=1), 0.01)
        if drop:
            ref2 = ref2 * _hash_mask(21, 12, n, 16, d, h, w) * 2.0
        assert (ncdhw(z2).cpu().double() - ref2).abs().max().item() < 8e-5
    # refusals: only the 16 -> 16 layers of the z-column kernel, activation epilogues, no aliasing of the partial sums
    with pytest.raises(_lib.VxError):
        _zc16_launch(xd, 16, wp1, bd1, n, d, h, w, stats=True, out_planar=True)
    w8, b8, wp8, bd8 = _pack16(8, 836)
    with pytest.raises(_lib.VxError):
        _zc16_launch(xd[..., :8].contiguous(), 8, wp8, bd8, n, d, h, w, act=_lib.VX_ACT_RELU, in_planar=True)
    with pytest.raises(_lib.VxError):      # a shape the z-column kernel does not take: the tile kernel cannot read the planar layout
        _zc16_launch(xd[:, :, :, :16].contiguous(), 16, wp1, bd1, n, d, h, 16, act=_lib.VX_ACT_RELU, in_planar=True)


@pytest.mark.parametrize("shape", ZC16_SHAPES)
@pytest.mark.parametrize("pmode", [1, 0])
def test_conv3d_zc16_prologue_and_pooled_output_match_oracle(shape, pmode, vxcfg):
    """The second conv of a contract block on the z-column kernel (contr_2_2): normalise-on-load of the raw first conv
    (InstanceNorm + LeakyReLU + hash dropout in the staging waves), its own statistics, the raw output into the x-blocked skip
    half, and the (y, x) half of the block's MaxPool from the epilogue -- vx_pool_finish_z then gives MaxPool3d of the normalised,
    dropped tensor BIT FOR BIT (unet3D_module.py:231-237, 303-310)."""
    lib = _lib.load()
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_pool_layout(d, h, w, 16, 16) == 2 and lib.vx_conv3d_k3_poolfuse_ok(d, h, w, 16, 16) == 1
    raw = (torch.from_numpy(formula_tensor((n, 16, d, h, w), 711, scale=2.0)) + 0.3).float()
    wt, b, wp, bd = _pack16(16, 712)
    mean = raw.double().mean((2, 3, 4)).float().contiguous().to(dev())
    rstd = (1.0 / torch.sqrt(raw.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
    keep_in = _hash_mask(61, 4, n, 16, d, h, w) if pmode else torch.ones((n, 16, d, h, w), dtype=torch.float64)
    xin = F.leaky_relu((raw.double() - mean.cpu().double().view(n, 16, 1, 1, 1)) * rstd.cpu().double().view(n, 16, 1, 1, 1), 0.01)
    xin = xin * keep_in * (2.0 if pmode else 1.0)
    ref = F.conv3d(xin, wt.double(), b.double(), padding=1)
    xd = cl(raw).to(dev())
    pre = (mean, rstd, _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE, 61, 4)
    dmode = _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE
    out, st, pl, kn = _zc16_launch(xd, 16, wp, bd, n, d, h, w, stats=True, pre=pre, pool=True, drop=dmode, seed=57, layer=5, out_xblk=4)
    assert kn == "conv3d_zc16_kernel<16,4,1,0,0>", kn
    got_cl = out.view(n, d, h, w // 4, 2, 4, 16)[:, :, :, :, 1].reshape(n, d, h, w, 16).contiguous()
    got = ncdhw(got_cl).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err < 4e-5, err
    ssum = st.double().sum(1).cpu()
    np.testing.assert_allclose(ssum[..., 0].numpy(), got.double().sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    # the prologue without the pooled epilogue (EPI 0) gives the same output bits
    out0, _, _, kn = _zc16_launch(xd, 16, wp, bd, n, d, h, w, stats=True, pre=pre)
    assert kn == "conv3d_zc16_kernel<16,0,1,0,0>", kn
    assert torch.equal(out0, got_cl)
    # pooled tensor: statistics of the conv's own float32 output, then the z pair + the normalisation
    mean2 = got.double().mean((2, 3, 4)).float().contiguous().to(dev())
    rstd2 = (1.0 / torch.sqrt(got.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
    pooled = torch.full((n, d // 2, h // 2, w // 2, 20), -9.0, dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pool_finish_z(_lib.ptr(pl[0]), _lib.ptr(pl[1]), _lib.ptr(mean2), _lib.ptr(rstd2), _lib.ptr(pooled), 20, n, d // 2,
                                    (h // 2) * (w // 2), 1 if pmode else 0, _lib.stream_ptr()), "vx_pool_finish_z")
    torch.cuda.synchronize()
    keep = _hash_mask(57, 5, n, 16, d, h, w) if pmode else torch.ones((n, 16, d, h, w), dtype=torch.float64)
    t = (got - mean2.cpu().view(n, 16, 1, 1, 1)) * rstd2.cpu().view(n, 16, 1, 1, 1)
    t = torch.maximum(t, 0.01 * t) * (2.0 if pmode else 1.0) * keep.float()
    want = F.max_pool3d(t, 2, 2)
    assert torch.equal(ncdhw(pooled[..., :16]).cpu(), want)
    assert (pooled[..., 16:] == -9.0).all()
    if pmode:
        assert (pl[1].cpu() & 0xF).float().ne(0).float().mean().item() > 0.5


@pytest.mark.parametrize("pmode", [1, 0])
def test_conv3d_zc16_pool_finish_on_load_is_pool_finish_then_conv(pmode, vxcfg):
    """contr_2_1 on the z-column kernel: the previous block's window maxima + flags finished while the tiles are staged
    (vx_conv3d_args.in_pool_flags) = vx_pool_finish + the plain launch, bit for bit (output and statistics)."""
    lib = _lib.load()
    n, d, h, w = 2, 8, 16, 32
    rng = np.random.default_rng(3)
    praw = torch.from_numpy(rng.standard_normal((n, d, h, w, 8)).astype(np.float32) * 2.0 + 0.2)
    praw[0, 0, 0, :4] = -float("inf")                      # windows in which nothing was kept
    pfl = torch.from_numpy(rng.integers(0, 16, size=(n, d, h, w, 2)).astype(np.int32))
    pfl[0, 0, 0, :4] = 15
    mean = torch.from_numpy(rng.standard_normal((n, 8)).astype(np.float32) * 0.3).to(dev())
    rstd = torch.from_numpy((0.5 + rng.random((n, 8))).astype(np.float32)).to(dev())
    prd, pfd = praw.to(dev()), pfl.to(dev())
    dense = torch.empty((n, d, h, w, 8), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pool_finish(_lib.ptr(prd), _lib.ptr(pfd), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(dense), 8, n, d * h * w,
                                  1 if pmode else 0, _lib.stream_ptr()), "vx_pool_finish")
    wt, b, wp, bd = _pack16(8, 722)
    o_sep, s_sep, _, kn = _zc16_launch(dense, 8, wp, bd, n, d, h, w, stats=True)
    assert kn == "conv3d_zc16_kernel<8,0,0,0,0>", kn
    o_fus, s_fus, _, kn = _zc16_launch(prd, 8, wp, bd, n, d, h, w, stats=True,
                                       poolfin=(mean, rstd, pfd, _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE))
    assert kn == "conv3d_zc16_kernel<8,0,3,0,0>", kn
    assert torch.isfinite(o_fus).all()
    assert torch.equal(o_sep, o_fus) and torch.equal(s_sep, s_fus)
    ref = F.conv3d(ncdhw(dense).cpu().double(), wt.double(), b.double(), padding=1)
    assert (ncdhw(o_fus).cpu().double() - ref).abs().max().item() < 4e-5


@pytest.mark.parametrize("shape,cout", [((2, 8, 16, 32), 16), ((1, 5, 12, 20), 16), ((2, 4, 8, 8), 32)])
@pytest.mark.parametrize("pmode", [1, 0])
def test_conv3d_tile_kernel_prologue_on_the_skip_half_of_a_concat_input(shape, cout, pmode, vxcfg):
    """Round 5 (expand_2_1 reading contr_2_2's RAW output from the x-blocked concat buffer, unet3D_module.py:231-237, 332-356):
    the tile kernel normalises ONLY the chunks of the skip half on load -- statistics of the skip tensor, the dropout bits of its
    dense element space recovered from the x-blocked load offset -- and leaves the up half as it is.  Ragged tiles, 32 and 64
    concatenated channels."""
    lib = _lib.load()
    n, d, h, w = shape
    cs = cout                                   # channels per half (the decoder's first conv: 2 C -> C)
    cin = 2 * cs
    assert lib.vx_conv3d_k3_skip_prologue_ok(d, h, w, cin, cout, 4) == 1
    up = torch.from_numpy(formula_tensor((n, cs, d, h, w), 731, scale=1.2)).float()
    raw = (torch.from_numpy(formula_tensor((n, cs, d, h, w), 732, scale=2.0)) + 0.3).float()
    wt = torch.from_numpy(formula_tensor((cout, cin, 3, 3, 3), 733, scale=(1.0 / (27 * cin)) ** 0.5)).float().contiguous()
    b = torch.from_numpy(formula_tensor((cout,), 734, scale=0.2)).float().contiguous()
    mean = raw.double().mean((2, 3, 4)).float().contiguous().to(dev())
    rstd = (1.0 / torch.sqrt(raw.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
    keep = _hash_mask(61, 3, n, cs, d, h, w) if pmode else torch.ones((n, cs, d, h, w), dtype=torch.float64)
    skip = F.leaky_relu((raw.double() - mean.cpu().double().view(n, cs, 1, 1, 1)) * rstd.cpu().double().view(n, cs, 1, 1, 1), 0.01)
    skip = skip * keep * (2.0 if pmode else 1.0)
    ref = F.leaky_relu(F.conv3d(torch.cat([up.double(), skip], 1), wt.double(), b.double(), padding=1), 0.01)
    cat = torch.empty((n, d, h, w // 4, 2, 4, cs), dtype=torch.float32)
    cat[:, :, :, :, 0] = cl(up).view(n, d, h, w // 4, 4, cs)
    cat[:, :, :, :, 1] = cl(raw).view(n, d, h, w // 4, 4, cs)
    catd = cat.to(dev())
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wt.to(dev())), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "pack")
    bd = b.to(dev())
    out = torch.full((n, d, h, w, cout), -77.0, dtype=torch.float32, device=dev())
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, cout)
    a.in_ = catd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = cin, cout, 0
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, cout
    a.act, a.in_xblk = _lib.VX_ACT_LRELU, 4
    a.in_mean, a.in_rstd, a.in_repeat = mean.data_ptr(), rstd.data_ptr(), 1
    a.in_drop_mode, a.in_drop_seed, a.in_drop_layer = (_lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE), 61, 3
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    torch.cuda.synchronize()
    assert lib.vx_last_kernel_name().decode().startswith("conv3d_k3_s16_kernel")
    err = (ncdhw(out).cpu().double() - ref).abs().max().item()
    assert err < 4e-5, err
    # the same launch with the hash-dropout epilogue of the decoder (the compile-time instance the network runs)
    a.drop_mode, a.drop_seed, a.drop_layer = _lib.VX_DROP_HASH, 9, 11
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    torch.cuda.synchronize()
    want = ref * _hash_mask(9, 11, n, cout, d, h, w) * 2.0
    assert (ncdhw(out).cpu().double() - want).abs().max().item() < 8e-5


@pytest.mark.parametrize("shape", [(2, 8, 16, 32), (1, 4, 24, 64)])
@pytest.mark.parametrize("pmode", [1, 0])
def test_conv3d_zc16_partial_sums_two_launches_are_the_conv_over_the_concatenation(shape, pmode, vxcfg):
    """expand_2_1 without a concatenated tensor (vx_conv3d_args.acc_in): conv(cat([up, skip])) = conv_skip(normalised skip) + bias
    -> partial sums, then conv_up(up) + partial -> LeakyReLU -> dropout, IN PLACE -- against the float64 conv over the 32
    concatenated channels (unet3D_module.py:332-356, 263-267)."""
    lib = _lib.load()
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_acc_ok(d, h, w, 16, 16) == 1 and lib.vx_conv3d_k3_acc_ok(d, h, 16, 16, 16) == 0
    up = torch.from_numpy(formula_tensor((n, 16, d, h, w), 741, scale=1.2)).float()
    raw = (torch.from_numpy(formula_tensor((n, 16, d, h, w), 742, scale=2.0)) + 0.3).float()
    wt = torch.from_numpy(formula_tensor((16, 32, 3, 3, 3), 743, scale=(1.0 / (27 * 32)) ** 0.5)).float().contiguous()
    b = torch.from_numpy(formula_tensor((16,), 744, scale=0.2)).float().contiguous()
    mean = raw.double().mean((2, 3, 4)).float().contiguous().to(dev())
    rstd = (1.0 / torch.sqrt(raw.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
    keep_in = _hash_mask(61, 3, n, 16, d, h, w) if pmode else torch.ones((n, 16, d, h, w), dtype=torch.float64)
    skip = F.leaky_relu((raw.double() - mean.cpu().double().view(n, 16, 1, 1, 1)) * rstd.cpu().double().view(n, 16, 1, 1, 1), 0.01)
    skip = skip * keep_in * (2.0 if pmode else 1.0)
    ref = F.leaky_relu(F.conv3d(torch.cat([up.double(), skip], 1), wt.double(), b.double(), padding=1), 0.01)
    if pmode:
        ref = ref * _hash_mask(9, 14, n, 16, d, h, w) * 2.0
    bd = b.to(dev())
    packed = []
    for half in range(2):
        part = wt[:, 16 * half:16 * half + 16].contiguous().to(dev())
        pk = torch.empty(lib.vx_conv3d_k3_packed_floats(16, 16), dtype=torch.float32, device=dev())
        _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(part), _lib.ptr(pk), 16, 16, _lib.stream_ptr()), "pack")
        packed.append(pk)
    upd, rawd = cl(up).to(dev()), cl(raw).to(dev())
    out = torch.full((n, d, h, w, 16), -77.0, dtype=torch.float32, device=dev())

    def args(x, pk):
        a = _lib.ConvArgs()
        a.w_family = lib.vx_conv3d_k3_family(16, 16)
        a.in_ = x.data_ptr(); a.w_packed = pk.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
        a.in_pitch, a.out_pitch, a.out_coff = 16, 16, 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, 16, 16
        return a
    a1 = args(rawd, packed[1])
    a1.in_mean, a1.in_rstd, a1.in_repeat = mean.data_ptr(), rstd.data_ptr(), 1
    a1.in_drop_mode, a1.in_drop_seed, a1.in_drop_layer = (_lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE), 61, 3
    _lib.check(lib.vx_conv3d_k3(C.byref(a1), _lib.stream_ptr()), "vx_conv3d_k3 (skip half)")
    assert lib.vx_last_kernel_name().decode() == "conv3d_zc16_kernel<16,3,1,0,0>"
    a2 = args(upd, packed[0])
    a2.act = _lib.VX_ACT_LRELU
    if pmode:
        a2.drop_mode, a2.drop_seed, a2.drop_layer = _lib.VX_DROP_HASH, 9, 14
    a2.acc_in, a2.acc_pitch = out.data_ptr(), 16
    _lib.check(lib.vx_conv3d_k3(C.byref(a2), _lib.stream_ptr()), "vx_conv3d_k3 (up half)")
    torch.cuda.synchronize()
    assert lib.vx_last_kernel_name().decode() == ("conv3d_zc16_kernel<16,1,0,1,0>" if pmode else "conv3d_zc16_kernel<16,3,0,1,0>")
    err = (ncdhw(out).cpu().double() - ref).abs().max().item()
    assert err < 8e-5, err
    # refused where the z-column kernel does not run
    vxcfg.set(s16_no_zc16=1)
    with pytest.raises(_lib.VxError):
        _lib.check(lib.vx_conv3d_k3(C.byref(a2), _lib.stream_ptr()), "vx_conv3d_k3")


@pytest.mark.parametrize("shape", [(2, 8, 16, 32), (1, 4, 24, 64)])
@pytest.mark.parametrize("split", [0, 1])
def test_conv3d_zc16_fused_upconvolution_matches_oracle(shape, split, vxcfg):
    """upscale3 inside expand_2_1's up-half launch (vx_conv3d_k3_upfuse_ok == 2): the 16 input channels are
    ConvTranspose3d(32 -> 16, k = 2, s = 2) of the coarse tensor, evaluated by the staging waves (plain and pre-split coarse
    tensor), with and without partial sums, LeakyReLU + hash dropout -- against the float64 oracle
    (unet3D_module.py:157-190, 263-267, 332-356); every border class of the coarse window (first / last column tile in x and y,
    first / last step in z)."""
    lib = _lib.load()
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_upfuse_ok(d, h, w, 16, 16) == 2
    coarse = torch.from_numpy(formula_tensor((n, 32, d // 2, h // 2, w // 2), 751, scale=1.3)).float()
    uw = torch.from_numpy(formula_tensor((32, 16, 2, 2, 2), 752, scale=(1.0 / 32) ** 0.5)).float().contiguous()
    ub = torch.from_numpy(formula_tensor((16,), 753, scale=0.3)).float().contiguous()
    wt, b, wp, bd = _pack16(16, 754)
    part = torch.from_numpy(formula_tensor((n, 16, d, h, w), 755, scale=0.7)).float()
    up = F.conv_transpose3d(coarse.double(), uw.double(), ub.double(), stride=2)
    conv = F.conv3d(up, wt.double(), None, padding=1)
    keep = _hash_mask(9, 14, n, 16, d, h, w)
    uwp = torch.empty(lib.vx_convT_zc16_packed_floats(), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_convT_zc16(_lib.ptr(uw.to(dev())), _lib.ptr(uwp), _lib.stream_ptr()), "vx_pack_convT_zc16")
    ubd = ub.to(dev())
    cd = cl(coarse).to(dev())
    if split:
        v = cd.reshape(-1, 4)
        hi = v.half()
        lo = ((v - hi.float()) * 2048.0).half()
        cd = torch.cat([hi, lo], 1).view(torch.float32).reshape(cd.shape).contiguous()
    for acc in (0, 1):
        out = cl(part).to(dev()).contiguous() if acc else torch.full((n, d, h, w, 16), -77.0, dtype=torch.float32, device=dev())
        a = _lib.ConvArgs()
        a.w_family = lib.vx_conv3d_k3_family(16, 16)
        a.in_ = cd.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
        a.in_pitch, a.out_pitch, a.out_coff = 16, 16, 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, 16, 16
        a.act, a.drop_mode, a.drop_seed, a.drop_layer = _lib.VX_ACT_LRELU, _lib.VX_DROP_HASH, 9, 14
        a.up_in, a.up_w, a.up_b, a.up_pitch, a.up_split = cd.data_ptr(), uwp.data_ptr(), ubd.data_ptr(), 32, split
        if acc:
            a.acc_in, a.acc_pitch = out.data_ptr(), 16
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3 (fused up-convolution)")
        torch.cuda.synchronize()
        assert lib.vx_last_kernel_name().decode() == "conv3d_zc16_kernel<16,1,0,%d,1>" % acc
        pre_act = conv + (part.double() if acc else b.double().view(1, 16, 1, 1, 1))
        ref = F.leaky_relu(pre_act, 0.01) * keep * 2.0
        dlt = (ncdhw(out).cpu().double() - ref).abs()
        err = dlt.max().item()
        where = np.unravel_index(int(dlt.argmax()), dlt.shape)
        bad_z = (dlt.amax((0, 1, 3, 4)) > 8e-5).nonzero().flatten().tolist()
        bad_y = (dlt.amax((0, 1, 2, 4)) > 8e-5).nonzero().flatten().tolist()
        bad_x = (dlt.amax((0, 1, 2, 3)) > 8e-5).nonzero().flatten().tolist()
        assert err < 8e-5, (acc, err, where, "z", bad_z, "y", bad_y, "x", bad_x)
    vxcfg.set(s16_no_upfuse=1)
    assert lib.vx_conv3d_k3_upfuse_ok(d, h, w, 16, 16) == 0
    with pytest.raises(_lib.VxError):
        _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: the role-split kernel of the deep layers (conv3d_deep.hip: Cout % 32 == 0, volumes of 32^3 and below)
def _deep_launch(x_dev, cin, cout, wp, bd, n, d, h, w, *, act=0, drop=0, seed=0, layer=0, stats=False, pre=None, in_xblk=0,
                 out_split=False, out_pitch=None, out_coff=0):
    """one vx_conv3d_k3 launch on a dense channels-last (or x-blocked concat) device input.
    Returns (out device tensor, stats or None, kernel name)"""
    lib = _lib.load()
    a = _lib.ConvArgs()
    a.w_family = lib.vx_conv3d_k3_family(cin, cout)
    op = out_pitch or cout
    out = torch.full((n, d, h, w, op), -77.0, dtype=torch.float32, device=dev())
    a.in_ = x_dev.data_ptr(); a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr(); a.out = out.data_ptr()
    a.in_pitch, a.out_pitch, a.out_coff = cin, op, out_coff
    a.in_xblk = in_xblk
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, cout
    a.act, a.drop_mode, a.drop_seed, a.drop_layer = act, drop, seed, layer
    a.out_half, a.out_split = 1, 1 if out_split else 0
    st = None
    if stats:
        nt = lib.vx_conv3d_k3_tiles_for(d, h, w, cout)
        st = torch.full((n, nt, cout, 2), 5.0, dtype=torch.float32, device=dev())
        a.stats_partial = st.data_ptr()
    if pre is not None:
        a.in_mean, a.in_rstd, a.in_repeat = pre[0].data_ptr(), pre[1].data_ptr(), 1
        a.in_drop_mode, a.in_drop_seed, a.in_drop_layer = pre[2], pre[3], pre[4]
    _lib.check(lib.vx_conv3d_k3(C.byref(a), _lib.stream_ptr()), "vx_conv3d_k3")
    torch.cuda.synchronize()
    return out, st, lib.vx_last_kernel_name().decode()


def _pack_deep(cin, cout, seed):
    lib = _lib.load()
    wt = torch.from_numpy(formula_tensor((cout, cin, 3, 3, 3), seed, scale=(1.0 / (27 * cin)) ** 0.5)).float().contiguous()
    b = torch.from_numpy(formula_tensor((cout,), seed + 1, scale=0.2)).float().contiguous()
    wp = torch.empty(lib.vx_conv3d_k3_packed_floats(cin, cout), dtype=torch.float32, device=dev())
    _lib.check(lib.vx_pack_conv3d_k3(_lib.ptr(wt.to(dev())), _lib.ptr(wp), cin, cout, _lib.stream_ptr()), "pack")
    return wt, b, wp, b.to(dev())


# (Cin, Cout, (N, D, H, W)): the 64^3 network's deep layers (16^3 / 8^3 / 4^3), several tiles per sample in every direction,
# four 4^3 samples per tile, non-cubic volumes, more tiles than workgroups' first round
DEEP_CASES = [(16, 32, (2, 16, 16, 16)), (32, 32, (1, 8, 16, 32)), (32, 64, (3, 8, 8, 8)), (64, 64, (2, 8, 8, 8)),
              (64, 128, (8, 4, 4, 4)), (128, 128, (4, 4, 4, 4)), (24, 32, (1, 4, 8, 16)), (32, 32, (2, 12, 8, 8)),
              (16, 64, (2, 8, 16, 16)), (32, 32, (36, 16, 16, 16)), (64, 128, (84, 4, 4, 4)),      # (the last two: 288 / 336 tiles on 256 workgroups)
              (64, 128, (5, 4, 4, 4)), (128, 128, (1, 4, 4, 4)), (64, 128, (7, 4, 4, 4))]          # round 6: batches that are no multiple of the four-sample tile


@pytest.mark.parametrize("cin,cout,shape", DEEP_CASES)
def test_conv3d_deep_plain_and_activation_epilogues_match_oracle(cin, cout, shape, vxcfg):
    """conv3d_deep.hip (round 5; unet3D_module.py:81-120, 231-243, 263-267 at the 16^3 / 8^3 / 4^3 levels): bias + statistics,
    LeakyReLU / ReLU, LeakyReLU + hash dropout (+ the pre-split hand-over, + an output pitch wider than Cout) against the
    float64 oracle; the knob that switches the kernel off gives the tile kernel's numbers."""
    lib = _lib.load()
    n, d, h, w = shape
    assert lib.vx_conv3d_k3_family(cin, cout) == 7
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 801, scale=1.5)).float()
    wt, b, wp, bd = _pack_deep(cin, cout, 802)
    xd = cl(x).to(dev())
    ref = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    tol = 4e-5
    # --- bias + statistics (EPI 0)
    out, st, kn = _deep_launch(xd, cin, cout, wp, bd, n, d, h, w, stats=True)
    multi = d * h * w < 256                  # several samples per tile: per-sample statistics stay with the tile kernel
    assert (kn.startswith("conv3d_k3_s16_kernel") if multi else kn.startswith("conv3d_deep_kernel<") and kn.endswith(",0,0>")), kn
    got = ncdhw(out).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err < tol, err
    ssum = st.double().sum(1).cpu()
    np.testing.assert_allclose(ssum[..., 0].numpy(), got.double().sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(ssum[..., 1].numpy(), (got.double() ** 2).sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    # --- run-time activation (EPI 3), into a wider output at a channel offset
    for act, fn in ((_lib.VX_ACT_LRELU, lambda t: F.leaky_relu(t, 0.01)), (_lib.VX_ACT_RELU, F.relu), (0, lambda t: t)):
        o2, _, kn = _deep_launch(xd, cin, cout, wp, bd, n, d, h, w, act=act, out_pitch=cout + 8, out_coff=4)
        assert kn.startswith("conv3d_deep_kernel<") and kn.endswith(",3,0>"), kn
        assert (ncdhw(o2[..., 4:4 + cout]).cpu().double() - fn(ref)).abs().max().item() < tol
        assert (o2[..., :4] == -77.0).all() and (o2[..., 4 + cout:] == -77.0).all()
    # --- LeakyReLU + hash dropout (EPI 1), plain / pre-split
    keep = _hash_mask(77, 13, n, cout, d, h, w)
    want = F.leaky_relu(ref, 0.01) * keep * 2.0
    o3, _, kn = _deep_launch(xd, cin, cout, wp, bd, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=77, layer=13)
    assert kn.startswith("conv3d_deep_kernel<") and kn.endswith(",1,0>"), kn
    assert (ncdhw(o3).cpu().double() - want).abs().max().item() < tol
    o4, _, _ = _deep_launch(xd, cin, cout, wp, bd, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=77, layer=13, out_split=True)
    hl = o4.cpu().contiguous().view(torch.float16).view(n, d, h, w, cout // 4, 2, 4).float()      # [quad][hi | lo][4]
    back = (hl[..., 0, :] + hl[..., 1, :] / 2048.0).reshape(n, d, h, w, cout)
    assert (back - o3.cpu()).abs().max().item() <= 1e-6 * max(1.0, o3.abs().max().item())
    # --- the knob: the tile kernel on the same launch
    vxcfg.set(s16_no_deep=1)
    o6, st6, kn = _deep_launch(xd, cin, cout, wp, bd, n, d, h, w, stats=True)
    assert kn.startswith("conv3d_k3_s16_kernel"), kn
    assert (ncdhw(o6).cpu().double() - ref).abs().max().item() < tol
    assert (o6 - out).abs().max().item() < 2e-5
    np.testing.assert_allclose(st6.double().sum(1).cpu().numpy(), ssum.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("cin,cout,shape,n_big", [(64, 128, (4, 4, 4), 7), (128, 128, (4, 4, 4), 9), (32, 64, (8, 8, 8), 70), (32, 32, (16, 16, 16), 5)])
def test_conv3d_deep_is_batch_independent(cin, cout, shape, n_big, vxcfg):
    """Round-5 advice: the tile of conv3d_deep.hip is a function of the volume's shape and the channel count only -- never of the
    batch size -- and the last four-sample tile of a 4^3 batch may hold fewer samples (masked loads and stores): a sample's output
    and its statistics partials are the same BITS alone and among n_big - 1 batch mates (R = 2 at small batches gave other fp32 tile
    sums, N % 4 != 0 sent the 4^3 layers to another kernel)."""
    d, h, w = shape
    x = torch.from_numpy(formula_tensor((n_big, cin, d, h, w), 811, scale=1.5)).float()
    wt, b, wp, bd = _pack_deep(cin, cout, 812)
    xd = cl(x).to(dev())
    stats = d * h * w >= 256
    big, stb, knb = _deep_launch(xd, cin, cout, wp, bd, n_big, d, h, w, act=_lib.VX_ACT_LRELU)
    assert knb.startswith("conv3d_deep_kernel<"), knb
    for i in (0, n_big // 2, n_big - 1):
        one, _, kn1 = _deep_launch(xd[i:i + 1].contiguous(), cin, cout, wp, bd, 1, d, h, w, act=_lib.VX_ACT_LRELU)
        assert kn1 == knb, (kn1, knb)
        assert torch.equal(one[0], big[i]), i
    if stats:
        bigs, stb, knb = _deep_launch(xd, cin, cout, wp, bd, n_big, d, h, w, stats=True)
        one, st1, kn1 = _deep_launch(xd[n_big - 1:].contiguous(), cin, cout, wp, bd, 1, d, h, w, stats=True)
        assert kn1 == knb and knb.startswith("conv3d_deep_kernel<"), (kn1, knb)
        assert torch.equal(one[0], bigs[n_big - 1]) and torch.equal(st1[0], stb[n_big - 1])
    ref = F.leaky_relu(F.conv3d(x.double(), wt.double(), b.double(), padding=1), 0.01)
    assert (ncdhw(big).cpu().double() - ref).abs().max().item() < 4e-5


@pytest.mark.parametrize("cin,cout,shape", [(32, 64, (1, 6, 8, 16)), (64, 64, (2, 6, 8, 32)), (32, 32, (1, 10, 16, 16))])
def test_conv3d_deep_statistics_fall_back_where_its_tiles_do_not_divide_the_partials(cin, cout, shape, vxcfg):
    """Round-5 advice: volumes with D % 4 == 2 at the deep levels (48 x 64 x 128 -> 6 x 8 x 16 at level 3): the deep kernel's small tile
    gives 3 tiles per sample against a partials buffer of 4 entries -- the launch used to FAIL (VX_E_SHAPE); now the general tile
    kernel takes it.  Statistics and output against the oracle either way."""
    lib = _lib.load()
    n, d, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, d, h, w), 821, scale=1.5)).float()
    wt, b, wp, bd = _pack_deep(cin, cout, 822)
    xd = cl(x).to(dev())
    ref = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    out, st, kn = _deep_launch(xd, cin, cout, wp, bd, n, d, h, w, stats=True)
    got = ncdhw(out).cpu()
    assert (got.double() - ref).abs().max().item() < 4e-5, kn
    ssum = st.double().sum(1).cpu()
    np.testing.assert_allclose(ssum[..., 0].numpy(), got.double().sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(ssum[..., 1].numpy(), (got.double() ** 2).sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    # without statistics the deep kernel still runs these shapes
    o2, _, kn2 = _deep_launch(xd, cin, cout, wp, bd, n, d, h, w, act=_lib.VX_ACT_RELU)
    assert (ncdhw(o2).cpu().double() - F.relu(ref)).abs().max().item() < 4e-5, kn2


@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (2, 16, 16, 16)), (64, 64, (3, 8, 8, 8)), (16, 32, (1, 8, 8, 16)), (16, 32, (33, 16, 16, 16))])
@pytest.mark.parametrize("pmode", [1, 0])
def test_conv3d_deep_prologue_matches_oracle(cin, cout, shape, pmode, vxcfg):
    """The second conv of a contract block at the deep levels (contr_3_2 / contr_4_2): normalise-on-load of the raw first conv
    (InstanceNorm + LeakyReLU + hash dropout in the staging waves) + its own statistics (unet3D_module.py:231-237)."""
    n, d, h, w = shape
    raw = (torch.from_numpy(formula_tensor((n, cin, d, h, w), 811, scale=2.0)) + 0.3).float()
    wt, b, wp, bd = _pack_deep(cin, cout, 812)
    mean = raw.double().mean((2, 3, 4)).float().contiguous().to(dev())
    rstd = (1.0 / torch.sqrt(raw.double().var((2, 3, 4), unbiased=False) + 1e-5)).float().contiguous().to(dev())
    keep_in = _hash_mask(61, 4, n, cin, d, h, w) if pmode else torch.ones((n, cin, d, h, w), dtype=torch.float64)
    xin = F.leaky_relu((raw.double() - mean.cpu().double().view(n, cin, 1, 1, 1)) * rstd.cpu().double().view(n, cin, 1, 1, 1), 0.01)
    xin = xin * keep_in * (2.0 if pmode else 1.0)
    ref = F.conv3d(xin, wt.double(), b.double(), padding=1)
    pre = (mean, rstd, _lib.VX_DROP_HASH if pmode else _lib.VX_DROP_NONE, 61, 4)
    out, st, kn = _deep_launch(cl(raw).to(dev()), cin, cout, wp, bd, n, d, h, w, stats=True, pre=pre)
    assert kn.startswith("conv3d_deep_kernel<") and kn.endswith(",0,1>"), kn
    got = ncdhw(out).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err < 6e-5, err
    ssum = st.double().sum(1).cpu()
    np.testing.assert_allclose(ssum[..., 0].numpy(), got.double().sum((2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)
    vxcfg.set(s16_no_deep=1)
    o2, _, kn = _deep_launch(cl(raw).to(dev()), cin, cout, wp, bd, n, d, h, w, stats=True, pre=pre)
    assert kn.startswith("conv3d_k3_s16_kernel"), kn
    assert (o2 - out).abs().max().item() < 3e-5


@pytest.mark.parametrize("csrc,cout,shape,xb", [(32, 32, (2, 16, 16, 16), 4), (64, 64, (2, 8, 8, 8), 4), (16, 32, (1, 8, 8, 16), 2),
                                                 (128, 128, (4, 4, 4, 4), 2)])
def test_conv3d_deep_reads_the_x_blocked_concat_buffer(csrc, cout, shape, xb, vxcfg):
    """The first conv of an expand block (expand_4_1 / expand_3_1) reads cat([up, skip], 1) from the x-blocked buffer
    [N][D][H][W/xb][2][xb][C] (unet3D_module.py:263-267, 332-356): chunks of the up half, then of the skip half."""
    n, d, h, w = shape
    cin = 2 * csrc
    up = torch.from_numpy(formula_tensor((n, csrc, d, h, w), 821, scale=1.2)).float()
    skip = torch.from_numpy(formula_tensor((n, csrc, d, h, w), 822, scale=0.9)).float()
    wt, b, wp, bd = _pack_deep(cin, cout, 823)
    ref = F.leaky_relu(F.conv3d(torch.cat([up, skip], 1).double(), wt.double(), b.double(), padding=1), 0.01)
    keep = _hash_mask(5, 9, n, cout, d, h, w)
    buf = to_xblk(up, skip, xb).to(dev())
    out, _, kn = _deep_launch(buf, cin, cout, wp, bd, n, d, h, w, act=_lib.VX_ACT_LRELU, drop=_lib.VX_DROP_HASH, seed=5, layer=9, in_xblk=xb)
    assert kn.startswith("conv3d_deep_kernel<") and kn.endswith(",1,0>"), kn
    err = (ncdhw(out).cpu().double() - ref * keep * 2.0).abs().max().item()
    assert err < 6e-5, err


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: streaming passes that reduce the producing conv's statistics partials themselves (vx_stat_src)
def _stat_src(partials, tiles, count, mean_out, rstd_out):
    st = _lib.StatSrc()
    st.stats_partial = partials.data_ptr(); st.tiles = tiles; st.eps = 1e-5; st.count = count
    st.mean_out = mean_out.data_ptr(); st.rstd_out = rstd_out.data_ptr()
    return st


def _finalize(partials, n, tiles, c, count):
    lib = _lib.load()
    mean = torch.empty((n, c), dtype=torch.float32, device=dev()); rstd = torch.empty_like(mean)
    _lib.check(lib.vx_instnorm_finalize(_lib.ptr(partials), n, tiles, c, count, 1e-5, _lib.ptr(mean), _lib.ptr(rstd), _lib.stream_ptr()), "finalize")
    return mean, rstd


@pytest.mark.parametrize("c,shape,tiles,pool", [(32, (3, 8, 8, 16), 16, True), (64, (2, 4, 8, 8), 2, True), (16, (2, 4, 4, 8), 5, False),
                                                 (256, (1, 2, 4, 4), 3, True), (8, (2, 4, 8, 32), 300, True)])
def test_norm_pass_with_its_own_statistics_equals_finalize_then_pass(c, shape, tiles, pool):
    """vx_norm_act_drop_pool_stats (the normalise + pool passes of the deep levels, unet3D_module.py:231-237, 314-323) against
    vx_instnorm_finalize + vx_norm_act_drop_pool on the same partials: mean / rstd to one ulp, outputs to float32 rounding."""
    lib = _lib.load()
    n, d, h, w = shape
    x = cl(torch.from_numpy(formula_tensor((n, c, d, h, w), 901, scale=2.0)).float()).to(dev())
    count = d * h * w
    # partials of a made-up tiling whose sums are the tensor's
    part = torch.from_numpy(formula_tensor((n, tiles, c, 2), 902, scale=3.0)).float()
    tot = torch.stack([x.cpu().double().sum((1, 2, 3)), (x.cpu().double() ** 2).sum((1, 2, 3))], -1)
    part[:, 0] += (tot - part.double().sum(1)).float()
    part = part.contiguous().to(dev())
    mean, rstd = _finalize(part, n, tiles, c, count)

    def run(stats):
        a = _lib.NormArgs()
        out = torch.full((n, d, h, w, c), -7.0, dtype=torch.float32, device=dev())
        pl = torch.full((n, d // 2, h // 2, w // 2, c), -7.0, dtype=torch.float32, device=dev())
        a.x = x.data_ptr(); a.x_pitch = c; a.out = out.data_ptr(); a.out_pitch = c
        a.N, a.D, a.H, a.W, a.C = n, d, h, w, c
        a.act, a.drop_mode, a.drop_seed, a.drop_layer = _lib.VX_ACT_LRELU, _lib.VX_DROP_HASH, 5, 3
        if pool:
            a.pool_out = pl.data_ptr(); a.pool_pitch = c
        if stats:
            mo = torch.zeros((n, c), dtype=torch.float32, device=dev()); ro = torch.zeros_like(mo)
            st = _stat_src(part, tiles, count, mo, ro)
            _lib.check(lib.vx_norm_act_drop_pool_stats(C.byref(a), C.byref(st), _lib.stream_ptr()), "norm_stats")
            torch.cuda.synchronize()
            return out, pl, mo, ro, lib.vx_last_kernel_name().decode()
        a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
        _lib.check(lib.vx_norm_act_drop_pool(C.byref(a), _lib.stream_ptr()), "norm")
        torch.cuda.synchronize()
        return out, pl, mean, rstd, lib.vx_last_kernel_name().decode()

    o1, p1, m1, r1, k1 = run(True)
    o0, p0, m0, r0, k0 = run(False)
    assert k1.count(",") == 2 and k1.endswith(",true>") and k0.count(",") == 1, (k1, k0)      # <POOL, WIDE, FOLD> / <POOL, WIDE>
    assert ((m1 - m0).abs() <= 1.2e-7 * (1.0 + m0.abs())).all() and ((r1 - r0).abs() <= 1.2e-7 * r0.abs()).all()
    assert (o1 - o0).abs().max().item() <= 2e-6 * max(1.0, o0.abs().max().item())
    if pool:
        assert (p1 - p0).abs().max().item() <= 2e-6 * max(1.0, p0.abs().max().item())


def test_prenorm_split_and_pool_finish_z_with_their_own_statistics():
    """vx_prenorm_split_stats / vx_pool_finish_z_stats against vx_instnorm_finalize followed by the plain entry points."""
    lib = _lib.load()
    # --- the once-per-volume pre-split of the first block
    n, nvox, tiles = 3, 4096, 37
    x = torch.from_numpy(formula_tensor((n, nvox, 8), 911, scale=2.0)).float().contiguous()
    part = torch.from_numpy(formula_tensor((n, tiles, 8, 2), 912, scale=3.0)).float()
    tot = torch.stack([x.double().sum(1), (x.double() ** 2).sum(1)], -1)
    part[:, 0] += (tot - part.double().sum(1)).float()
    part = part.contiguous().to(dev())
    mean, rstd = _finalize(part, n, tiles, 8, nvox)
    a0 = x.clone().to(dev()); a1 = x.clone().to(dev())
    _lib.check(lib.vx_prenorm_split(_lib.ptr(a0), _lib.ptr(mean), _lib.ptr(rstd), n, nvox, 2.0, _lib.stream_ptr()), "presplit")
    mo = torch.zeros((n, 8), dtype=torch.float32, device=dev()); ro = torch.zeros_like(mo)
    st = _stat_src(part, tiles, nvox, mo, ro)
    _lib.check(lib.vx_prenorm_split_stats(_lib.ptr(a1), C.byref(st), n, nvox, 2.0, _lib.stream_ptr()), "presplit_stats")
    torch.cuda.synchronize()
    assert lib.vx_last_kernel_name().decode() == "prenorm_split_kernel<true>"
    assert ((mo - mean).abs() <= 1.2e-7 * (1.0 + mean.abs())).all() and ((ro - rstd).abs() <= 1.2e-7 * rstd.abs()).all()
    def unsplit(t):
        hl = t.cpu().contiguous().view(torch.float16).view(n, nvox, 2, 2, 4).float()      # [quad][hi | lo][4]
        return hl[..., 0, :] + hl[..., 1, :] / 2048.0
    assert (unsplit(a1) - unsplit(a0)).abs().max().item() <= 2e-6 * max(1.0, unsplit(a0).abs().max().item())
    # --- the z pair + statistics of the 16-channel level's pooled tensor
    n, dp, pv, tiles = 2, 3, 20, 9
    raw = torch.from_numpy(formula_tensor((n, 2 * dp, pv, 16), 921, scale=2.0)).float().contiguous().to(dev())
    flags = (torch.from_numpy(formula_tensor((n, 2 * dp, pv, 4), 922)) > 0.5).to(torch.int32) * 5
    flags = flags.contiguous().to(dev())
    part = torch.from_numpy(formula_tensor((n, tiles, 16, 2), 923, scale=3.0)).float()
    part[..., 1] = part[..., 1].abs() * 40 + 50.0        # (a positive variance)
    part = part.contiguous().to(dev())
    count = 8 * 2 * dp * pv
    mean, rstd = _finalize(part, n, tiles, 16, count)
    o0 = torch.empty((n, dp, pv, 16), dtype=torch.float32, device=dev()); o1 = torch.empty_like(o0)
    _lib.check(lib.vx_pool_finish_z(_lib.ptr(raw), _lib.ptr(flags), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(o0), 16, n, dp, pv, 1, _lib.stream_ptr()), "pfz")
    mo = torch.zeros((n, 16), dtype=torch.float32, device=dev()); ro = torch.zeros_like(mo)
    st = _stat_src(part, tiles, count, mo, ro)
    _lib.check(lib.vx_pool_finish_z_stats(_lib.ptr(raw), _lib.ptr(flags), C.byref(st), _lib.ptr(o1), 16, n, dp, pv, 1, _lib.stream_ptr()), "pfz_stats")
    torch.cuda.synchronize()
    assert lib.vx_last_kernel_name().decode() == "pool_finish_z_kernel<true>"
    assert ((mo - mean).abs() <= 1.2e-7 * (1.0 + mean.abs())).all() and ((ro - rstd).abs() <= 1.2e-7 * rstd.abs()).all()
    assert (o1 - o0).abs().max().item() <= 2e-6 * max(1.0, o0.abs().max().item())


def test_fuzz_conv_fixed_seed_slice():
    """tools/fuzz_conv.py for a fixed seed, 60 cases (round-5 verdict: so that the driver's box, not only the builder's, runs it):
    random (Cin, Cout, N, D, H, W, epilogue) through vx_conv3d_k3 -- the specialised instances against the generic split-fp16
    kernel bit for bit, the role-split kernels of the 16-channel / deep layers against it to float32 rounding (2e-5), everything
    against the native-fp32 kernels; batches that are no multiple of the deep kernel's four-sample tile included."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vx_fuzz_conv", os.path.join(root, "tools", "fuzz_conv.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    nzc, bad = mod.run_cases(60, 2026)
    assert bad == 0, bad
    assert nzc >= 8, nzc          # (the slice must reach conv3d_zc16.hip / conv3d_deep.hip)


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: the dropout bit generator (csrc/common.h: vx_drop_key / vx_drop_word) -- independence of the exported masks and the
# key space (round-5 verdict, items "missing 7" / "next 3"; the reference draws torch's Bernoulli stream, unet3D_module.py:236, 242, 266)
_M32 = np.uint64(0xFFFFFFFF)


def _np_mix32(h):
    h = h.astype(np.uint64) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7feb352d)) & _M32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846ca68b)) & _M32
    h ^= h >> np.uint64(16)
    return h


def _np_key(seed, layer, sample):
    """host restatement of vx_drop_key: the two key words of a (seed, layer, sample) stream"""
    s, l, n = (np.asarray(v, dtype=np.uint64) for v in (seed, layer, sample))
    a = _np_mix32((s * np.uint64(0x9E3779B1) + l * np.uint64(0x85EBCA6B) + n * np.uint64(0xC2B2AE35) + np.uint64(0x27D4EB2F)) & _M32)
    b = _np_mix32((s * np.uint64(0xC2B2AE3D) + l * np.uint64(0x27D4EB2F) + n * np.uint64(0x165667B1) + np.uint64(0x9E3779B9)) & _M32)
    return a, b


def _np_words(a, b, nwords, old=False):
    """keep-words 0 .. nwords - 1 of the stream keyed (a, b); old = the one-word construction of rounds 1-5 (vx_mix32(index ^ a))"""
    w = np.arange(nwords, dtype=np.uint64) ^ np.uint64(a)
    if old:
        return _np_mix32(w)
    w ^= w >> np.uint64(16)
    w = (w * np.uint64(0x7feb352d)) & _M32
    w = (w + np.uint64(b)) & _M32
    w ^= w >> np.uint64(15)
    w = (w * np.uint64(0x846ca68b)) & _M32
    w ^= w >> np.uint64(16)
    return w


def _device_words(seed, layer, n, elems):
    """the keep-words vx_drop_hash_mask exports for samples 0 .. n - 1 of (seed, layer): (n, elems / 32) uint64"""
    lib = _lib.load()
    m = torch.empty((n, elems), dtype=torch.uint8, device=dev())
    _lib.check(lib.vx_drop_hash_mask(int(seed) & 0xFFFFFFFF, layer, n, elems, _lib.ptr(m), _lib.stream_ptr()), "vx_drop_hash_mask")
    bits = m.view(n, elems // 32, 32).to(torch.int64)
    return (bits << torch.arange(32, device=dev())).sum(-1).cpu().numpy().astype(np.uint64)


def test_hash_dropout_key_space_no_stream_is_a_permutation_of_another():
    """Two streams whose first key words agree above the stream's word count were, until round 5, the SAME keep-words in XOR-permuted
    order.  The colliding pair is constructed on the host (a second seed whose key agrees with seed 123's in the upper 16 bits, layer
    1 = contr_1_2: 64^3 x 8 channels = 2^16 words), the old construction is shown to collide on it, and the device's streams for the
    very same pair share (next to) no word; the host restatement of the generator is pinned to the device's words on the way."""
    layer, elems = 1, 64 ** 3 * 8
    nw = elems // 32
    seeds = np.arange(1, 1 << 21, dtype=np.uint64)
    a0, b0 = _np_key(123, layer, 0)
    a, b = _np_key(seeds, layer, 0)
    hit = np.nonzero(((a ^ a0) < np.uint64(nw)) & (seeds != np.uint64(123)))[0]
    assert len(hit) >= 1
    s1 = int(seeds[hit[0]])
    a1, b1 = int(a[hit[0]]), int(b[hit[0]])
    # the old one-word construction on this pair: a permuted copy
    o0, o1 = _np_words(int(a0), 0, nw, old=True), _np_words(a1, 0, nw, old=True)
    assert np.array_equal(np.sort(o0), np.sort(o1)) and not np.array_equal(o0, o1)
    # the device's streams: the host restatement reproduces them, and they are unrelated
    d0 = _device_words(123, layer, 1, elems)[0]
    d1 = _device_words(s1, layer, 1, elems)[0]
    assert np.array_equal(d0, _np_words(int(a0), int(b0), nw)) and np.array_equal(d1, _np_words(a1, b1, nw))
    assert len(np.intersect1d(d0, d1)) <= 4          # 2^16 words out of 2^32: ~1 common value by chance
    assert not np.array_equal(np.sort(d0), np.sort(d1))
    # ... and no two of the 170 streams of one T = 10 volume (17 layers x 10 samples), nor those of the next seed, share their word sets
    # at the full-resolution layers (layers 0, 1, 15, 16: 2^16 words each)
    sets = {}
    for seed in (123, 124):
        for lay in (0, 1, 15, 16):
            w = _device_words(seed, lay, 10, elems)
            for t in range(10):
                sets[(seed, lay, t)] = np.sort(w[t])
    keys = list(sets)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            assert len(np.intersect1d(sets[keys[i]], sets[keys[j]], assume_unique=False)) <= 6, (keys[i], keys[j])


def test_hash_dropout_masks_are_independent_across_samples_layers_seeds_channels_and_neighbours():
    """The exported masks of a 64^3, T = 10 forward under seeds s and s + 1 (what bench.py feeds: consecutive seeds): keep rate per
    (layer, channel) and the correlations between sample pairs, layer pairs of one shape, consecutive seeds, channel pairs of a voxel
    and lag-1 neighbours along x / y / z, each within 4.5 sigma of a fair coin's (sigma = 1 / sqrt(N))."""
    from tests.test_gpu_unet3d import make_model
    model = make_model(do_dropout=True)
    T, S, seed = 10, 64, 123
    m0 = model.hash_dropout_masks(seed, T, S, S, S)
    m1 = model.hash_dropout_masks(seed + 1, T, S, S, S)
    sg = lambda m: m.float() * 2.0 - 1.0           # +-1
    worst = 0.0

    def chk(val, n, what):
        nonlocal worst
        z = abs(float(val)) * n ** 0.5
        worst = max(worst, z)
        assert z < 4.5, (what, float(val), n, z)

    for li, m in enumerate(m0):
        n_c = m.shape[0] * m.shape[2] * m.shape[3] * m.shape[4]
        rate = sg(m).mean((0, 2, 3, 4))
        assert (rate.abs() * n_c ** 0.5).max().item() < 4.5, (li, "keep rate per channel")
        s = sg(m)
        n_all = s[0].numel()
        for t in range(T):          # sample pairs
            for u in range(t + 1, T):
                chk((s[t] * s[u]).mean(), n_all, (li, "samples", t, u))
        chk((s * sg(m1[li])).mean(), s.numel(), (li, "consecutive seeds"))
        chk((s[:, :-1] * s[:, 1:]).mean(), s[:, 1:].numel(), (li, "neighbouring channels"))
        chk((s[:, 0] * s[:, s.shape[1] // 2]).mean(), s[:, 0].numel(), (li, "channels 0 / C/2"))
        chk((s[..., :-1] * s[..., 1:]).mean(), s[..., 1:].numel(), (li, "lag 1 in x"))
        chk((s[..., :-1, :] * s[..., 1:, :]).mean(), s[..., 1:, :].numel(), (li, "lag 1 in y"))
        chk((s[:, :, :-1] * s[:, :, 1:]).mean(), s[:, :, 1:].numel(), (li, "lag 1 in z"))
    for li in range(len(m0)):       # layer pairs of one shape
        for lj in range(li + 1, len(m0)):
            if m0[li].shape == m0[lj].shape:
                chk((sg(m0[li]) * sg(m0[lj])).mean(), m0[li].numel(), ("layers", li, lj))
    print(f"hash dropout: worst |z| over all checks = {worst:.2f}")
