"""GPU: the 2D (HRNet) kernels through the C ABI vs float64 torch-CPU restatements of the reference ops."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from values_amd import _lib
from tests.formula import formula_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def run_conv2d(x, w, b, ks, s, stats=True, out_pitch=None, out_coff=0, narrow=False):
    """narrow: the input arrives with pitch round4(cin) < Cin = round16(cin) -- the next pixel's channels follow directly,
    so a kernel that read "its" 16-channel block would multiply them in (vx_conv2d_args.in_pitch)"""
    lib = _lib.load()
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    cin_pad = (cin + 15) // 16 * 16
    in_pitch = (cin + 3) // 4 * 4 if narrow else cin_pad
    xd = torch.zeros((n, h, wd, in_pitch), dtype=torch.float32, device=DEV)
    xd[..., :cin] = nhwc(x.float()).to(DEV)
    wdv = w.float().contiguous().to(DEV)
    wp = torch.empty(lib.vx_conv2d_packed_floats(cin, cout, ks), dtype=torch.float32, device=DEV)
    _lib.check(lib.vx_pack_conv2d(_lib.ptr(wdv), _lib.ptr(wp), cin, cout, ks, _lib.stream_ptr()), "pack2d")
    oh = (h + 2 * (ks // 2) - ks) // s + 1
    ow = (wd + 2 * (ks // 2) - ks) // s + 1
    out_pitch = out_pitch or (cout + 3) // 4 * 4
    out = torch.full((n, oh, ow, out_pitch), -77.0, dtype=torch.float32, device=DEV)
    nt = lib.vx_conv2d_tiles(h, wd, ks, s)
    st = torch.zeros((n * nt, cout, 2), dtype=torch.float32, device=DEV)
    bd = b.float().to(DEV) if b is not None else None
    a = _lib.Conv2dArgs()
    a.w_family = lib.vx_conv2d_family(cin, cout, ks)
    a.in_ = xd.data_ptr(); a.in_pitch = in_pitch; a.w_packed = wp.data_ptr(); a.bias = bd.data_ptr() if bd is not None else None
    a.out = out.data_ptr(); a.out_pitch = out_pitch; a.out_coff = out_coff
    a.N, a.H, a.W, a.Cin, a.Cout, a.KS, a.S = n, h, wd, cin_pad, cout, ks, s
    if stats:
        a.stats_partial = st.data_ptr()
    _lib.check(lib.vx_conv2d(C.byref(a), _lib.stream_ptr()), "vx_conv2d")
    torch.cuda.synchronize()
    return nchw(out[..., out_coff:out_coff + cout]).cpu(), st.cpu(), out


@pytest.mark.parametrize("cin,cout,ks,s,shape", [
    (16, 16, 3, 1, (2, 16, 16)), (48, 48, 3, 1, (1, 20, 33)), (64, 64, 3, 1, (1, 8, 15)), (96, 96, 3, 1, (2, 16, 30)),
    (3, 64, 3, 2, (2, 64, 96)), (64, 64, 3, 2, (1, 32, 48)), (48, 96, 3, 2, (1, 17, 31)), (192, 384, 3, 2, (1, 16, 30)),
    (64, 64, 1, 1, (1, 16, 24)), (256, 64, 1, 1, (1, 16, 24)), (96, 48, 1, 1, (2, 8, 15)), (720, 720, 1, 1, (1, 8, 15)),
    (240, 4, 1, 1, (1, 16, 24)), (64, 24, 1, 1, (1, 7, 9)), (16, 128, 3, 1, (1, 5, 7)),
])
@pytest.mark.parametrize("mode", ["split16", "fp32"])
def test_conv2d_matches_oracle(cin, cout, ks, s, shape, mode, vxcfg):
    # default: split-fp16 schedule (conv2d_s16.hip); VX_CONV_FP32=1: native-fp32 kernels (conv2d_mfma.hip)
    if mode == "fp32":
        vxcfg.setenv("VX_CONV_FP32", "1")
    else:
        vxcfg.delenv("VX_CONV_FP32", raising=False)
    n, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, h, w), 201))
    wt = torch.from_numpy(formula_tensor((cout, cin, ks, ks), 202, scale=(1.0 / (ks * ks * cin)) ** 0.5))
    b = torch.from_numpy(formula_tensor((cout,), 203, scale=0.2)) if cout in (4, 24, 720) else None
    ref = F.conv2d(x.float().double(), wt.float().double(), None if b is None else b.float().double(), stride=s, padding=ks // 2)
    got, st, _ = run_conv2d(x, wt, b, ks, s)
    assert got.shape == ref.shape
    assert (got.double() - ref).abs().max().item() < 3e-5
    ssum = st.double().sum(0)
    np.testing.assert_allclose(ssum[:, 0].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(ssum[:, 1].numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("cin,cout,ks,s,shape", [
    (18, 18, 3, 1, (2, 20, 33)), (36, 36, 3, 1, (1, 16, 30)), (18, 36, 3, 2, (2, 17, 31)), (3, 64, 3, 2, (1, 32, 48)),
    (36, 72, 3, 2, (1, 16, 30)), (270, 19, 1, 1, (1, 9, 14)), (40, 48, 3, 1, (1, 8, 15)),
    (8, 16, 3, 1, (1, 16, 20)), (24, 24, 3, 2, (1, 16, 31)), (17, 18, 3, 1, (1, 9, 17)), (5, 36, 3, 2, (2, 33, 18)),
])
@pytest.mark.parametrize("mode", ["split16", "fp32"])
def test_conv2d_narrow_input_pitch_and_whole_cin_items(cin, cout, ks, s, shape, mode, vxcfg):
    """Round 3: (1) an input of C channels may arrive with pitch round4(C) -- channels at and beyond the pitch read as
    zeros (HRNet-W18's 18-channel branch at 20 floats per pixel); (2) 3x3 layers of <= 48 input channels run one work item
    per tile with all 16-channel sub-blocks staged together -- the same bits as one item per sub-block (c2s_no_wide)."""
    if mode == "fp32":
        vxcfg.setenv("VX_CONV_FP32", "1")
    else:
        vxcfg.delenv("VX_CONV_FP32", raising=False)
    n, h, w = shape
    x = torch.from_numpy(formula_tensor((n, cin, h, w), 221))
    wt = torch.from_numpy(formula_tensor((cout, cin, ks, ks), 222, scale=(1.0 / (ks * ks * cin)) ** 0.5))
    ref = F.conv2d(x.float().double(), wt.float().double(), None, stride=s, padding=ks // 2)
    got, st, _ = run_conv2d(x, wt, None, ks, s, narrow=True)
    assert (got.double() - ref).abs().max().item() < 3e-5
    ssum = st.double().sum(0)
    np.testing.assert_allclose(ssum[:, 0].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=2e-3)
    wide, _, _ = run_conv2d(x, wt, None, ks, s, narrow=False)
    assert torch.equal(got, wide)                     # the pitch changes the addressing, not a bit of the result
    if mode == "split16":
        vxcfg.set(c2s_no_wide=1)
        try:
            per_sub, st2, _ = run_conv2d(x, wt, None, ks, s, narrow=True)
        finally:
            vxcfg.set(c2s_no_wide=0)
        assert torch.equal(got, per_sub) and torch.equal(st, st2)
        # (3) 3x3 layers of <= 8 / 17..24 real input channels are packed for the octet-granular K schedule (3 / 7 steps of four
        # (tap, octet) units instead of 5 / 10 of two taps x a 16-channel block); c2s_no_oct: the sub-block packing -- another
        # summation order, the same oracle
        lib = _lib.load()
        oct_family = lib.vx_conv2d_family(cin, cout, ks) >= 100
        assert oct_family == (ks == 3 and (cin <= 8 or 16 < cin <= 24))
        vxcfg.set(c2s_no_oct=1)
        try:
            assert lib.vx_conv2d_family(cin, cout, ks) < 100
            sub_k, _, _ = run_conv2d(x, wt, None, ks, s, narrow=True)
        finally:
            vxcfg.set(c2s_no_oct=0)
        assert (sub_k.double() - ref).abs().max().item() < 3e-5
        if not oct_family:
            assert torch.equal(got, sub_k)


def test_conv2d_pitch_offset_into_concat():
    x = torch.from_numpy(formula_tensor((1, 32, 8, 16), 211))
    wt = torch.from_numpy(formula_tensor((32, 32, 3, 3), 212, scale=0.06))
    ref = F.conv2d(x.float().double(), wt.float().double(), None, padding=1)
    got, _, raw = run_conv2d(x, wt, None, 3, 1, out_pitch=80, out_coff=16)
    assert (got.double() - ref).abs().max().item() < 3e-5
    assert (raw[..., :16] == -77.0).all() and (raw[..., 48:] == -77.0).all()


def bn_scale_shift(x_nchw, gamma, beta):
    """through the real conv-epilogue-style partials + vx_bn_finalize"""
    lib = _lib.load()
    c = x_nchw.shape[1]
    xs = x_nchw.double()
    part = torch.stack([xs.sum((0, 2, 3)), (xs * xs).sum((0, 2, 3))], -1).float().reshape(1, c, 2).contiguous().to(DEV)
    scale = torch.empty(c, dtype=torch.float32, device=DEV)
    shift = torch.empty(c, dtype=torch.float32, device=DEV)
    g, b = gamma.float().to(DEV), beta.float().to(DEV)
    cnt = x_nchw.shape[0] * x_nchw.shape[2] * x_nchw.shape[3]
    _lib.check(lib.vx_bn_finalize(_lib.ptr(part), 1, c, cnt, 1e-5, _lib.ptr(g), _lib.ptr(b), _lib.ptr(scale), _lib.ptr(shift),
                                  _lib.stream_ptr()), "bn_finalize")
    return scale, shift


def run_affine(x, scale=None, shift=None, add=None, act=0, out_hw=None, mask=None, out_pitch=None, out_coff=0, inplace_add=False):
    lib = _lib.load()
    n, c, h, w = x.shape
    oh, ow = out_hw or (h, w)
    xd = nhwc(x.float()).to(DEV)
    out_pitch = out_pitch or c
    out = torch.full((n, oh, ow, out_pitch), -77.0, dtype=torch.float32, device=DEV)
    a = _lib.AffineArgs()
    a.x = xd.data_ptr(); a.x_pitch = c
    if scale is not None:
        a.scale = scale.data_ptr(); a.shift = shift.data_ptr()
    addd = None
    if add is not None:
        addd = nhwc(add.float()).to(DEV)
        if inplace_add:
            out[..., out_coff:out_coff + c] = addd
            a.add = out.data_ptr() + 4 * out_coff; a.add_pitch = out_pitch
        else:
            a.add = addd.data_ptr(); a.add_pitch = c
    a.out = out.data_ptr(); a.out_pitch = out_pitch; a.out_coff = out_coff
    a.N, a.H, a.W, a.C, a.OH, a.OW = n, h, w, c, oh, ow
    a.act = act
    md = None
    if mask is not None:
        md = nhwc(mask).to(torch.uint8).to(DEV)
        a.drop_mode = _lib.VX_DROP_MASK; a.drop_mask = md.data_ptr()
    _lib.check(lib.vx_affine_gather(C.byref(a), _lib.stream_ptr()), "affine")
    torch.cuda.synchronize()
    return nchw(out[..., out_coff:out_coff + c]).cpu().double()


def test_batchnorm_train_relu_residual():
    x = torch.from_numpy(formula_tensor((3, 48, 9, 13), 221, scale=2.0)) + 0.7
    res = torch.from_numpy(formula_tensor((3, 48, 9, 13), 222))
    gamma = 1 + torch.from_numpy(formula_tensor((48,), 223, scale=0.3))
    beta = torch.from_numpy(formula_tensor((48,), 224, scale=0.2))
    scale, shift = bn_scale_shift(x.float(), gamma, beta)
    ref_bn = F.batch_norm(x.float().double(), None, None, gamma.float().double(), beta.float().double(), training=True, eps=1e-5)
    got = run_affine(x, scale, shift, act=_lib.VX_ACT_RELU)
    assert (got - F.relu(ref_bn)).abs().max().item() < 2e-5
    got = run_affine(x, scale, shift, add=res, act=_lib.VX_ACT_RELU)            # BasicBlock end (hrnet_module.py:72-75)
    assert (got - F.relu(ref_bn + res.float().double())).abs().max().item() < 2e-5
    got = run_affine(x, scale, shift, add=res, act=0, inplace_add=True, out_pitch=64, out_coff=8)  # fusion accumulate
    assert (got - (ref_bn + res.float().double())).abs().max().item() < 2e-5


@pytest.mark.parametrize("src,dst", [((8, 15), (64, 120)), ((16, 30), (64, 120)), ((32, 60), (64, 120)), ((5, 7), (9, 20)),
                                     ((64, 120), (256, 478))])
def test_bilinear_matches_f_interpolate(src, dst):
    x = torch.from_numpy(formula_tensor((2, 8, *src), 231))
    # the reference interpolates in float32 (2D path is float32 throughout): source coordinates are
    # float32 there too, so compare against the float32 op (a float64 one differs by 1e-5 at ratio 120/478)
    ref = F.interpolate(x.float(), size=dst, mode="bilinear", align_corners=False).double()
    got = run_affine(x, out_hw=dst)
    # integer ratios are exact; at 120/478 ATen's CPU kernel (index/weight tables) and the float32 closed form of its
    # CUDA kernel (which this kernel follows) differ by ~1e-5 in the interpolation weight
    assert (got - ref).abs().max().item() < (2e-6 if dst[1] % src[1] == 0 else 2e-5)
    # with BN affine + accumulate (fuse layer j > i, hrnet_module.py:324-329)
    gamma = 1 + torch.from_numpy(formula_tensor((8,), 232, scale=0.3))
    beta = torch.from_numpy(formula_tensor((8,), 233, scale=0.2))
    scale, shift = bn_scale_shift(x.float(), gamma, beta)
    y = torch.from_numpy(formula_tensor((2, 8, *dst), 234))
    ref2 = y.float().double() + F.interpolate(
        F.batch_norm(x.float().double(), None, None, gamma.float().double(), beta.float().double(), training=True).float(),
        size=dst, mode="bilinear", align_corners=False).double()
    got = run_affine(x, scale, shift, add=y, out_hw=dst)
    assert (got - ref2).abs().max().item() < 4e-5


def test_dropout_then_upsample_into_concat():
    """F.dropout(x, 0.5, training=True) then F.interpolate then torch.cat (hrnet_module.py:642-660)"""
    x = torch.from_numpy(formula_tensor((2, 32, 8, 15), 241))
    mask = torch.from_numpy(formula_tensor((2, 32, 8, 15), 242)) > 0
    ref = F.interpolate(x.float().double() * mask * 2.0, size=(16, 30), mode="bilinear", align_corners=False)
    got = run_affine(x, out_hw=(16, 30), mask=mask, out_pitch=96, out_coff=48)
    assert (got - ref).abs().max().item() < 2e-6
    got = run_affine(x, mask=mask)  # same resolution (x0)
    assert (got - x.float().double() * mask * 2.0).abs().max().item() == 0


def test_bilinear_nchw_slots_and_hflip():
    lib = _lib.load()
    x = torch.from_numpy(formula_tensor((3, 5, 16, 30), 251))
    xd = torch.zeros((3, 16, 30, 8), dtype=torch.float32, device=DEV)
    xd[..., :5] = nhwc(x.float()).to(DEV)
    out = torch.full((4, 5, 64, 120), -77.0, dtype=torch.float32, device=DEV)
    dst = torch.tensor([2, 0, 3], dtype=torch.int32, device=DEV)
    flip = torch.tensor([0, 1, 0], dtype=torch.int32, device=DEV)
    _lib.check(lib.vx_bilinear_nchw(_lib.ptr(xd), 8, 3, 16, 30, 5, 64, 120, _lib.ptr(out), _lib.ptr(dst), _lib.ptr(flip),
                                    _lib.stream_ptr()), "bilinear_nchw")
    torch.cuda.synchronize()
    ref = F.interpolate(x.float().double(), size=(64, 120), mode="bilinear", align_corners=False)
    assert (out[2].cpu().double() - ref[0]).abs().max().item() < 2e-6
    assert (out[0].cpu().double() - torch.flip(ref[1], [-1])).abs().max().item() < 2e-6
    assert (out[3].cpu().double() - ref[2]).abs().max().item() < 2e-6
    assert (out[1] == -77.0).all()


@pytest.mark.parametrize("C", [2, 5, 19, 40])
def test_bilinear_softmax_nchw_is_bilinear_then_softmax(C):
    """vx_bilinear_softmax_nchw (round 3): the final upsample with F.softmax(dim=1) of the upsampled logits taken before the
    store (test_2D.py:300-303) -- bit for bit vx_bilinear_nchw followed by vx_softmax_planar, slots and both un-flips
    included, and the float64 reference within 1e-6."""
    lib = _lib.load()
    pitch = (C + 3) // 4 * 4
    x = torch.from_numpy(formula_tensor((3, C, 16, 30), 261, scale=3.0))
    xd = torch.zeros((3, 16, 30, pitch), dtype=torch.float32, device=DEV)
    xd[..., :C] = nhwc(x.float()).to(DEV)
    dst = torch.tensor([2, 0, 3], dtype=torch.int32, device=DEV)
    flip = torch.tensor([0, 1, 3], dtype=torch.int32, device=DEV)
    lg = torch.full((4, C, 64, 120), -77.0, dtype=torch.float32, device=DEV)
    pr = torch.full((4, C, 64, 120), -77.0, dtype=torch.float32, device=DEV)
    _lib.check(lib.vx_bilinear_nchw(_lib.ptr(xd), pitch, 3, 16, 30, C, 64, 120, _lib.ptr(lg), _lib.ptr(dst), _lib.ptr(flip),
                                    _lib.stream_ptr()), "bilinear_nchw")
    _lib.check(lib.vx_bilinear_softmax_nchw(_lib.ptr(xd), pitch, 3, 16, 30, C, 64, 120, _lib.ptr(pr), _lib.ptr(dst), _lib.ptr(flip),
                                            _lib.stream_ptr()), "bilinear_softmax_nchw")
    two = torch.empty_like(lg)
    _lib.check(lib.vx_softmax_planar(_lib.ptr(lg), 4, C, 64 * 120, _lib.ptr(two), _lib.stream_ptr()), "softmax_planar")
    torch.cuda.synchronize()
    for slot in (0, 2, 3):
        assert torch.equal(pr[slot], two[slot]), slot
    assert (pr[1] == -77.0).all()                                  # an unused slot stays untouched
    ref = torch.softmax(F.interpolate(x.float().double(), size=(64, 120), mode="bilinear", align_corners=False), dim=1)
    assert (pr[2].cpu().double() - ref[0]).abs().max().item() < 1e-6
    assert (pr[0].cpu().double() - torch.flip(ref[1], [-1])).abs().max().item() < 1e-6
    assert (pr[3].cpu().double() - torch.flip(ref[2], [-1, -2])).abs().max().item() < 1e-6


def test_conv2d_weights_of_another_k_schedule_are_refused(vxcfg):
    """The octet-granular packing of an 18-channel 3x3 layer (family 300 + row tiles) is accepted only for the Cin it pads
    to and only while the library is configured for it: the same buffer under c2s_no_oct, or passed off as a 48-channel
    layer's weights, is VX_E_DTYPE -- not numbers from the wrong layout."""
    lib = _lib.load()
    x = torch.zeros((1, 8, 16, 20), dtype=torch.float32, device=DEV)
    wt = torch.zeros((18, 18, 3, 3), dtype=torch.float32, device=DEV)
    wp = torch.empty(lib.vx_conv2d_packed_floats(18, 18, 3), dtype=torch.float32, device=DEV)
    _lib.check(lib.vx_pack_conv2d(_lib.ptr(wt), _lib.ptr(wp), 18, 18, 3, _lib.stream_ptr()), "pack2d")
    out = torch.zeros((1, 8, 16, 20), dtype=torch.float32, device=DEV)
    a = _lib.Conv2dArgs()
    a.w_family = lib.vx_conv2d_family(18, 18, 3)
    assert a.w_family == 312
    a.in_ = x.data_ptr(); a.in_pitch = 20; a.w_packed = wp.data_ptr(); a.out = out.data_ptr(); a.out_pitch = 20
    a.N, a.H, a.W, a.Cin, a.Cout, a.KS, a.S = 1, 8, 16, 32, 18, 3, 1
    _lib.check(lib.vx_conv2d(C.byref(a), _lib.stream_ptr()), "the configuration it was packed under")
    a.Cin, a.in_pitch = 48, 48          # 312 = three octets: pads to 32 input channels, not 48
    big = torch.zeros((1, 8, 16, 48), dtype=torch.float32, device=DEV)
    a.in_ = big.data_ptr()
    assert lib.vx_conv2d(C.byref(a), _lib.stream_ptr()) == -3 and b"family" in lib.vx_last_error_string()
    a.Cin, a.in_pitch, a.in_ = 32, 20, x.data_ptr()
    vxcfg.set(c2s_no_oct=1)
    assert lib.vx_conv2d(C.byref(a), _lib.stream_ptr()) == -3 and b"family" in lib.vx_last_error_string()
