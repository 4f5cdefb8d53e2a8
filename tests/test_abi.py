"""CPU: the C-ABI library loads and exports every symbol include/values_amd.h declares; the ctypes
struct mirrors have the C sizes; host-side logic that needs no GPU."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

from tests.helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "values_amd.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from values_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vx_[a-zA-Z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from values_amd import _lib
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.vx_version() >= 200


def test_struct_sizes_match_c(lib):
    from values_amd import _lib
    code = r'''
#include <stdio.h>
#include "values_amd.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(vx_conv3d_args), sizeof(vx_norm_args), sizeof(vx_convT_args),
 sizeof(vx_unet3d_weights), sizeof(vx_unet3d_run), sizeof(vx_conv2d_args), sizeof(vx_affine_args), sizeof(vx_config),
 sizeof(vx_unc_outputs), sizeof(vx_fuse_args), sizeof(vx_stat_src)); return 0;}
'''
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "s.c")
        open(src, "w").write(code)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])  # header is plain C
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(c) for c in (_lib.ConvArgs, _lib.NormArgs, _lib.ConvTArgs, _lib.UNet3DWeights, _lib.UNet3DRun,
                                        _lib.Conv2dArgs, _lib.AffineArgs, _lib.Config, _lib.UncOutputs, _lib.FuseArgs, _lib.StatSrc)]
    assert mine == sizes


def test_product_library_has_no_wrong_by_design_switches(lib):
    """Round-2 verdict: the phase-ablation switches (VX_S16_DBG / VX_C8_DBG / VX_DMA_DBG skipped the epilogue or the staging
    of a conv: wrong numbers by design) and the rejected schedules (LDS-DMA, ping-pong, the un-specialised z-column kernel,
    the 16-wide c8 tile whose statistics layout differed) must not be reachable in libvalues_amd.so: neither as vx_config
    fields nor as environment variables.  They live in the diagnostic build only (tools/build_stamps.sh,
    -DVX_CONV_STAMPS) and under tools/rejected/."""
    from values_amd import _lib
    fields = {n for n, _ in _lib.Config._fields_}
    gone = {"s16_dbg", "c8_dbg", "dma_dbg", "dma_nw16", "conv_dma", "s16_ping", "c8_tile16", "s16_no_wspec"}
    assert not (fields & gone), fields & gone
    hdr = open(os.path.join(ROOT, "include", "values_amd.h")).read()
    for f in gone:
        assert f not in hdr, f
    # the shipped binary does not even contain the variable names (the diagnostic build reads them with getenv)
    blob = open(os.path.join(ROOT, "values_amd", "libvalues_amd.so"), "rb").read()
    for name in (b"VX_S16_DBG", b"VX_C8_DBG", b"VX_DMA_DBG", b"VX_CONV_DMA", b"VX_S16_PING", b"VX_XP_ABL", b"VX_CONV_DBG_PTR"):
        assert name not in blob, name
    assert not os.path.exists(os.path.join(ROOT, "values_amd", "csrc", "conv3d_dma.hip"))
    assert not os.path.exists(os.path.join(ROOT, "values_amd", "csrc", "conv3d_xp8.hip"))


def test_config_surface_is_the_documented_one(lib):
    """Round-3 verdict: vx_config had 25+ fields, each multiplying template instances and the configurations a maintainer can
    get wrong.  Round 4 keeps ONE fallback family (conv_fp32), the data-flow A/B switches the parity tests exercise, the
    generic-instance reference and the opt-in storage16: the ctypes mirror lists exactly the header's fields, in order, the
    removed tuning knobs / measured-slower variants are in neither, and the binary no longer reads their variables."""
    import re
    from values_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "values_amd.h")).read()
    body = hdr[hdr.index("typedef struct vx_config {"):hdr.index("} vx_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [n for n, _ in _lib.Config._fields_]
    assert len(names) <= 20, names      # (round 6: + s16_no_l1dma, a data-flow switch the parity tests run)
    gone = {"conv_no_c8", "conv_no_xcd", "conv_per_cu", "s16_per_cu", "c8_per_cu", "convt_wgs", "s16_no_xp", "s16_no_db",
            "s16_no_db3", "s16_no_epi", "s16_no_ty8", "s16_no_wall", "c2s_no_nt5", "convt_no_mfma", "s16_range_check", "s16_pw",
            "s16_prio"}
    assert not (set(names) & gone)
    blob = open(os.path.join(ROOT, "values_amd", "libvalues_amd.so"), "rb").read()
    for f in gone:
        assert ("VX_" + f.upper()).encode() + b"\0" not in blob, f
    assert lib.vx_version() >= 600


def test_host_only_queries(lib, vxcfg):
    # default = split-fp16 schedule: [row groups][chunks of CB][K=32 steps][NT][hi|lo][64 lanes][8 halves], in floats
    def s16(cin, cout):
        nt = 2 if cout % 32 == 0 else 1
        xp = cout == 8                      # x-pair packing: chunks of 8 channels, 9 (kz, ky) steps
        cb = 8 if xp else (16 if cin % 16 == 0 else 8)
        steps = 9 if xp else -(-27 // (32 // cb))
        rows = -(-cout // (16 * nt)) * 16 * nt
        pieces = 32 if xp else 64           # x-pair blocks are [co >> 2][kx 4][co & 3] pieces, not one per lane
        return (rows // 16) * (cin // cb) * steps * 2 * pieces * 8 // 2
    # round 5, family 7 (Cout % 32 == 0, Cin >= 16): the tile kernel's fragments followed by the deep-layer kernel's
    # [Cout / 32][Cin / 8][7 K steps][2 row tiles][hi | lo][64 lanes][8 halves] (conv3d_deep.hip)
    def deep(cin, cout):
        return (cout // 32) * (cin // 8) * 7 * 2 * 2 * 64 * 8 // 2 if cout % 32 == 0 and cin >= 16 else 0
    for cin, cout in ((16, 8), (24, 8), (32, 16), (32, 32), (128, 64), (8, 32)):
        assert lib.vx_conv3d_k3_packed_floats(cin, cout) == s16(cin, cout) + deep(cin, cout)
    # round 5, family 6 (Cout = 16, Cin in {8, 16}): the tile kernel's fragments followed by the z-column kernel's
    # [14 or 9 K steps][hi | lo][64 lanes][8 halves] (conv3d_zc16.hip; which kernel runs depends on the volume's shape)
    assert lib.vx_conv3d_k3_packed_floats(16, 16) == s16(16, 16) + 14 * 2 * 64 * 8 // 2
    assert lib.vx_conv3d_k3_packed_floats(8, 16) == s16(8, 16) + 9 * 2 * 64 * 8 // 2
    assert [lib.vx_conv3d_k3_family(ci, co) for ci, co in ((16, 8), (16, 16), (8, 16), (32, 16), (16, 32), (8, 32), (128, 128), (3, 8))] == [2, 6, 6, 1, 7, 1, 7, 0]
    assert lib.vx_conv3d_k3_pool_layout(64, 64, 64, 8, 8) == 1 and lib.vx_conv3d_k3_pool_layout(32, 32, 32, 16, 16) == 2
    assert lib.vx_conv3d_k3_pool_layout(16, 16, 16, 16, 16) == 0 and lib.vx_conv3d_k3_pool_layout(32, 32, 32, 32, 32) == 0
    vxcfg.set(s16_no_zc16=1)
    assert lib.vx_conv3d_k3_pool_layout(32, 32, 32, 16, 16) == 0 and lib.vx_conv3d_k3_family(16, 16) == 6   # the knob never changes the layout
    vxcfg.set(s16_no_zc16=0)
    # 2D split-fp16 family = 10 + row tiles per workgroup (the packed layout is [row group][...][row tile])
    assert [lib.vx_conv2d_family(64, co, ks) for co, ks in ((64, 3), (720, 1), (64, 5))] == [12, 15, 0]
    # + 100 x the octets of the octet-granular K schedule for 3x3 layers of <= 8 or 17..24 REAL input channels (round 3)
    assert [lib.vx_conv2d_family(18, co, 3) for co in (18, 36, 72, 144, 48, 96, 19)] == [312, 313, 313, 313, 313, 313, 312]
    assert [lib.vx_conv2d_family(ci, 64, 3) for ci in (3, 8, 9, 16, 17, 24, 25, 36)] == [112, 112, 12, 12, 312, 312, 12, 12]
    assert lib.vx_conv2d_family(18, 36, 1) == 13                    # 1x1 layers keep the sub-block schedule
    assert lib.vx_conv2d_packed_floats(18, 18, 3) == 7 * 2 * 2 * 64 * 8 // 2      # 7 K steps x 2 row tiles, not 10
    assert lib.vx_conv2d_packed_floats(3, 64, 3) == 2 * 3 * 2 * 2 * 64 * 8 // 2   # 2 row groups x 3 steps x 2 tiles
    vxcfg.set(c2s_no_oct=1)
    assert lib.vx_conv2d_family(18, 18, 3) == 12 and lib.vx_conv2d_packed_floats(18, 18, 3) == 10 * 2 * 2 * 64 * 8 // 2
    vxcfg.set(c2s_no_oct=0)
    assert [lib.vx_conv2d_family(270, co, 1) for co in (270, 720, 19)] == [15, 15, 12]
    vxcfg.set(conv_fp32=1)      # native-fp32 kernels: their own packings and families
    assert lib.vx_conv3d_k3_packed_floats(16, 8) == 27 * 16 * 8   # Cout == 8, Cin in {8, 16}: 4x4x1 kernel, dense
    assert lib.vx_conv3d_k3_packed_floats(24, 8) == 16 * 24 * 36  # other Cout == 8: x-pair packing, 16 rows x (9*4 taps)
    assert lib.vx_conv3d_k3_packed_floats(16, 16) == 16 * 16 * 27
    assert lib.vx_conv3d_k3_packed_floats(32, 32) == 32 * 32 * 27
    assert lib.vx_conv3d_k3_tiles_for(64, 64, 64, 8) == 2 * 16 * 16  # x-pair / 4x4x1 tiles are 32 voxels wide
    assert [lib.vx_conv3d_k3_family(ci, co) for ci, co in ((16, 8), (24, 8), (16, 16))] == [5, 4, 3]
    assert lib.vx_conv2d_family(64, 64, 3) == 3
    vxcfg.set(conv_fp32=0)
    assert lib.vx_conv3d_k3_packed_floats(3, 8) == -1
    assert lib.vx_convT_k2s2_packed_floats(16, 8) == 16 * 8 * 8
    assert lib.vx_conv3d_k3_tiles(64, 64, 64) == 4 * 16 * 16
    assert lib.vx_conv3d_k3_tiles_for(64, 64, 64, 8) == 2 * 8 * 16   # split-fp16 x-pair tiles of large layers: 32 x 8 x 4
    assert lib.vx_conv3d_k3_tiles_for(16, 16, 16, 16) == 1 * 4 * 4  # small layers: 16 x 4 x 4
    # vx_conv3d_k3_tiles sizes stats_partial for any layer: an upper bound of every tiling, in both modes (for
    # 8 <= W < 16 the x-pair tiling makes the most tiles, elsewhere the plain one: tools/fuzz_conv.py found the gap)
    for mode in (0, 1):
        vxcfg.set(conv_fp32=mode)
        for d, h, w in ((8, 52, 13), (3, 5, 9), (4, 40, 15), (64, 64, 64), (5, 37, 70), (1, 1, 1), (6, 33, 16), (7, 31, 33)):
            for cout in (8, 16, 32, 64):
                assert lib.vx_conv3d_k3_tiles(d, h, w) >= lib.vx_conv3d_k3_tiles_for(d, h, w, cout), (mode, d, h, w, cout)
    vxcfg.set(conv_fp32=0)
    assert lib.vx_unet3d_workspace_bytes(1, 64, 64, 64, 8) > 40e6
    assert lib.vx_unet3d_workspace_bytes(0, 64, 64, 64, 8) == 0


def test_round6_host_queries_and_arg_checks_without_a_gpu(lib, vxcfg):
    """Round 6, host side only: where the planar pre-split hand-over applies; the argument checks of vx_conv3d_k3 run BEFORE any
    launch, so a wrong combination is refused on a box without a GPU too (the round-5 advisor found a shape bug this way); the
    deep-layer kernel's applicability no longer depends on the batch size."""
    import ctypes as C
    from values_amd import _lib
    assert lib.vx_conv3d_k3_planar_ok(32, 32, 32, 16, 16) == 1 and lib.vx_conv3d_k3_planar_ok(16, 24, 64, 16, 16) == 1
    assert lib.vx_conv3d_k3_planar_ok(32, 32, 16, 16, 16) == 0 and lib.vx_conv3d_k3_planar_ok(32, 32, 32, 8, 16) == 0
    assert lib.vx_conv3d_k3_planar_ok(64, 64, 64, 8, 8) == 0
    vxcfg.set(s16_no_zc16=1)
    assert lib.vx_conv3d_k3_planar_ok(32, 32, 32, 16, 16) == 0
    vxcfg.set(s16_no_zc16=0)

    def args(cin, cout, d, h, w, n=2, **kw):
        a = _lib.ConvArgs()
        dummy = 0x10000                      # never dereferenced: every call below is refused by a host-side check
        a.in_ = dummy; a.w_packed = dummy; a.bias = dummy; a.out = dummy
        a.in_pitch, a.out_pitch, a.out_coff = cin, cout, 0
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w, cin, cout
        a.w_family = lib.vx_conv3d_k3_family(cin, cout)
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    refused = [args(16, 16, 16, 16, 16, in_planar=1, act=_lib.VX_ACT_RELU),                 # a shape the z-column kernel does not take
               args(8, 16, 32, 32, 32, in_planar=1, act=_lib.VX_ACT_RELU),                   # Cin = 8
               args(16, 16, 32, 32, 32, out_planar=1, stats_partial=0x10000),               # statistics epilogue
               args(16, 16, 32, 32, 32, out_planar=1, act=_lib.VX_ACT_RELU, out_pitch=32),  # not a dense 16-channel output
               args(32, 32, 16, 16, 16, products=1, act=_lib.VX_ACT_RELU),                   # one-product mode away from the full-resolution kernels
               args(8, 8, 64, 64, 64, products=1, act=_lib.VX_ACT_LRELU, drop_mode=_lib.VX_DROP_HASH),   # ... and for an instance that does not exist
               args(16, 16, 32, 32, 32, products=2, act=_lib.VX_ACT_RELU)]
    for a in refused:
        rc = lib.vx_conv3d_k3(C.byref(a), None)
        assert rc == -2, (rc, lib.vx_last_error_string())      # VX_E_SHAPE
    assert lib.vx_zero(None, 0, None) == 0 and lib.vx_zero(None, 16, None) < 0


def test_derive_seed_keeps_index_zero_and_separates_the_driver_levels():
    """predict.derive_seed (round-5 advice): chunk 0 / block 0 keep the caller's seed (a one-chunk run IS the plain run, the parity
    tests export masks for that seed); (block k, chunk 0) and (block 0, chunk k) -- which the additive stride of rounds 4-5 made
    the SAME seed -- differ, and so do all (level, index) pairs of a realistic run."""
    from values_amd.predict import derive_seed
    for s in (0, 1, 123, 0xFFFFFFFF, 4242):
        assert derive_seed(s, 0, 0) == s & 0xFFFFFFFF and derive_seed(s, 1, 0) == s & 0xFFFFFFFF
        seen = {}
        for block in range(0, 64):
            sb = derive_seed(s, 1, block)
            for chunk in range(0, 8):
                v = derive_seed(sb, 0, chunk)
                assert 0 <= v <= 0xFFFFFFFF
                assert v not in seen, (s, (block, chunk), seen[v])
                seen[v] = (block, chunk)
        old = lambda seed, b, c: (seed + 0x9E3779B1 * b + 0x9E3779B1 * c) & 0xFFFFFFFF       # rounds 4-5
        assert old(s, 3, 0) == old(s, 0, 3) and derive_seed(derive_seed(s, 1, 3), 0, 0) != derive_seed(derive_seed(s, 1, 0), 0, 3)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from values_amd import UNet3D, _lib, calculate_uncertainty
    with pytest.raises(_lib.VxError):
        calculate_uncertainty(torch.zeros(2, 2, 4))
    with pytest.raises(_lib.VxError):
        UNet3D(num_classes=2)(torch.zeros(1, 1, 16, 16, 16))
    # round 4's entry points: the device-built TTA views (no host fallback: the host restatement is a different function)
    from values_amd.data import tta_views_2d_device, tta_views_8_device
    with pytest.raises(_lib.VxError):
        tta_views_2d_device(torch.zeros((4, 6, 3), dtype=torch.uint8), [0.5] * 3, [0.2] * 3)
    with pytest.raises(_lib.VxError):
        tta_views_8_device(torch.zeros(1, 3, 4, 6), torch.zeros(1, 3, 4, 6))


def test_tta_view_codes_and_nhwc_views_container():
    """host logic of the device TTA path: the view codes of vx_tta_views_2d for the reference's four views and config C4's
    eight, the transform names test_2D.py un-flips by, the NhwcViews container's checks"""
    import torch
    from values_amd.data import TTA_2D_TRANSFORMS, TTA_2D_VIEW_CODES, TTA_8_VIEW_CODES, hflip_flags
    from values_amd.predict2d import NhwcViews
    assert TTA_2D_VIEW_CODES == [0, 1, 4, 1 | 4 | 16]
    assert [bool(c & 1) for c in TTA_2D_VIEW_CODES] == hflip_flags(TTA_2D_TRANSFORMS)
    assert [c & 3 for c in TTA_8_VIEW_CODES] == [0, 1, 2, 3, 0, 1, 2, 3] and all((c & 4) for c in TTA_8_VIEW_CODES[4:])
    v = NhwcViews(torch.zeros(4, 2, 5, 6, 4), [False, True, False, True])
    assert v.vflip == [False] * 4 and v.t.is_contiguous()
    with pytest.raises(ValueError):
        NhwcViews(torch.zeros(4, 2, 5, 6, 3), [False] * 4)
    with pytest.raises(ValueError):
        NhwcViews(torch.zeros(4, 2, 5, 6, 4), [False] * 3)


def test_crop_indices_match_reference_fixture():
    import json
    from values_amd import crop_indices
    with open(os.path.join(GOLDEN, "patch_index.json")) as f:
        g = json.load(f)
    shapes = g.pop("_shapes")
    for key, crops in g.items():
        _, name, patch, overlap = key.split("|")
        mine = crop_indices(shapes[name], int(patch), float(overlap))
        assert [[list(c) for c in ci] for ci in mine] == crops, key


def test_tta_flip_codes():
    from values_amd.predict import FLIP_DIMS, TTA_FLIP_CODES
    assert TTA_FLIP_CODES == [0, 1, 2, 4, 3, 5, 6, 7]
    assert FLIP_DIMS == [(2,), (3,), (4,), (2, 3), (2, 4), (3, 4), (2, 3, 4)]


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under values_amd/ or tools/ may reference it -- only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
    for sub in ("values_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".cpp", ".h", ".sh")):
                    txt = open(os.path.join(dirpath, fn)).read()
                    assert "import oracle" not in txt and "from oracle" not in txt, (sub, fn)
    bench = open(os.path.join(ROOT, "bench.py")).read()          # bench.py: inside the cpu_baseline legs only (C2's and C4's)
    import re
    outside = re.sub(r"\ndef cpu_baseline_leg\w*\(.*?(?=\n(?:def |# -{20}))", "\n", bench, flags=re.S)
    assert outside.count("def cpu_baseline_leg") == 0 and len(outside) < len(bench)
    assert "from oracle" not in outside and "import oracle" not in outside


def test_product_package_holds_no_test_infrastructure():
    """Round-2 verdict: the closed-form test-input generators left the product package (tests/formula.py); values_amd/
    keeps only the shipped HRNet layouts (values_amd/hrnet_configs.py).  Neither values_amd/ nor bench.py imports tests/."""
    assert not os.path.exists(os.path.join(ROOT, "values_amd", "formula.py"))
    for dirpath, _, files in os.walk(os.path.join(ROOT, "values_amd")):
        for fn in files:
            if fn.endswith(".py"):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "from tests" not in txt and "import tests" not in txt and "values_amd.formula" not in txt, fn
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "from tests" not in bench and "import tests" not in bench


def test_load_patch_and_tta_views(tmp_path):
    import numpy as np
    from values_amd.data import hflip_flags, load_patch, tta_views_2d
    vol = np.arange(6 * 5 * 4, dtype=np.float32).reshape(6, 5, 4)
    np.save(tmp_path / "v.npy", vol)
    np.save(tmp_path / "v_00.npy", (vol > 50).astype(np.uint8))
    s = {"image_path": str(tmp_path / "v.npy"), "label_paths": [str(tmp_path / "v_00.npy")], "crop_idx": ((1, 5), (0, 4), (2, 4))}
    d = load_patch(s)
    assert d["data"].shape == (1, 4, 4, 2) and d["seg"].shape == (1, 1, 4, 4, 2) and d["seg"].dtype == np.intc
    np.testing.assert_array_equal(d["data"][0], vol[1:5, 0:4, 2:4])
    assert d["org_image_size"] == [(6, 5, 4)]
    img = (np.arange(4 * 6 * 3) % 256).astype(np.uint8).reshape(4, 6, 3)
    views, tr = tta_views_2d(img, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    assert len(views) == 4 and views[0].shape == (3, 4, 6) and views[0].dtype == np.float32
    np.testing.assert_array_equal(views[1], views[0][:, :, ::-1])
    np.testing.assert_allclose(views[0][0, 0, 0], (img[0, 0, 0] / 255.0 - 0.485) / 0.229, rtol=1e-6)
    assert hflip_flags(tr) == [False, True, False, True]
    # the host function against the oracle's restatement of the dataset branch over albumentations 1.3.0's published
    # arithmetic (oracle/tta2d_oracle.py; the library itself is absent: parity unpinned), bit for bit: random images with both
    # ends of the uint8 clip, float32 noise fields, a missing field
    from oracle import tta2d_oracle
    rng = np.random.default_rng(11)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    for trial in range(4):
        H, W = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        im = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        im[0, :] = 0; im[-1, :] = 255
        g0 = (rng.standard_normal((H, W, 3)) * 40).astype(np.float32)
        g1 = (rng.standard_normal((H, W, 3)) * 40).astype(np.float32) if trial != 2 else None
        got, tr = tta_views_2d(im, mean, std, noise=g0, noise_flipped=g1)
        want, tr_o = tta2d_oracle.tta_branch(im, mean, std, g0, g1)
        assert tr == tr_o
        for g in range(4):
            assert got[g].dtype == np.float32 and got[g].shape == (3, H, W)
            np.testing.assert_array_equal(got[g], want[g], err_msg=f"trial {trial} view {g}")
