"""GPU: whole-network parity of values_amd.UNet3D / predict_* (HIP, fp32) against
 (1) the golden fixtures produced by the imported reference (float64, test_3D.py:425), and
 (2) the oracle on fresh seeded inputs,
plus size-independent properties at the full 64^3, T=10 configuration.

Tolerances (BASELINE.json north_star): uncertainty maps within 1e-4 abs; argmax masks bit-exact
(outside voxels whose top-2 margin in the reference is below 1e-5, where float32 vs float64
rounding can legitimately flip a tie -- SURVEY section 7 "Precision")."""
import numpy as np
import pytest
import torch

from tests.helpers import formula_sd_torch, load_npz, unpack_masks
from tests.formula import formula_unet3d_state_dict, formula_volume

pytestmark = pytest.mark.gpu

KEYS = ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty")
MAP_TOL = 1e-4
LOGIT_TOL = 1e-4
TIE = 1e-5
# Regression alarms beside the 1e-4 contract (round-5 verdict, "tighten the alarms"): the path measures 3e-6 on logits and 3e-7 on the
# maps against the float64 oracle.  DESIGN section 4 records a gfx950 hazard (v_fma_mix* feeding a matrix instruction without wait
# states) that produced 1e-4-sized, run-to-run different errors -- a contract-sized assert would have let it through.  These bounds sit
# at ~5x the measured deviation of the split-fp16 path and are asserted wherever the network runs on its DEFAULT kernels.
REG_LOGIT_TOL = 2e-5
REG_MAP_TOL = 3e-6


def assert_close(err, contract, regression, what):
    """the north star's tolerance AND the regression bound (the message says which one broke)"""
    assert err < contract, (what, "contract", err)
    assert err < regression, (what, "regression bound: the path measured ~5x below this in round 5", err)


def make_model(seed_tag=0, do_dropout=True, double=False, **kw):
    from values_amd import UNet3D
    m = UNet3D(num_classes=2, do_dropout=do_dropout, **kw)
    sd = {k: torch.from_numpy(v).float() for k, v in formula_unet3d_state_dict(seed_tag=seed_tag).items()}
    missing = m.load_state_dict(sd, strict=True)  # same key names as the reference checkpoint
    assert not missing.missing_keys and not missing.unexpected_keys
    m = m.cuda()
    return m.double() if double else m


def stacked_masks(g, T):
    """17 masks, each (T, C, D,H,W) bool: sample t of the single volume uses pass t's masks."""
    from oracle.unet3d_oracle import DROPOUT_ORDER
    per_t = [unpack_masks(g, t) for t in range(T)]
    return [torch.from_numpy(np.concatenate([per_t[t][name] for t in range(T)], 0)) for name in DROPOUT_ORDER]


def test_state_dict_names_match_reference():
    from values_amd import UNet3D
    from tests.formula import unet3d_param_shapes
    m = UNet3D(num_classes=2, do_dropout=True)
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ours == dict(unet3d_param_shapes())
    m2 = UNet3D(num_classes=3, aleatoric_loss=True)
    assert tuple(m2.state_dict()["final_aleatoric.weight"].shape) == (6, 8, 1, 1, 1)


@pytest.mark.parametrize("size", [16, 32])
def test_golden_mc_dropout_with_reference_masks(size):
    """Config C1 (32^3, T=4) and its 16^3 sibling: the reference's own dropout masks injected."""
    from values_amd import predict_uncertainty
    g = load_npz(f"unet3d_{size}.npz")
    T = g["logits"].shape[0]
    model = make_model(double=True)  # .double() like test_3D.py:425 must be accepted
    x = torch.from_numpy(g["input"]).double().cuda()
    out = predict_uncertainty([model], x, n_pred=T, dropout_masks=[stacked_masks(g, T)], want_sample_argmax=True)
    logits = out["logits"][0].cpu().numpy()
    assert_close(np.abs(logits - g["logits"]).max(), LOGIT_TOL, REG_LOGIT_TOL, "logits")
    for k in KEYS:
        assert_close(np.abs(out[k][0].cpu().numpy() - g[k]).max(), MAP_TOL, REG_MAP_TOL, k)
    assert_close(np.abs(out["mean_softmax"][0].cpu().numpy() - g["mean_softmax"]).max(), MAP_TOL, REG_MAP_TOL, "mean_softmax")
    clear = g["mean_margin"] > TIE
    assert (out["pred_seg_mean"][0].cpu().numpy() == g["mean_seg"])[clear].all()
    assert clear.mean() > 0.999
    sm = torch.softmax(torch.from_numpy(g["logits"]).double(), 1).numpy()
    ssrt = np.sort(sm, axis=1)
    sclear = (ssrt[:, -1] - ssrt[:, -2]) > TIE
    assert (out["pred_seg"][0].cpu().numpy() == g["pred_seg"])[sclear].all()


@pytest.mark.parametrize("size", [16, 32])
def test_golden_no_dropout_forward(size):
    g = load_npz(f"unet3d_{size}.npz")
    model = make_model(do_dropout=False)
    x = torch.from_numpy(g["input"]).cuda()
    with torch.no_grad():
        y = model(x)
    assert y.shape == (1, 2, size, size, size) and y.dtype == torch.float32
    assert_close(np.abs(y[0].cpu().numpy() - g["logits_nodrop"]).max(), LOGIT_TOL, REG_LOGIT_TOL, "logits_nodrop")
    # eval() turns MC-dropout off like nn.Dropout does
    md = make_model(do_dropout=True).eval()
    with torch.no_grad():
        y2 = md(x)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("size", [16, 32])
def test_native_fp32_mode_network_vs_golden(size, vxcfg):
    """VX_CONV_FP32=1: the native-fp32 matrix kernels (16x16x4 / 4x4x1 with the final 1x1x1 conv fused into
    expand_1_2) give the same network within tolerance, with the reference's dropout masks and TTA flips"""
    from values_amd import predict_uncertainty
    vxcfg.setenv("VX_CONV_FP32", "1")
    g = load_npz(f"unet3d_{size}.npz")
    T = g["logits"].shape[0]
    model = make_model(do_dropout=True)
    x = torch.from_numpy(g["input"]).cuda()
    with torch.no_grad():
        logits = model(x, n_samples=T, dropout_masks=stacked_masks(g, T))
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < LOGIT_TOL
    ge = load_npz("ensemble_tta_16.npz")
    models = [make_model(seed_tag=s, do_dropout=False) for s in range(3)]
    out = predict_uncertainty(models, torch.from_numpy(ge["input"]).cuda(), tta=True, x_noise=torch.from_numpy(ge["input_noise"]).cuda())
    for k in KEYS:
        assert np.abs(out[k][0].cpu().numpy() - ge[k]).max() < MAP_TOL, k


def test_golden_ensemble_and_tta_order():
    """Config C3 ordering (members) and the 16-view TTA of test_3D.py:426-456, against the buffer the
    reference's own concat_data filled."""
    from values_amd import predict_uncertainty
    g = load_npz("ensemble_tta_16.npz")
    models = [make_model(seed_tag=s, do_dropout=False) for s in range(3)]
    x = torch.from_numpy(g["input"]).cuda()
    xn = torch.from_numpy(g["input_noise"]).cuda()
    out = predict_uncertainty(models, x, tta=True, x_noise=xn)
    sm = torch.softmax(out["logits"][0], 1).cpu().numpy()
    assert sm.shape == g["softmax_pred"].shape
    assert np.abs(sm - g["softmax_pred"]).max() < MAP_TOL
    for k in KEYS:
        assert np.abs(out[k][0].cpu().numpy() - g[k]).max() < MAP_TOL, k
    clear = g["mean_margin"] > TIE
    assert (out["pred_seg_mean"][0].cpu().numpy() == g["mean_seg"])[clear].all()
    ens = predict_uncertainty(models, x, n_pred=1)
    sm = torch.softmax(ens["logits"][0], 1).cpu().numpy()
    assert np.abs(sm - g["ens_softmax_pred"]).max() < MAP_TOL
    for k in KEYS:
        assert np.abs(ens[k][0].cpu().numpy() - g["ens_" + k]).max() < MAP_TOL, k


def test_fresh_input_vs_oracle_48_and_batch_independence():
    """Oracle (float64) on an input/shape that is NOT in the fixtures (48^3 exercises ragged tiles at the deeper
    levels: 48 -> 24 -> 12 -> 6 -> 3), two different volumes in one batch."""
    from oracle.unet3d_oracle import unet3d_forward
    sd = formula_sd_torch(seed_tag=2)
    model = make_model(seed_tag=2, do_dropout=False)
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1, 48, 48, 48), tag=21),
                                         formula_volume((1, 1, 48, 48, 48), tag=22)], 0))
    with torch.no_grad():
        ref = unet3d_forward(sd, x, masks=None).numpy()
        got = model(x.float().cuda())
    assert np.abs(got.cpu().numpy() - ref).max() < LOGIT_TOL
    # InstanceNorm is per sample: a sample's result must not depend on its batch mates (bit exact)
    with torch.no_grad():
        g0 = model(x[:1].float().cuda())
        g1 = model(x[1:].float().cuda())
    assert torch.equal(got[0], g0[0]) and torch.equal(got[1], g1[0])


def test_non_cubic_volume_vs_oracle():
    from oracle.unet3d_oracle import unet3d_forward
    sd = formula_sd_torch(seed_tag=1)
    model = make_model(seed_tag=1, do_dropout=False)
    x = torch.from_numpy(formula_volume((1, 1, 16, 32, 48), tag=23))
    with torch.no_grad():
        ref = unet3d_forward(sd, x, masks=None).numpy()
        got = model(x.float().cuda()).cpu().numpy()
    assert np.abs(got - ref).max() < LOGIT_TOL
    with pytest.raises(Exception):  # 40 is not divisible by 16: the reference's InstanceNorm/concat would fail too
        model(torch.zeros(1, 1, 40, 16, 16).cuda())


def test_non_cubic_volume_on_the_level1_z_column_kernel_vs_oracle():
    """(32, 48, 128): level 1 is 16 x 24 x 64 -- two column tiles in x, three in y, eight items in z on conv3d_zc16.hip (round 5),
    the two-launch expand_2_1 with upscale3 inside; MC-dropout with the exported hash masks and dropout off, against the float64
    oracle.  Two volumes: several samples per workgroup."""
    from values_amd import predict_uncertainty
    import bench
    S, T, seed = (32, 48, 128), 2, 99
    model = make_model(seed_tag=3, do_dropout=True)
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1) + S, tag=91), formula_volume((1, 1) + S, tag=92)], 0))
    out = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    names = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    assert any(n.startswith("conv3d_zc16_kernel<16,5,0,1,1>") for n in names) and any(n.startswith("conv3d_zc16_kernel<16,4,1,0,0>") for n in names), names
    sd = formula_sd_torch(seed_tag=3)
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, 2 * T, *S)]           # sample n = volume * T + pass
    for v in range(2):
        logits, ref = _oracle_maps(sd, x[v:v + 1], [[m[v * T + t:v * T + t + 1] for m in masks] for t in range(T)])
        assert np.abs(out["logits"][v].cpu().numpy() - logits).max() < LOGIT_TOL
        for k in KEYS:
            assert np.abs(out[k][v].cpu().numpy() - ref[k]).max() < MAP_TOL, k
    det = make_model(seed_tag=3, do_dropout=False)
    from oracle.unet3d_oracle import unet3d_forward
    with torch.no_grad():
        y = det(x.float().cuda())
        ref0 = unet3d_forward(sd, x, masks=None).numpy()
    assert np.abs(y.cpu().numpy() - ref0).max() < LOGIT_TOL


def test_non_cubic_volume_with_six_planes_at_level_3_vs_oracle():
    """(48, 64, 128) -> 24 x 32 x 64 -> 12 x 16 x 32 -> 6 x 8 x 16 -> 3 x 4 x 8: D % 4 == 2 at the level of contr_4_1 / contr_4_2, where the
    deep-layer kernel's tiles do not divide the statistics buffer (round-5 advice: the whole forward failed with VX_E_SHAPE); two
    volumes, dropout off and MC-dropout through the exported hash masks."""
    from oracle.unet3d_oracle import unet3d_forward
    from values_amd import predict_uncertainty
    S = (48, 64, 128)
    sd = formula_sd_torch(seed_tag=6)
    det = make_model(seed_tag=6, do_dropout=False)
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1) + S, tag=95), formula_volume((1, 1) + S, tag=96)], 0))
    with torch.no_grad():
        y = det(x.float().cuda())
        ref0 = unet3d_forward(sd, x, masks=None).numpy()
    assert_close(np.abs(y.cpu().numpy() - ref0).max(), LOGIT_TOL, REG_LOGIT_TOL, "logits")
    T, seed = 2, 31
    model = make_model(seed_tag=6, do_dropout=True)
    out = predict_uncertainty([model], x[:1].float().cuda(), n_pred=T, seeds=[seed])
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, T, *S)]
    logits, ref = _oracle_maps(sd, x[:1], [[m[t:t + 1] for m in masks] for t in range(T)])
    assert_close(np.abs(out["logits"][0].cpu().numpy() - logits).max(), LOGIT_TOL, REG_LOGIT_TOL, "logits (MC)")
    for k in KEYS:
        assert_close(np.abs(out[k][0].cpu().numpy() - ref[k]).max(), MAP_TOL, REG_MAP_TOL, k)


@pytest.mark.parametrize("f,size,ncls", [(16, 32, 2), (32, 16, 3)])
def test_wider_networks_vs_oracle(f, size, ncls):
    """initial_filter_size 16 / 32 (ctor kwarg of unet3D_module.py:8-35): every layer on the 16x16x4 kernels, separate
    1x1x1 head (no fusion), more classes"""
    from oracle.unet3d_oracle import unet3d_forward
    from values_amd import UNet3D
    sd64 = formula_unet3d_state_dict(seed_tag=3, num_classes=ncls, f=f)
    sd = {k: torch.from_numpy(v) for k, v in sd64.items()}
    m = UNet3D(num_classes=ncls, initial_filter_size=f, do_dropout=False)
    res = m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.cuda()
    x = torch.from_numpy(formula_volume((2, 1, size, size, size), tag=29))
    with torch.no_grad():
        ref = unet3d_forward(sd, x, masks=None).numpy()
        got = m(x.float().cuda()).cpu().numpy()
    assert got.shape == (2, ncls, size, size, size)
    assert np.abs(got - ref).max() < LOGIT_TOL


def test_large_non_cubic_volume_vs_oracle():
    """a monolithic 96 x 128 x 80 volume (larger than any patch config; many workgroup rounds, ragged last tiles in y)"""
    from oracle.unet3d_oracle import unet3d_forward
    sd = formula_sd_torch(seed_tag=4)
    model = make_model(seed_tag=4, do_dropout=False)
    x = torch.from_numpy(formula_volume((1, 1, 96, 128, 80), tag=31))
    with torch.no_grad():
        ref = unet3d_forward(sd, x, masks=None).numpy()
        got = model(x.float().cuda()).cpu().numpy()
    assert np.abs(got - ref).max() < LOGIT_TOL


def test_large_input_magnitudes_vs_oracle():
    """the split-fp16 conv operands must stay below 65504: raw intensities of any scale are fine because the first conv
    (Cin = 1) runs in plain fp32 and every later conv input is post-InstanceNorm / post-activation"""
    from oracle.unet3d_oracle import unet3d_forward
    sd = formula_sd_torch(seed_tag=5)
    model = make_model(seed_tag=5, do_dropout=False)
    x = torch.from_numpy(formula_volume((1, 1, 32, 32, 32), tag=33)) * 3.0e4 + 1.0e5      # e.g. unnormalised CT-like values
    with torch.no_grad():
        ref = unet3d_forward(sd, x, masks=None).numpy()
        got = model(x.float().cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 5e-4      # float32 cancellation in the first layer's statistics at offset 1e5


def test_full_size_properties_64_T10():
    """BASELINE config C2 (64^3, T=10, hash dropout) -- too big for a float64 CPU oracle inside the suite
    (7 s/pass), so check what must hold at any size."""
    from values_amd import predict_uncertainty
    model = make_model(do_dropout=True)
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1, 64, 64, 64), tag=31),
                                         formula_volume((1, 1, 64, 64, 64), tag=32)], 0)).float().cuda()
    a = predict_uncertainty([model], x, n_pred=10, seeds=[77])
    b = predict_uncertainty([model], x, n_pred=10, seeds=[77])
    c = predict_uncertainty([model], x, n_pred=10, seeds=[78])
    ln2 = float(np.log(2.0))
    for k in KEYS + ("mean_softmax",):
        assert torch.equal(a[k], b[k]), k                       # same seed -> bit-identical (no atomics anywhere)
        assert not torch.isnan(a[k]).any()
    assert not torch.equal(a["epistemic_uncertainty"], c["epistemic_uncertainty"])  # other seed -> other samples
    pe, ee, mi = a["pred_entropy"], a["aleatoric_uncertainty"], a["epistemic_uncertainty"]
    assert pe.min().item() >= 0 and pe.max().item() <= ln2 + 1e-6
    assert ee.min().item() >= 0 and ee.max().item() <= ln2 + 1e-6
    assert mi.min().item() > -1e-6                               # Jensen: H[mean p] >= mean H[p]
    assert torch.equal(mi, pe - ee)
    assert (a["mean_softmax"].sum(1) - 1).abs().max().item() < 1e-5
    assert torch.equal(a["pred_seg_mean"], (a["mean_softmax"][:, 1] > a["mean_softmax"][:, 0]).to(torch.uint8))
    # MC samples really differ, and the MI statistic is in the range the reference shows at 32^3 (max 0.063)
    lg = a["logits"]
    assert (lg[:, 0] - lg[:, 1]).abs().max().item() > 1e-3
    assert 1e-4 < mi.mean().item() < 0.2
    # the two volumes of a batch see different dropout draws but the same statistics
    assert abs(mi[0].mean().item() - mi[1].mean().item()) < 0.5 * mi.mean().item()
    # mean of the hash-dropout network approaches the same answer from two disjoint seed sets
    assert (a["mean_softmax"] - c["mean_softmax"]).abs().mean().item() < 0.05


def test_volume_chunks_on_two_streams_equal_one_stream():
    """predict's n_streams: two volume chunks on two HIP streams (own workspace, own map reduction) give the bits of the
    one-stream run -- everything for a deterministic network, the first chunk's rows with hash dropout (the second
    chunk draws its masks from another hash stream); TTA views and small batches included."""
    from values_amd import predict_uncertainty
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1, 16, 16, 16), tag=40 + i) for i in range(8)], 0)).float().cuda()
    keys = KEYS + ("mean_softmax", "pred_seg_mean", "logits")
    det = make_model(do_dropout=False)
    a = predict_uncertainty([det], x, n_pred=16, n_streams=1)
    b = predict_uncertainty([det], x, n_pred=16, n_streams=2)      # 2 chunks x 4 volumes x 16 passes
    for k in keys:
        assert torch.equal(a[k], b[k]), k
    a = predict_uncertainty([det], x, tta=True, x_noise=x * 1.01, n_streams=1)
    b = predict_uncertainty([det], x, tta=True, x_noise=x * 1.01, n_streams=2)
    for k in keys:
        assert torch.equal(a[k], b[k]), k
    mc = make_model(do_dropout=True)
    a = predict_uncertainty([mc], x, n_pred=16, seeds=[5], n_streams=1)
    b = predict_uncertainty([mc], x, n_pred=16, seeds=[5], n_streams=2)
    c = predict_uncertainty([mc], x, n_pred=16, seeds=[5], n_streams=2)
    for k in keys:
        assert torch.equal(a[k][:4], b[k][:4]), k
        assert torch.equal(b[k], c[k]), k                               # and the chunked run is reproducible
        assert not torch.isnan(b[k].float()).any()
    assert not torch.equal(a["logits"][4:], b["logits"][4:])
    assert abs(a["epistemic_uncertainty"][4:].mean().item() - b["epistemic_uncertainty"][4:].mean().item()) < \
        0.5 * a["epistemic_uncertainty"][4:].mean().item()
    # too few samples per chunk: stays on one stream, same result object layout
    s1 = predict_uncertainty([det], x[:3], n_pred=2, n_streams=2)
    s0 = predict_uncertainty([det], x[:3], n_pred=2, n_streams=1)
    assert torch.equal(s0["pred_entropy"], s1["pred_entropy"])
    # the same at 64^3, where the role-split kernels of every level run (16^3 only reaches the general tile kernel): chunks of
    # 2 volumes x 3 passes against 4 x 3 in one batch
    x64 = torch.from_numpy(np.concatenate([formula_volume((1, 1, 64, 64, 64), tag=140 + i) for i in range(4)], 0)).float().cuda()
    a = predict_uncertainty([det], x64, n_pred=3, n_streams=1)
    b = predict_uncertainty([det], x64, n_pred=3, n_streams=2)
    for k in keys:
        assert torch.equal(a[k], b[k]), k


def test_mode_switch_repacks_the_weights(vxcfg):
    """The packed weight layout belongs to the kernel family: switching VX_CONV_FP32 on a live model must re-pack
    (the cache is keyed on _lib.pack_mode()), not feed split-fp16 blocks to the native-fp32 kernels."""
    model = make_model(do_dropout=False)
    x = torch.from_numpy(formula_volume((2, 1, 16, 16, 16), tag=51)).float().cuda()
    vxcfg.delenv("VX_CONV_FP32", raising=False)
    a = model(x)
    vxcfg.setenv("VX_CONV_FP32", "1")
    b = model(x)
    vxcfg.delenv("VX_CONV_FP32", raising=False)
    c = model(x)
    assert (a - b).abs().max().item() < LOGIT_TOL
    assert torch.equal(a, c)


def test_hash_dropout_is_the_same_distribution_as_torch_dropout_16():
    """Our dropout bit generator cannot reproduce torch's CPU bernoulli stream ("Dropout RNG parity"),
    so compare DISTRIBUTIONS: the oracle with T_ref independent numpy-drawn Bernoulli(0.5) masks vs our hash
    dropout with T_ours samples.  Per voxel the two sample means differ by sampling error only:
    E[(m_ours - m_ref)^2] = var_v * (1/T_ours + 1/T_ref); the ratio of observed to predicted must be ~1."""
    from oracle.unet3d_oracle import DROPOUT_ORDER, unet3d_forward
    from oracle import uncertainty_oracle as uo
    from values_amd import predict_logits
    g = load_npz("unet3d_16.npz")
    T_ref, T_ours = 48, 768
    sd = formula_sd_torch()
    x = torch.from_numpy(g["input"]).double()
    rng = np.random.default_rng(1234)
    shapes = {n: tuple(int(v) for v in g[f"maskshape_{n}"]) for n in DROPOUT_ORDER}
    ref_p = []
    with torch.no_grad():
        for _ in range(T_ref):
            mk = {n: torch.from_numpy(rng.random(shapes[n]) < 0.5) for n in DROPOUT_ORDER}
            ref_p.append(uo.softmax(unet3d_forward(sd, x, masks=mk).numpy(), axis=1)[0, 1])
    ref_p = np.stack(ref_p)
    model = make_model(do_dropout=True)
    lg = predict_logits([model], x.float().cuda(), n_pred=T_ours, seeds=[9])
    ours_p = torch.softmax(lg[0].double(), 1)[:, 1].cpu().numpy()
    var = ours_p.var(axis=0) + 1e-12
    d2 = (ours_p.mean(0) - ref_p.mean(0)) ** 2
    ratio = float((d2 / var).mean() / (1.0 / T_ours + 1.0 / T_ref))
    assert 0.7 < ratio < 1.4, ratio
    # second moment too: per-voxel variances agree (F-ratio averaged over voxels ~ 1)
    vr = float((ref_p.var(axis=0, ddof=1) / ours_p.var(axis=0, ddof=1)).mean())
    assert 0.85 < vr < 1.15, vr
    # and the reference's own T=4 golden maps sit inside the same distribution
    assert abs(ours_p.mean() - g["mean_softmax"][1].mean()) < 0.01


def test_aleatoric_head_and_errors():
    from values_amd import UNet3D
    m = UNet3D(num_classes=2, aleatoric_loss=True).cuda()
    x = torch.from_numpy(formula_volume((1, 1, 16, 16, 16))).float().cuda()
    with torch.no_grad():
        mu, s = m(x)
    assert mu.shape == s.shape == (1, 2, 16, 16, 16)
    with pytest.raises(NotImplementedError):
        UNet3D(num_classes=2, in_channels=9)
    with pytest.raises(NotImplementedError):
        UNet3D(num_classes=2, kernel_size=5)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 2, 16, 16, 16).cuda())


def test_sharded_ensemble_statistics_equal_single_pass():
    """C3: members' sufficient statistics accumulated in ANY split and finalised == the one-pass reduction over the
    stacked members (what the reference computes), to float32 summation-order noise; and vs the golden ensemble."""
    from values_amd import predict_uncertainty
    from values_amd.dist import ensemble_uncertainty_sharded, ensemble_work_items, finalize_stats
    g = load_npz("ensemble_tta_16.npz")
    models = [make_model(seed_tag=s, do_dropout=False) for s in range(3)]
    x = torch.from_numpy(g["input"]).cuda()
    one = predict_uncertainty(models, x, n_pred=1)
    sh = ensemble_uncertainty_sharded(models, x, world=1, rank=0, n_pred=1)
    for k in KEYS:
        assert (sh[k] - one[k]).abs().max().item() < 2e-6, k
        assert np.abs(sh[k][0].cpu().numpy() - g["ens_" + k]).max() < MAP_TOL, k
    assert (sh["mean_softmax"] - one["mean_softmax"]).abs().max().item() < 1e-6
    # emulate a 2-rank run: each "rank" accumulates its own items, the buffers are summed (what dist.reduce does)
    from values_amd import _lib, predict_logits
    lib = _lib.load()
    parts = []
    for rank in range(2):
        st = torch.zeros((1, 3, 16, 16, 16), device="cuda")
        for (m, lo, hi) in ensemble_work_items(3, 1, 2)[rank]:
            lg = predict_logits([models[m]], x[lo:hi], n_pred=1)
            _lib.check(lib.vx_unc_stats_accumulate(_lib.ptr(lg), hi - lo, 1, 2, 16 ** 3, _lib.ptr(st[lo:hi]),
                                                   _lib.stream_ptr()), "acc")
        parts.append(st)
    two = finalize_stats(parts[0] + parts[1], 3)
    for k in KEYS:
        assert (two[k] - one[k]).abs().max().item() < 2e-6, k
    clear = g["mean_margin"] > 0  # ensemble argmax: compare against the one-pass kernel (same float32 means)
    assert torch.equal(two["pred_seg_mean"], one["pred_seg_mean"]) or clear.any()


def test_aleatoric_head_sampling_matches_reference_formula():
    """test_3D.py:458-469 restated: output = mu + exp(s/2) * eps, softmax, calculate_uncertainty -- with the SAME eps."""
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import unet3d_forward
    from values_amd import UNet3D, predict_uncertainty
    from tests.formula import formula_tensor
    sd = formula_unet3d_state_dict(seed_tag=4, aleatoric_loss=True)
    model = UNet3D(num_classes=2, aleatoric_loss=True)
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    model = model.cuda()
    x = torch.from_numpy(formula_volume((1, 1, 16, 16, 16), tag=44))
    Ts = 5
    eps = torch.from_numpy(formula_tensor((1, Ts, 2, 16, 16, 16), 45, scale=1.7))
    out = predict_uncertainty([model], x.float().cuda(), n_aleatoric_samples=Ts, eps=[eps])
    with torch.no_grad():
        mu, s = unet3d_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x, aleatoric_loss=True, num_classes=2)
    samples = mu.numpy()[0][None] + np.exp(s.numpy()[0][None] / 2) * eps.numpy()[0]  # (Ts, 2, ...)
    assert np.abs(out["logits"][0].cpu().numpy() - samples).max() < LOGIT_TOL
    ref = uo.calculate_uncertainty(uo.softmax(samples, axis=1))
    for k in KEYS:
        assert np.abs(out[k][0].cpu().numpy() - ref[k]).max() < MAP_TOL, k
    # generated noise: standard normal (mean 0, variance 1, no correlation between samples)
    lg = predict_uncertainty([model], x.float().cuda(), n_aleatoric_samples=64, seeds=[3])["logits"][0]
    z = ((lg - torch.from_numpy(mu.numpy()[0]).float().cuda()) / torch.exp(torch.from_numpy(s.numpy()[0]).float().cuda() / 2))
    assert abs(z.mean().item()) < 0.01 and abs(z.var().item() - 1) < 0.02
    assert abs((z[0] * z[1]).mean().item()) < 0.02
    assert abs(((z ** 4).mean().item()) - 3.0) < 0.1  # kurtosis of a Gaussian


def test_ssn_matches_reference_fixture_and_generated_noise_is_standard_normal():
    """SsnUNet3D (ssn_unet3D_module.py) + predict_cases_ssn sampling (test_3D.py:373-388): head, samples with the
    reference's captured normals, and the ssn=True maps; then the on-device generator's moments."""
    from values_amd import SsnUNet3D, predict_uncertainty
    from tests.formula import formula_ssn_state_dict
    from values_amd.io import instantiate
    g = load_npz("ssn_16.npz")
    NC, R, size = 2, 10, 16
    m = instantiate({"_target_": "uncertainty_modeling.models.ssn_unet3D_module.SsnUNet3D", "num_classes": NC, "rank": R})
    assert isinstance(m, SsnUNet3D)
    sd = {k: torch.from_numpy(v).float() for k, v in formula_ssn_state_dict(NC, R).items()}
    res = m.load_state_dict(sd, strict=True)          # the reference's key names, incl. the unused parent `final`
    assert not res.missing_keys and not res.unexpected_keys
    m = m.cuda()
    x = torch.from_numpy(g["input"]).cuda()
    dist = m(x)
    vox = size ** 3
    np.testing.assert_allclose(dist.mean.cpu().numpy(), g["loc"], atol=LOGIT_TOL)
    head = dist._head.cpu().numpy().reshape(1, (2 + R) * NC, vox)
    np.testing.assert_allclose(np.exp(head[:, NC:2 * NC]).reshape(1, -1) + 1e-5, g["cov_diag"], rtol=2e-4)
    fac = head[:, 2 * NC:].reshape(1, R, NC * vox).transpose(0, 2, 1)      # view/flatten/transpose of :52-55
    np.testing.assert_allclose(fac, g["cov_factor"], atol=LOGIT_TOL)
    S = g["samples"].shape[0]
    smp = dist.sample([S], eps_w=torch.from_numpy(g["eps_w"]), eps_d=torch.from_numpy(g["eps_d"]))
    assert tuple(smp.shape) == g["samples"].shape
    np.testing.assert_allclose(smp.cpu().numpy(), g["samples"], atol=2e-4)
    r = predict_uncertainty([m], x, n_pred=S, ssn=True, eps_w=torch.from_numpy(g["eps_w"]), eps_d=torch.from_numpy(g["eps_d"]))
    for k in KEYS:
        assert np.abs(r[k][0].cpu().numpy() - g[k]).max() < MAP_TOL, k
    # generated normals: first two moments over 64 draws, and the low-rank term is shared by all voxels of a draw
    big = dist.sample_volumes(64, seed=5)[0]                                 # (64, C, D,H,W)
    mean = dist.mean.reshape(NC, size, size, size)
    z = (big - mean[None]).reshape(64, -1)
    var_expected = torch.from_numpy(g["cov_diag"]).cuda().reshape(-1) + (torch.from_numpy(g["cov_factor"]).cuda()[0] ** 2).sum(1)
    ratio = (z.var(0, unbiased=True) / var_expected).mean().item()
    assert abs(ratio - 1.0) < 0.05, ratio
    # the diagonal part alone (mean_only: rank 0, ssn_unet3D_module.py:49-50): element-wise standard normals
    d0 = m(x, mean_only=True)
    z0 = (d0.sample_volumes(64, seed=9)[0] - mean[None]).reshape(64, -1) / torch.from_numpy(g["cov_diag"]).cuda().reshape(-1).sqrt()
    assert abs(z0.mean().item()) < 0.01 and abs(z0.var().item() - 1.0) < 0.01
    assert abs((z0 ** 4).mean().item() - 3.0) < 0.1       # kurtosis of a normal
    assert not torch.equal(dist.sample_volumes(2, seed=5)[0, 0], dist.sample_volumes(2, seed=6)[0, 0])


def test_host_pipeline_returns_every_step_in_order():
    """values_amd.HostPipeline (pinned host in, pinned host out, copies on their own streams, a step's download enqueued
    after the next step's upload): same maps as predict_uncertainty + .cpu(), every step exactly once, in order."""
    from values_amd import HostPipeline, predict_uncertainty
    model = make_model(do_dropout=True)
    xs = [torch.from_numpy(np.concatenate([formula_volume((1, 1, 16, 16, 16), tag=60 + 3 * s + i) for i in range(3)], 0)).float()
          for s in range(5)]
    hp = HostPipeline([model], n_pred=4)
    got = []
    for s, xh in enumerate(xs):
        r = hp.submit(xh, seeds=[s])
        if r is not None:
            got.append({k: v.copy() for k, v in r.items()})     # views of pinned buffers: valid until the next submit
    got += [{k: v.copy() for k, v in r.items()} for r in hp.flush()]
    assert len(got) == len(xs)
    for s, xh in enumerate(xs):
        ref = predict_uncertainty([model], xh.cuda(), n_pred=4, seeds=[s])
        for k in HostPipeline.KEYS:
            assert np.array_equal(got[s][k], ref[k].cpu().numpy()), (s, k)
    assert hp.flush() == []


def _oracle_maps(sd, x, masks_per_pass):
    """float64 restatement of predict_cases + calculate_uncertainty for one volume: logits (T, C, ...) and the maps"""
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import DROPOUT_ORDER, unet3d_forward
    logits = []
    with torch.no_grad():
        for mk in masks_per_pass:
            logits.append(unet3d_forward(sd, x, masks=None if mk is None else dict(zip(DROPOUT_ORDER, mk)))[0].numpy())
    logits = np.stack(logits)
    return logits, uo.calculate_uncertainty(uo.softmax(logits, axis=1))


def test_timed_instances_at_64_vs_oracle_with_exported_hash_masks():
    """BASELINE config C2's shape through the kernels the bench times: ONE 64^3 volume, T = 2 MC-dropout samples on
    the production path (hash dropout -> the large-tile double-buffered instances with compile-time epilogues
    <8,1,16,8,4,8,1,{2,3},{0,1,2}>, head fused into expand_1_2, shared first layer + fan-out).  The bit generator's
    masks are exported (vx_drop_hash_mask) and handed to the float64 oracle, so logits and maps compare at the north
    star's tolerance.  Then the same volume with dropout off."""
    from values_amd import predict_uncertainty
    S, T, seed = 64, 2, 4242
    model = make_model(do_dropout=True)
    x = torch.from_numpy(formula_volume((1, 1, S, S, S), tag=71))
    out = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, T, S, S, S)]
    keep = float(np.mean([m.float().mean().item() for m in masks]))
    assert 0.49 < keep < 0.51, keep
    sd = formula_sd_torch()
    logits, ref = _oracle_maps(sd, x, [[m[t:t + 1] for m in masks] for t in range(T)])
    got = out["logits"][0].cpu().numpy()
    assert_close(np.abs(got - logits).max(), LOGIT_TOL, REG_LOGIT_TOL, "logits")
    for k in KEYS:
        assert_close(np.abs(out[k][0].cpu().numpy() - ref[k]).max(), MAP_TOL, REG_MAP_TOL, k)
    # replaying the exported masks through VX_DROP_MASK (run-time epilogue instances) gives the same logits
    rep = predict_uncertainty([model], x.float().cuda(), n_pred=T, dropout_masks=[masks])
    assert (rep["logits"] - out["logits"]).abs().max().item() < 2e-5
    # Round 5: the level-1 layers ran on the role-split z-column kernel (conv3d_zc16.hip) with the level-0 data flow (raw skip
    # half, pooled epilogue + vx_pool_finish_z, expand_2_1 normalising on load); the knob that switches it off runs the round-4
    # launches (tile kernels + the normalise / pool pass) on the same seeds: the same function, compared with the oracle too
    from values_amd import _lib
    import bench
    names = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    assert any(n.startswith("conv3d_zc16_kernel<16,4,1,0,0>") for n in names) and any(n.startswith("conv3d_zc16_kernel<8,0,3,0,0>") for n in names), names
    assert any(n.startswith("pool_finish_z_kernel<true>") for n in names) and len(names) <= 30      # (three of the eight finalize launches ride in their consumers)
    assert sum(n.startswith("instnorm_finalize_kernel") for n in names) == 5 and sum(n.startswith("norm_act_drop_pool_kernel<true,false,true>") for n in names) == 2
    # expand_2_1 as two launches over its halves, upscale3 inside; round 6: its output leaves as the planar pre-split tensor (EPI 5)
    # and expand_2_2 stages it by LDS-DMA (PRE 4)
    assert any(n.startswith("conv3d_zc16_kernel<16,5,0,1,1>") for n in names) and any(n.startswith("conv3d_zc16_kernel<16,1,4,0,0>") for n in names), names
    with _lib.config(s16_no_l1dma=1):        # ... the float hand-over staged through registers: the same bits
        flt = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
        names6 = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    assert any(n.startswith("conv3d_zc16_kernel<16,1,0,1,1>") for n in names6) and any(n.startswith("conv3d_zc16_kernel<16,1,0,0,0>") for n in names6), names6
    assert torch.equal(flt["logits"], out["logits"])
    assert not any(n.startswith("convT_k2s2_mfma_kernel<32,") for n in names), names           # (no upscale3 launch)
    with _lib.config(s16_no_upfuse=1):       # ... separate upscale launches (upscale2 and upscale3)
        sep = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    assert np.abs(sep["logits"][0].cpu().numpy() - logits).max() < LOGIT_TOL
    assert (sep["logits"] - out["logits"]).abs().max().item() < 3e-5
    with _lib.config(s16_no_halves=1):       # ... or as one launch of the tile kernel over the x-blocked buffer (skip half normalised on load)
        one = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
        names1 = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    assert not any(n.startswith("conv3d_zc16_kernel<16,1,0,1,") or n.startswith("conv3d_zc16_kernel<16,5,0,1,") for n in names1) and len(names1) == len(names)    # (+ upscale3, - one half)
    assert np.abs(one["logits"][0].cpu().numpy() - logits).max() < LOGIT_TOL
    assert (one["logits"] - out["logits"]).abs().max().item() < 2e-5
    with _lib.config(s16_no_zc16=1):
        alt = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
        names4 = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    assert not any("zc16" in n for n in names4)
    assert np.abs(alt["logits"][0].cpu().numpy() - logits).max() < LOGIT_TOL
    assert (alt["logits"] - out["logits"]).abs().max().item() < 2e-5
    for k in KEYS:
        assert np.abs(alt[k][0].cpu().numpy() - ref[k]).max() < MAP_TOL, k
    # Round 5: the 16^3 / 8^3 layers (Cout = 32 / 64) ran on the role-split kernel of the deep layers (conv3d_deep.hip: statistics,
    # normalise-on-load, the x-blocked concat input, LeakyReLU + dropout (+ the pre-split hand-over to the fused upscale3)); the knob
    # runs the tile kernel on them
    import re
    count = lambda pat: sum(re.match(pat, n) is not None for n in names)      # (tile size 4 / 2: picked by the number of workgroups it fills)
    assert count(r"conv3d_deep_kernel<[24],0,0>") == 2 and count(r"conv3d_deep_kernel<[24],0,1>") == 2, names
    assert count(r"conv3d_deep_kernel<[24],1,0>") == 4, names
    assert any(n.startswith("convT_k2s2_s16_kernel<64,4,false>") for n in names) and any(n.startswith("convT_k2s2_s16_kernel<128,2,false>") for n in names), names
    with _lib.config(s16_no_deep=1):
        alt = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
        names5 = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    assert not any("deep" in n for n in names5) and len(names5) == len(names)
    assert np.abs(alt["logits"][0].cpu().numpy() - logits).max() < LOGIT_TOL
    assert (alt["logits"] - out["logits"]).abs().max().item() < 2e-5
    for k in KEYS:
        assert np.abs(alt[k][0].cpu().numpy() - ref[k]).max() < MAP_TOL, k
    # dropout off: the plain-epilogue instances, EPI = 0 / run-time activation
    det = make_model(do_dropout=False)
    with torch.no_grad():
        y = det(x.float().cuda())
    l0, _ = _oracle_maps(sd, x, [None])
    assert_close(np.abs(y[0].cpu().numpy() - l0[0]).max(), LOGIT_TOL, REG_LOGIT_TOL, "logits, dropout off")


def test_bench_geometry_two_volumes_T10_at_64_vs_oracle():
    """BASELINE config C2 at the BENCH's geometry against the oracle: V = 2 volumes x T = 10 MC-dropout samples at 64^3 on the default
    (timed) kernels, the hash generator's masks exported and handed to the float64 oracle (20 CPU passes).  The only place where
    contr_1_2's (volume, column, sample) column order (round 5: it differs from the plain order only for V > 1, T > 2) and
    in_repeat = 10 meet the oracle; the 1-volume T = 2 test above cannot see either."""
    from values_amd import predict_uncertainty
    import bench
    S, T, V, seed = 64, 10, 2, 777
    model = make_model(do_dropout=True)
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1, S, S, S), tag=171 + v) for v in range(V)], 0))
    out = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    names = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    assert any(n.startswith("conv3d_xp8w_kernel<1,4,2,") for n in names), names          # contr_1_2 on the pre-split shared tensor
    assert any(n.startswith("conv3d_xp8w_kernel<2,1,1,2,8") for n in names), names        # upscale2 composed into expand_1_1
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, V * T, S, S, S)]             # sample n = volume * T + pass
    sd = formula_sd_torch()
    for v in range(V):
        logits, ref = _oracle_maps(sd, x[v:v + 1], [[m[v * T + t:v * T + t + 1] for m in masks] for t in range(T)])
        assert_close(np.abs(out["logits"][v].cpu().numpy() - logits).max(), LOGIT_TOL, REG_LOGIT_TOL, ("logits", v))
        for k in KEYS:
            assert_close(np.abs(out[k][v].cpu().numpy() - ref[k]).max(), MAP_TOL, REG_MAP_TOL, (k, v))


def test_a_samples_bits_do_not_depend_on_the_batch_size_at_64():
    """InstanceNorm is per sample and every kernel's tile geometry is a function of the volume's shape only: a 64^3 sample computes
    the same BITS alone, among 4 and among 65 (round-5 advice: conv3d_deep.hip picked its tile from how the batch filled the
    workgroups and ran the 4^3 layers on another kernel when N % 4 != 0 -- 65 leaves one sample in the last four-sample tile)."""
    model = make_model(do_dropout=False)
    g = torch.Generator(device="cpu").manual_seed(65)
    x = torch.randn((65, 1, 64, 64, 64), generator=g).cuda()
    with torch.no_grad():
        y65 = model(x)
        y4 = model(x[:4])
        y1 = model(x[:1])
        ylast = model(x[64:65])
        y3 = model(x[61:64])
    assert torch.equal(y65[:4], y4) and torch.equal(y4[:1], y1)
    assert torch.equal(y65[64:65], ylast) and torch.equal(y65[61:64], y3)


def test_config_C3_full_size_five_members_64_vs_oracle():
    """BASELINE config C3 as worded: a 5-member deep ensemble on 64^3 volumes (n_pred = 1, no dropout).  One volume
    against the float64 oracle (5 passes), a second one for the batch path; the member-sharded statistics path
    (what 8 ranks run, here dealt over 8 emulated ranks and summed) against the one-pass reduction."""
    from values_amd import _lib, predict_logits, predict_uncertainty
    from values_amd.dist import ensemble_uncertainty_sharded, ensemble_work_items, finalize_stats
    S, M = 64, 5
    models = [make_model(seed_tag=s, do_dropout=False) for s in range(M)]
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1, S, S, S), tag=81 + i) for i in range(2)], 0))
    out = predict_uncertainty(models, x.float().cuda(), n_pred=1)
    assert out["logits"].shape == (2, M, 2, S, S, S)
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import unet3d_forward
    with torch.no_grad():
        lg = np.stack([unet3d_forward(formula_sd_torch(seed_tag=s), x[:1])[0].numpy() for s in range(M)])
    assert_close(np.abs(out["logits"][0].cpu().numpy() - lg).max(), LOGIT_TOL, REG_LOGIT_TOL, "logits")
    ref = uo.calculate_uncertainty(uo.softmax(lg, axis=1))
    for k in KEYS:
        assert_close(np.abs(out[k][0].cpu().numpy() - ref[k]).max(), MAP_TOL, REG_MAP_TOL, k)
    sh = ensemble_uncertainty_sharded(models, x.float().cuda(), world=1, rank=0, n_pred=1)
    for k in KEYS + ("mean_softmax",):
        assert (sh[k] - out[k]).abs().max().item() < 2e-6, k
    # 8 ranks: (member, volume block) items dealt round robin, each rank's statistics summed as dist.reduce does
    lib = _lib.load()
    xd = x.float().cuda()
    total = torch.zeros((2, 3, S, S, S), device="cuda")
    items = ensemble_work_items(M, 2, 8)
    assert sum(len(it) for it in items) == 10 and all(len(it) >= 1 for it in items)
    for rank in range(8):
        st = torch.zeros_like(total)
        for (m, lo, hi) in items[rank]:
            l1 = predict_logits([models[m]], xd[lo:hi], n_pred=1).contiguous()
            _lib.check(lib.vx_unc_stats_accumulate(_lib.ptr(l1), hi - lo, 1, 2, S ** 3, _lib.ptr(st[lo:hi]),
                                                   _lib.stream_ptr()), "acc")
        total += st
    eight = finalize_stats(total, M)
    for k in KEYS:
        assert (eight[k] - out[k]).abs().max().item() < 2e-6, k
        assert np.abs(eight[k][0].cpu().numpy() - ref[k]).max() < MAP_TOL, k


def test_variance_rides_in_the_reduction_pass():
    """north_star's fourth map: predict_uncertainty returns softmax_variance from the same pass over the logits
    (vx_unc_reduce_ex).  No reference counterpart -> checked against numpy in float64 and against the
    stand-alone kernel."""
    from values_amd import predict_uncertainty, softmax_variance
    model = make_model(do_dropout=True)
    x = torch.from_numpy(np.concatenate([formula_volume((1, 1, 16, 16, 16), tag=90 + i) for i in range(3)], 0)).float().cuda()
    out = predict_uncertainty([model], x, n_pred=6, seeds=[3])
    p = torch.softmax(out["logits"].double(), 2).cpu().numpy()          # (V, T, C, ...)
    want = p.var(axis=1).mean(axis=1)
    got = out["softmax_variance"].cpu().numpy()
    assert np.abs(got - want).max() < 1e-7, np.abs(got - want).max()
    alone = softmax_variance(out["logits"], from_logits=True).cpu().numpy()
    assert np.abs(alone - want).max() < 1e-6
    assert got.min() >= 0
    # where the samples agree to 1e-4 the shifted accumulation keeps the relative error small
    small = want < 1e-8
    if small.any():
        assert np.abs(got[small] - want[small]).max() < 1e-10


def test_fp16_range_guard_never_returns_nan_maps():
    """Decoder blocks have no norm (unet3D_module.py:263-267), so a checkpoint can drive their activations past what the
    split-fp16 convolutions represent (65504).  Here the center's transposed conv is scaled up until that happens: the
    kernels' range word reports it, "raise" raises, the default re-runs the batch on the native-fp32 kernels -- finite
    maps equal to the float64 oracle's either way."""
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import unet3d_forward
    from values_amd import UNet3D, _lib, predict_uncertainty
    sd = formula_unet3d_state_dict(seed_tag=2)
    sd["center.4.weight"] = sd["center.4.weight"] * 3e7
    model = UNet3D(num_classes=2, do_dropout=False)
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    model = model.cuda()
    x = torch.from_numpy(formula_volume((2, 1, 16, 16, 16), tag=95))
    with torch.no_grad():
        taps = {}
        ref = unet3d_forward({k: torch.from_numpy(v).float().double() for k, v in sd.items()}, x, taps=taps)
    assert taps["center"].abs().max().item() > 65504          # the oracle's own activations are past the limit
    with pytest.raises(_lib.VxError):
        predict_uncertainty([model], x.float().cuda(), n_pred=1, range_check="raise")
    out = predict_uncertainty([model], x.float().cuda(), n_pred=1)           # default: fallback to the fp32 kernels
    assert torch.isfinite(out["logits"]).all()
    lg = out["logits"][:, 0].cpu().double()
    assert ((lg - ref).abs() / (1.0 + ref.abs())).max().item() < 1e-4
    sm = uo.softmax(ref.numpy()[:, None], axis=2)
    for v in range(2):
        r = uo.calculate_uncertainty(sm[v])
        assert np.abs(out["pred_entropy"][v].cpu().numpy() - r["pred_entropy"]).max() < MAP_TOL
    # an ordinary checkpoint stays far below the limit and takes no second pass
    ok = make_model(do_dropout=True)
    predict_uncertainty([ok], x.float().cuda(), n_pred=3, seeds=[1], range_check="off")
    assert ok.range_max() == 0.0        # magnitudes below 32768 are not reported


def _overflowing_model(do_dropout=False, **kw):
    """a checkpoint whose center drives the decoder past the fp16 limit (see the test above)"""
    from values_amd import UNet3D
    sd = formula_unet3d_state_dict(seed_tag=2)
    sd["center.4.weight"] = sd["center.4.weight"] * 3e7
    model = UNet3D(num_classes=2, do_dropout=do_dropout, **kw)
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    return model.cuda(), sd


def test_fp16_range_guard_covers_every_driver():
    """the guard lived in predict_uncertainty only.  The same overflowing checkpoint through the sliding
    window driver (the reference's main test_3D path), the member-sharded ensemble, the captured graph and the host
    pipeline: each either raises or returns the native-fp32 result -- finite, equal to predict_uncertainty's fallback --
    and the fallback replays the SAME dropout bits when the caller gave no seeds."""
    from values_amd import GraphedPredictor, HostPipeline, _lib, predict_uncertainty
    from values_amd.dist import ensemble_uncertainty_sharded
    from values_amd.sliding import predict_image_sliding
    model, _ = _overflowing_model()
    x = torch.from_numpy(formula_volume((2, 1, 16, 16, 16), tag=95)).float().cuda()
    want = predict_uncertainty([model], x, n_pred=1)                       # fallback result (checked against the oracle above)
    assert torch.isfinite(want["logits"]).all()
    # sliding window: one 16^3 patch per image
    with pytest.raises(_lib.VxError):
        predict_image_sliding([model], x[0, 0], patch_size=16, patch_overlap=1, n_pred=1, range_check="raise")
    sl = predict_image_sliding([model], x[0, 0], patch_size=16, patch_overlap=1, n_pred=1)
    assert torch.isfinite(sl["pred_entropy"]).all()
    assert (sl["pred_entropy"] - want["pred_entropy"][0]).abs().max().item() < 2e-6
    assert torch.equal(sl["pred_seg_mean"], want["pred_seg_mean"][0])
    # member-sharded statistics path
    with pytest.raises(_lib.VxError):
        ensemble_uncertainty_sharded([model], x, world=1, rank=0, n_pred=1, range_check="raise")
    sh = ensemble_uncertainty_sharded([model], x, world=1, rank=0, n_pred=1)
    assert (sh["pred_entropy"] - want["pred_entropy"]).abs().max().item() < 2e-6
    # captured graph: the weights of the split-fp16 family stay alive while the fallback packs the fp32 family
    gp = GraphedPredictor([model], tuple(x.shape), n_pred=1)
    g = gp(x, check=True)
    assert (g["pred_entropy"] - want["pred_entropy"]).abs().max().item() < 2e-6
    raw = gp(x)                                                            # unchecked replay still runs (pointers valid) ...
    with pytest.raises(_lib.VxError):
        gp.check_range()                                                   # ... and the word says what happened
    # host pipeline: the word travels with the maps, no read on the submit path
    hp = HostPipeline([model], n_pred=1)
    outs = [hp.submit(x.cpu()) for _ in range(4)] + hp.flush()
    outs = [o for o in outs if o is not None]
    assert len(outs) == 4
    for o in outs:
        assert np.isfinite(o["pred_entropy"]).all()
        assert np.abs(o["pred_entropy"] - want["pred_entropy"].cpu().numpy()).max() < 2e-6
    hp = HostPipeline([model], n_pred=1, range_check="raise")
    with pytest.raises(_lib.VxError):
        [hp.submit(x.cpu()) for _ in range(4)], hp.flush()
    # un-seeded MC-dropout: the second (fp32) run draws the bits the first one drew
    drop, _ = _overflowing_model(do_dropout=True)
    calls = drop._calls
    a = predict_uncertainty([drop], x, n_pred=3)
    seed = (drop.seed * 1000003 + calls) & 0xFFFFFFFF
    with _lib.config(conv_fp32=1):
        b = predict_uncertainty([drop], x, n_pred=3, seeds=[seed], range_check="off")
    assert torch.equal(a["logits"], b["logits"])


def test_fp16_range_guard_without_instancenorm_covers_the_first_block():
    """do_instancenorm=False: nothing normalises the first layer's output on its way into contr_1_2 (a split-fp16 conv) --
    the activation / dropout pass records the range there (vx_norm_args.range_flag)."""
    from oracle.unet3d_oracle import unet3d_forward
    from values_amd import UNet3D, _lib, predict_uncertainty
    sd = formula_unet3d_state_dict(seed_tag=3)
    model = UNet3D(num_classes=2, do_dropout=False, do_instancenorm=False)
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()}, strict=False)
    model = model.cuda()
    x = torch.from_numpy(formula_volume((1, 1, 16, 16, 16), tag=96)) * 4e6        # raw intensities of any scale are legal input
    with pytest.raises(_lib.VxError):
        predict_uncertainty([model], x.float().cuda(), n_pred=1, range_check="raise")
    out = predict_uncertainty([model], x.float().cuda(), n_pred=1)
    assert torch.isfinite(out["logits"]).all()
    with torch.no_grad():
        ref = unet3d_forward({k: torch.from_numpy(v).float().double() for k, v in sd.items()}, x.float().double(),
                             instancenorm=False)
    lg = out["logits"][:, 0].cpu().double()
    assert ((lg - ref).abs().max() / ref.abs().max()).item() < 1e-5        # logits of magnitude 1e4: float32 arithmetic


def test_graphed_predictor_replays_equal_eager_and_draw_fresh_dropout():
    """values_amd.GraphedPredictor: forward + reduction captured into one hipGraph; a replay with device seed word s gives
    the bits of the eager call with the same effective seed, another word other dropout samples, another input its maps"""
    from values_amd import GraphedPredictor, predict_uncertainty
    model = make_model(do_dropout=True)
    x1 = torch.from_numpy(formula_volume((1, 1, 32, 32, 32), tag=97)).float().cuda()
    x2 = torch.from_numpy(formula_volume((1, 1, 32, 32, 32), tag=98)).float().cuda()
    gp = GraphedPredictor([model], x1.shape, n_pred=5)
    a = {k: v.clone() for k, v in gp(x1, seed=7).items()}
    b = {k: v.clone() for k, v in gp(x1, seed=7).items()}
    c = {k: v.clone() for k, v in gp(x1, seed=8).items()}
    d = {k: v.clone() for k, v in gp(x2, seed=7).items()}
    for k in KEYS + ("softmax_variance", "mean_softmax", "pred_seg_mean"):
        assert torch.equal(a[k], b[k]), k
    assert not torch.equal(a["epistemic_uncertainty"], c["epistemic_uncertainty"])
    assert not torch.equal(a["pred_entropy"], d["pred_entropy"])
    e = predict_uncertainty([model], x1, n_pred=5, seeds=[1000003 + 7])        # by-value seed + device word
    for k in KEYS + ("mean_softmax",):
        assert torch.equal(a[k], e[k]), k
    det = make_model(do_dropout=False)
    gd = GraphedPredictor([det], x1.shape, n_pred=1)
    assert torch.equal(gd(x2)["logits"], predict_uncertainty([det], x2, n_pred=1)["logits"])


@pytest.mark.parametrize("knobs", [dict(s16_skip_raw=0), dict(s16_no_prenorm=1), dict(s16_no_xp8=1), dict(s16_no_upfuse=1), dict(s16_no_upcompose=1), dict(no_head_fusion=1), dict(s16_no_poolfuse=1), dict(s16_no_poolfin=1), dict(s16_no_presplit=1), dict(s16_no_dbplain=1), dict(s16_no_upsplit=1), dict(s16_no_deep=1), dict(s16_no_l1dma=1)])
def test_level0_fusion_variants_vs_oracle_32(knobs, vxcfg):
    """The level-0 data-flow variants behind vx_config: a separate normalise pass for the skip half instead of expand_1_1
    normalising the raw tensor on load (s16_skip_raw=0), no normalise-on-load at all (s16_no_prenorm), the general tile
    kernels instead of the z-column kernel (s16_no_xp8), a separate upscale2 launch (s16_no_upfuse), the 1x1x1 head as its own
    launch (no_head_fusion), every sample normalising the shared first-layer tensor itself instead of masking the
    once-per-volume output of vx_prenorm_split (s16_no_presplit), a separate
    pooling pass instead of the window maxima from contr_1_2's epilogue (s16_no_poolfuse) -- each
    against the float64 oracle with the exported hash masks, MC-dropout (shared first layer) and TTA-style (src / flip)
    batches."""
    from values_amd import predict_uncertainty
    vxcfg.set(**knobs)
    S, T, seed = 32, 3, 77
    model = make_model(do_dropout=True)
    x = torch.from_numpy(formula_volume((1, 1, S, S, S), tag=73))
    out = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, T, S, S, S)]
    logits, ref = _oracle_maps(formula_sd_torch(), x, [[m[t:t + 1] for m in masks] for t in range(T)])
    assert np.abs(out["logits"][0].cpu().numpy() - logits).max() < LOGIT_TOL
    for k in KEYS:
        assert np.abs(out[k][0].cpu().numpy() - ref[k]).max() < MAP_TOL, k
    # the un-shared first layer (per-sample src / flip: the TTA path) with dropout off
    det = make_model(do_dropout=False)
    a = predict_uncertainty([det], x.float().cuda(), tta=True, x_noise=x.float().cuda() * 1.01)
    vxcfg.set(s16_skip_raw=1, s16_no_prenorm=0, s16_no_xp8=0, s16_no_upfuse=0, s16_no_upcompose=0, no_head_fusion=0, s16_no_poolfuse=0, s16_no_poolfin=0, s16_no_presplit=0, s16_no_dbplain=0, s16_no_upsplit=0, s16_no_deep=0, s16_no_l1dma=0)     # the defaults
    b = predict_uncertainty([det], x.float().cuda(), tta=True, x_noise=x.float().cuda() * 1.01)
    assert (a["logits"] - b["logits"]).abs().max().item() < 2e-5


@pytest.mark.parametrize("in_channels,instancenorm,size", [(1, False, 16), (3, True, 16), (2, False, 32), (4, True, 32)])
def test_ctor_variants_vs_oracle(in_channels, instancenorm, size):
    """UNet3D(in_channels > 1) and UNet3D(do_instancenorm=False) (unet3D_module.py:8-35, 238-243): state-dict names and
    shapes of the reference class, MC-dropout (masks injected and hash masks exported) and the per-sample src / flip path
    against the float64 oracle."""
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import DROPOUT_ORDER, unet3d_forward
    from values_amd import UNet3D, predict_uncertainty
    from tests.formula import unet3d_param_shapes
    sd = formula_unet3d_state_dict(seed_tag=6, in_channels=in_channels)
    model = UNet3D(num_classes=2, in_channels=in_channels, do_instancenorm=instancenorm, do_dropout=True)
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == dict(unet3d_param_shapes(in_channels=in_channels))
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    model = model.cuda()
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    V, T, seed = 2, 3, 31
    x = torch.from_numpy(np.stack([formula_volume((in_channels, size, size, size), tag=200 + v) for v in range(V)]))
    out = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, V * T, size, size, size)]
    for v in range(V):
        lg = []
        with torch.no_grad():
            for t in range(T):
                n = v * T + t
                mk = {name: masks[i][n:n + 1] for i, name in enumerate(DROPOUT_ORDER)}
                lg.append(unet3d_forward(sdt, x[v:v + 1], masks=mk, instancenorm=instancenorm)[0].numpy())
        lg = np.stack(lg)
        assert np.abs(out["logits"][v].cpu().numpy() - lg).max() < LOGIT_TOL, (v, np.abs(out["logits"][v].cpu().numpy() - lg).max())
        ref = uo.calculate_uncertainty(uo.softmax(lg, axis=1))
        for k in KEYS:
            assert np.abs(out[k][v].cpu().numpy() - ref[k]).max() < MAP_TOL, (k, v)
    # injected masks (general kernels, separate passes) give the hash run's logits
    rep = predict_uncertainty([model], x.float().cuda(), n_pred=T, dropout_masks=[masks])
    assert (rep["logits"] - out["logits"]).abs().max().item() < 2e-5
    # TTA views: per-sample source volume and flips on the multi-channel input; dropout off
    det = UNet3D(num_classes=2, in_channels=in_channels, do_instancenorm=instancenorm, do_dropout=False)
    det.load_state_dict({k: torch.from_numpy(v).float() for k, v in sd.items()})
    det = det.cuda()
    xn = x * 1.01
    tta = predict_uncertainty([det], x.float().cuda(), tta=True, x_noise=xn.float().cuda())
    from values_amd.predict import FLIP_DIMS
    with torch.no_grad():
        for vi, (src, dims) in enumerate([(x, None), (x, FLIP_DIMS[2]), (xn, FLIP_DIMS[6])]):
            k = [0, 3, 8 + 7][vi]
            xi = src[:1].float().double()
            y = unet3d_forward(sdt, torch.flip(xi, dims) if dims else xi, instancenorm=instancenorm)
            y = torch.flip(y, dims) if dims else y
            assert (tta["logits"][0, k].cpu().double() - y[0]).abs().max().item() < LOGIT_TOL, vi


def test_storage16_mode_reports_its_deviation_and_leaves_the_default_alone(vxcfg):
    """Opt-in reduced-storage mode (vx_config.storage16, BASELINE config 2 says "bf16"; SURVEY D7: 16-bit storage cannot meet
    1e-4 on the entropy maps): expand_1_1 hands its full-resolution tensor to expand_1_2 as fp16.  The test MEASURES what the
    mode costs against the float64 oracle (same exported hash masks), requires it to be small but does not pretend it
    meets the parity bar, and checks that the default path is bit for bit what it was."""
    from values_amd import _lib, predict_uncertainty
    S, T, seed = 32, 3, 91
    model = make_model(do_dropout=True)
    x = torch.from_numpy(formula_volume((1, 1, S, S, S), tag=74))
    base = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    vxcfg.set(storage16=1)
    red = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    assert _lib.load().vx_last_kernel_name is not None
    vxcfg.set(storage16=0)
    again = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    assert torch.equal(again["logits"], base["logits"])                       # the default path is untouched
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, T, S, S, S)]
    logits, ref = _oracle_maps(formula_sd_torch(), x, [[m[t:t + 1] for m in masks] for t in range(T)])
    assert np.abs(base["logits"][0].cpu().numpy() - logits).max() < LOGIT_TOL
    d_logit = np.abs(red["logits"][0].cpu().numpy() - logits).max()
    d_maps = {k: float(np.abs(red[k][0].cpu().numpy() - ref[k]).max()) for k in KEYS}
    flips = int((red["pred_seg_mean"] != base["pred_seg_mean"]).sum().item())
    print(f"storage16: max|d| logits {d_logit:.2e}, maps {d_maps}, argmax flips {flips} of {S ** 3}")
    assert d_logit > 0.0                                                      # the mode really ran (fp16 rounding is visible)
    assert d_logit < 2e-2 and all(v < 5e-3 for v in d_maps.values())          # ... and stays a rounding effect
    assert flips <= S ** 3 // 500


def test_fp16_products_mode_reports_its_deviation_at_64(vxcfg):
    """vx_config.storage16 = 2 (round 6; what BASELINE config 2 calls "bf16"): besides the fp16 tensor of mode 1 the three
    full-resolution launches run ONE fp16 product per fp32 product (vx_conv3d_args.products = 1: instances <1,4,2,0,4,2>,
    <2,1,1,2,8,2>, <1,2,0,0,4,3>).  Like mode 1 it is measured against the float64 oracle through the exported hash masks, has to
    stay a rounding effect, does not pretend to meet the parity bar, and leaves the default path's bits alone.  64^3: the
    smallest volume at which the z-column kernels that carry the mode run."""
    from values_amd import _lib, predict_uncertainty
    import bench
    S, T, seed = 64, 2, 4242
    model = make_model(do_dropout=True)
    x = torch.from_numpy(formula_volume((1, 1, S, S, S), tag=71))
    base = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    vxcfg.set(storage16=2)
    red = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    names = [r[1] for r in bench.profiled_forward(model, x.float().cuda(), T, seed)]
    for inst in ("conv3d_xp8w_kernel<1,4,2,0,4,2>", "conv3d_xp8w_kernel<2,1,1,2,8,2>", "conv3d_xp8w_kernel<1,2,0,0,4,3>"):
        assert any(n.startswith(inst) for n in names), (inst, names)
    vxcfg.set(storage16=0)
    again = predict_uncertainty([model], x.float().cuda(), n_pred=T, seeds=[seed])
    assert torch.equal(again["logits"], base["logits"])
    masks = [m.cpu() for m in model.hash_dropout_masks(seed, T, S, S, S)]
    logits, ref = _oracle_maps(formula_sd_torch(), x, [[m[t:t + 1] for m in masks] for t in range(T)])
    d_logit = np.abs(red["logits"][0].cpu().numpy() - logits).max()
    d_maps = {k: float(np.abs(red[k][0].cpu().numpy() - ref[k]).max()) for k in KEYS}
    flips = int((red["pred_seg_mean"] != base["pred_seg_mean"]).sum().item())
    print(f"fp16 products: max|d| logits {d_logit:.2e}, maps {d_maps}, argmax flips {flips} of {S ** 3}")
    assert d_logit > 1e-5                                                     # the mode really ran
    assert d_logit < 5e-2 and all(v < 1e-2 for v in d_maps.values())          # ... and stays a rounding effect
    assert flips <= S ** 3 // 200
