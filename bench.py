#!/usr/bin/env python3
"""Headline benchmark: uncertainty-volumes/sec at 64^3, T=10 MC-dropout (BASELINE.json config C2).

    python bench.py --gpus N --steps K --warmup W            (N > 1 with WORLD_SIZE unset: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one batch of V synthetic 64^3 1-channel volumes per GPU through the whole hot path:
T=10 MC-dropout forwards of UNet3D (batched on device as V*T samples, hash dropout, float32 tensors, the 3x3x3
products evaluated fp32-accurately on the f16 matrix cores by operand splitting) -> logits in their pred_idx slots ->
fused softmax / entropy / MI / variance / argmax reduction.
Inputs are resident in HBM before the timed region.  Volumes are sharded over ranks (weak scaling: V per GPU fixed),
no collective on the COMPUTE path; at N > 1 every step's maps are collected on rank 0 by an overlapped RCCL gather
INSIDE the timed region ("maps resident in rank-0 device memory after the gather") -- `value` is that rate;
the rate with the maps left on their ranks is printed next to it as `no_gather` (--no-gather makes it the only one).

--config C3: the 5-member deep ensemble of BASELINE config 3, (member, volume block) items dealt over the ranks, one
RCCL sum-reduce of sufficient statistics per step (values_amd.dist.ensemble_uncertainty_sharded).
--config C4: HRNet-W18 at 1024x512, 8 TTA views per image (BASELINE config 4), images sharded over the ranks.
--config C5: 128^3 images through the sliding-window path (64^3 patches, overlap 0.5, T = 20; BASELINE config 5).

Rank 0 prints ONE JSON line; it carries `roofline` (dominant kernel, live HIP-event timing, algorithmic FLOPs and
bytes, the binding roof named by max(flops / peak, bytes / bandwidth)) and, at N=1, `cpu_baseline` (the oracle =
PyTorch-CPU restatement of test_3D.py's float64 loop, timed on this host on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), dense
# split-fp16 kernels: every fp32 product costs three f16 matrix products; the f16 pipe runs 16 x the fp32 rate
# (same table: F16/BF16 ~2.5 PF dense = 16 x 157.3), so the matrix roof of fp32-EQUIVALENT work is 16/3 x 157.3
PEAK_SPLIT16_TFLOPS = round(157.3 * 16 / 3, 1)
PEAK_HBM_GBS = 8000.0           # HBM3E spec


def source_sha():
    """sha256 over the kernel sources: ties a PMC traffic record (profiles/traffic.json) to the build it was taken on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "values_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, fname):
    """HBM-side bytes per launch of `kernel` from the PMC counters: only a record taken on THIS build counts
    (tools/pmc_traffic.py writes profiles/<fname> with the hash of the kernel sources); otherwise None."""
    tpath = os.path.join(ROOT, "profiles", fname)
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            if rec.get("src_sha") == source_sha():
                return rec.get("kernels", {}).get(kernel)
        except Exception:
            pass
    return None


def layer_table(F=8, S=64):
    """label -> (kernel family, Cin, Cout, spatial edge of the OUTPUT/input grid the MACs run on)."""
    t = {}
    enc = [(1, F), (F, F), (F, 2 * F), (2 * F, 2 * F), (2 * F, 4 * F), (4 * F, 4 * F), (4 * F, 8 * F), (8 * F, 8 * F)]
    for i, (ci, co) in enumerate(enc):
        lvl = i // 2
        t[f"contr_{lvl + 1}_{i % 2 + 1}"] = ("conv", ci, co, S >> lvl)
    t["center.0"] = ("conv", 8 * F, 16 * F, S >> 4)
    t["center.2"] = ("conv", 16 * F, 16 * F, S >> 4)
    t["center.4"] = ("convT", 16 * F, 8 * F, S >> 4)
    for lvl in (4, 3, 2, 1):
        c = F << (lvl - 1)
        t[f"expand_{lvl}_1"] = ("conv", 2 * c, c, S >> (lvl - 1))
        # (round 5) the same layer as two launches over the halves of its input: conv_skip(skip) + bias -> partial sums, then
        # conv_up(up) + partial sums -> activation (the second launch also reads the 4 c bytes per voxel of partial sums)
        t[f"expand_{lvl}_1(skip half)"] = ("conv", c, c, S >> (lvl - 1))
        t[f"expand_{lvl}_1(up half)"] = ("conv_acc", c, c, S >> (lvl - 1))
        t[f"expand_{lvl}_2"] = ("conv", c, c, S >> (lvl - 1))
        if lvl > 1:
            t[f"upscale{lvl}"] = ("convT", c, c // 2, S >> (lvl - 1))
    t["final"] = ("conv1x1", F, 2, S)
    return t


def launch_cost(kind, ci, co, edge, N, N_in=None):
    """(algorithmic FLOPs, input bytes, output bytes, weight bytes) of one layer over N samples.  N_in: DISTINCT input
    samples when several output samples read the same input tensor (MC-dropout: the T samples of a volume share
    contr_1_1's output, so contr_1_2 reads it once per volume -- counting it T times overstated that layer's GB/s 1.6x)."""
    vox = edge ** 3
    if N_in is None:
        N_in = N
    if kind == "conv":
        return 2.0 * 27 * ci * co * vox * N, 4.0 * ci * vox * N_in, 4.0 * co * vox * N, 4.0 * 27 * ci * co
    if kind == "conv_acc":     # + the partial sums it adds (vx_conv3d_args.acc_in), counted with the weights: read whatever is fused in front
        return 2.0 * 27 * ci * co * vox * N, 4.0 * ci * vox * N_in, 4.0 * co * vox * N, 4.0 * 27 * ci * co + 4.0 * co * vox * N
    if kind == "convT":
        return 2.0 * ci * co * 8 * vox * N, 4.0 * ci * vox * N, 4.0 * 8 * co * vox * N, 4.0 * 8 * ci * co
    if kind == "conv1x1":
        return 2.0 * ci * co * vox * N, 4.0 * ci * vox * N, 4.0 * co * vox * N, 4.0 * ci * co
    return 0.0, 0.0, 0.0, 0.0


def fused_cost(parts, tab, n_of, n_in_of=None):
    """Algorithmic FLOPs and bytes of ONE launch that computes the layers `parts` in sequence: a tensor handed from one
    fused layer to the next is never moved, so the launch reads the first layer's input plus what each later layer takes
    from elsewhere (expand_1_1 after upscale2: the skip half) and writes the last layer's output."""
    fl = by = 0.0
    prev_out = None
    for k, part in enumerate(parts):
        if part not in tab:
            continue
        kind, ci, co, edge = tab[part]
        f1, bi, bo, bw = launch_cost(kind, ci, co, edge, n_of(part), n_in_of(part) if n_in_of else None)
        fl += f1
        by += bw + (bi if prev_out is None else max(bi - prev_out, 0.0))
        prev_out = bo
    return fl, by + (prev_out or 0.0)


def profiled_forward(model, x, n_samples, seed):
    """Per-launch (label, kernel name, milliseconds) of one eager forward (HIP events on the launch stream, inside the
    library; the kernel name is the template instance the launch dispatched to, as rocprofv3 prints it)."""
    import torch
    from values_amd import _lib
    lib = _lib.load()
    dev = x.device
    V, _, D, H, W = x.shape
    N = V * n_samples
    w, _keep = model._ensure_packed(dev)
    ws, off, ws_bytes = model._workspace(N, D, H, W, dev)
    out = torch.empty((N, w.num_classes, D, H, W), dtype=torch.float32, device=dev)
    run = _lib.UNet3DRun()
    run.x = x.data_ptr()
    run.N, run.D, run.H, run.W, run.repeat = N, D, H, W, n_samples
    # the dropout mode the model's own forward would pick (values_amd/unet3d.py:_run): deterministic ensemble members run none
    hashed = model.training and model.dropout_prob > 0
    run.drop_mode, run.seed = (_lib.VX_DROP_HASH if hashed else _lib.VX_DROP_NONE), seed
    run.logits = out.data_ptr()
    run.workspace = ws.data_ptr() + off
    run.workspace_bytes = ws_bytes
    ms = (C.c_float * 96)()
    labels = (C.c_char_p * 96)()
    n = C.c_int(0)
    _lib.check(lib.vx_unet3d_forward_profiled(C.byref(w), C.byref(run), _lib.stream_ptr(), 96, ms, labels, C.byref(n)),
               "vx_unet3d_forward_profiled")
    rows = []
    for i in range(n.value):
        lab = labels[i].decode()
        name = lab.split("|", 1)[1] if "|" in lab else None
        rows.append((lab.split("|", 1)[0], name, float(ms[i])))
    return rows


def roofline_leg(model, x, T, reps=3, chunks=None):
    """Per-launch HIP-event times of the forwards one step launches: the SAME volume chunks the timed path runs, one
    after the other on this stream -- so the average launch duration is the one a rocprofv3 trace of this command shows
    for the kernel.  The binding roof of the dominant kernel is the larger of flops / matrix peak and bytes / HBM peak."""
    from values_amd.predict import _volume_chunks
    tab = layer_table(S=x.shape[-1])
    V = x.shape[0]
    N = V * T
    if chunks is None:
        chunks, _ = _volume_chunks(V, None, False, T)
    acc = {}
    per_label = {}
    for rep in range(reps + 1):
        for ci_, (v0, v1) in enumerate(chunks):
            rows = profiled_forward(model, x[v0:v1].contiguous(), T, 1000 + 16 * rep + ci_)
            if rep == 0:
                continue  # warm-up
            Vc = v1 - v0
            for label, kname, ms in rows:
                per_label.setdefault(label, []).append(ms)
                # a fused launch carries every layer it computes; MC-dropout: contr_1_1 runs once per volume (its T
                # samples share input and statistics)
                # ... and contr_1_2 reads that once-per-volume tensor: Vc distinct input samples, Vc * T output samples
                fl, by = fused_cost(label.split("+"), tab, lambda part: Vc if part == "contr_1_1" else Vc * T,
                                    lambda part: Vc if part in ("contr_1_1", "contr_1_2") else Vc * T)
                name = kname or label.split(":")[0]
                a = acc.setdefault(name, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                a["ms"] += ms; a["flops"] += fl; a["bytes"] += by; a["launches"] += 1
    per_label = {k: [sum(v) / reps] for k, v in per_label.items()}     # per step: the chunks' launches summed
    total_ms = sum(a["ms"] for a in acc.values()) / reps
    name, a = max(acc.items(), key=lambda kv: kv[1]["ms"])
    sec = a["ms"] * 1e-3
    tflops = a["flops"] / sec / 1e12
    gbs = a["bytes"] / sec / 1e9
    split = any(k in name for k in ("s16", "xp8", "zc16", "deep"))      # the split-fp16 kernel families (convT_k2s2_s16 included)
    mpeak = PEAK_SPLIT16_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    t_mfma = a["flops"] / (mpeak * 1e12)
    t_hbm = a["bytes"] / (PEAK_HBM_GBS * 1e9)
    bound = "mfma" if t_mfma >= t_hbm else "hbm"
    roof = {"bound": bound, "kernel": name,
            "achieved": round(tflops if bound == "mfma" else gbs, 3), "peak": mpeak if bound == "mfma" else PEAK_HBM_GBS,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "frac": round((tflops / mpeak) if bound == "mfma" else (gbs / PEAK_HBM_GBS), 4),
            "frac_mfma": round(tflops / mpeak, 4), "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
            "achieved_tflops": round(tflops, 3), "achieved_GBps": round(gbs, 1),
            "mfma_peak_tflops": mpeak, "hbm_peak_GBps": PEAK_HBM_GBS,
            "frac_of_fp32_matrix_peak": round(tflops / PEAK_FP32_MFMA_TFLOPS, 4),
            "peak_note": ("matrix roof = fp32-equivalent rate of the split-fp16 scheme: f16 dense peak (16 x 157.3 TF) / 3 "
                          "products per fp32 product; HBM roof = 8 TB/s spec; bound = the roof that takes longer for this "
                          "kernel's algorithmic flops and bytes") if split else "fp32 matrix peak, dense; HBM 8 TB/s spec",
            "avg_launch_ms": round(a["ms"] / a["launches"], 4), "launches_per_step": a["launches"] // reps,
            "samples_per_launch": round(N / len(chunks), 1),
            "share_of_forward": round(a["ms"] / reps / total_ms, 3), "traffic": None}
    roof["traffic"] = pmc_traffic(name, "traffic.json")
    detail = {"forward_ms_sum_of_launches": round(total_ms, 3), "samples": N,
              "launches_per_forward": sum(v["launches"] for v in acc.values()) // reps // len(chunks),
              "kernels": {k: {"ms_per_step": round(v["ms"] / reps, 4), "launches": v["launches"] // reps,
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3) if v["flops"] else None,
                              "alg_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                          for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["ms"])},
              "layers_ms": {k: round(sum(v) / len(v), 4) for k, v in per_label.items()}}
    return roof, detail


def cpu_baseline_leg(state_dict, T=10, budget_s=25.0):
    """The oracle (kind 'port': PyTorch-CPU restatement of test_3D.py:417-482) on a bounded sample of the same
    workload: the T float64 autograd-on forwards of one 64^3 volume as the reference runs them (fewer, stated, if they
    would not fit the time budget) + the T-sample calculate_uncertainty restatement -> volumes/s; and the same loop in
    float32 without autograd (the "fair" CPU number of SURVEY 8d) next to it."""
    import numpy as np
    import torch
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import DROPOUT_ORDER, unet3d_forward
    ncpu = os.cpu_count() or 1
    # the weights the GPU leg ran with (torch default init, seed 123), as float64 like the reference's model.double()
    sd0 = {k: v.detach().cpu().double() for k, v in state_dict.items()}
    # pick the thread count that is fastest on a 32^3 pass (oneDNN/ATen float64 conv3d slows down when
    # oversubscribed: 256 threads were 3x slower than 8 on the first GPU-box run)
    cand = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
    xs = torch.randn((1, 1, 32, 32, 32), dtype=torch.float64)
    best, cores = None, cand[0]
    for c in cand:
        torch.set_num_threads(c)
        with torch.no_grad():
            unet3d_forward(sd0, xs)
            t0 = time.perf_counter()
            unet3d_forward(sd0, xs)
            dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, c
    torch.set_num_threads(cores)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    g = torch.Generator().manual_seed(123)
    x = torch.randn((1, 1, 64, 64, 64), generator=g, dtype=torch.float64)
    shapes = [(8, 64), (8, 64), (16, 32), (16, 32), (32, 16), (32, 16), (64, 8), (64, 8), (64, 8), (64, 8), (64, 8),
              (32, 16), (32, 16), (16, 32), (16, 32), (8, 64), (8, 64)]

    def passes(sdx, xx, n, grad):
        t0 = time.perf_counter()
        sms = []
        done = 0
        for _ in range(n):
            masks = {nm: torch.rand((1, c, s, s, s), generator=g) > 0.5 for nm, (c, s) in zip(DROPOUT_ORDER, shapes)}
            with torch.set_grad_enabled(grad):
                logits = unet3d_forward(sdx, xx, masks=masks)
            sms.append(torch.softmax(logits, 1).detach().numpy()[0])
            done += 1
            if time.perf_counter() - t0 > budget_s * 0.6 and done >= 2:
                break
        return (time.perf_counter() - t0) / done, done, sms

    t_pass, n64, sms = passes(sd, x, T, True)
    stack = np.stack((sms * T)[:T])                       # (T,2,64,64,64) float64 buffer like concat_data's
    t0 = time.perf_counter()
    uo.calculate_uncertainty(stack)
    t_red = time.perf_counter() - t0
    vps = 1.0 / (T * t_pass + t_red)
    sd32 = {k: v.detach().float() for k, v in sd.items()}
    t32, n32, sms32 = passes(sd32, x.float(), T, False)
    t0 = time.perf_counter()
    uo.calculate_uncertainty(np.stack((sms32 * T)[:T]))
    t_red32 = time.perf_counter() - t0
    return {"value": round(vps, 5), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"{n64} of the {T} float64 autograd-on MC-dropout forwards of one 64^3 volume ({t_pass:.2f} s/pass"
                      f"{'' if n64 == T else ', extrapolated to ' + str(T)}) + one T={T} entropy/MI reduction ({t_red:.2f} s); "
                      f"torch {torch.__version__} CPU, {cores} of {ncpu} host threads (fastest of {cand} on a 32^3 pass)",
            "fair_float32_nograd": {"value": round(1.0 / (T * t32 + t_red32), 5), "unit": "volumes/s",
                                    "sample": f"{n32} float32 no-grad passes ({t32:.3f} s/pass) + reduction ({t_red32:.2f} s)"}}


def cpu_baseline_leg_passes(state_dicts, passes_per_unit, reduce_T, dropout, unit, what, reduce_vox_scale=1, budget_s=20.0):
    """cpu_baseline of configs C3 / C5 (kind 'port'), bounded like C2's: the oracle's float64 autograd-on forward of ONE 64^3
    volume (patch) timed over a few passes -- one per entry of state_dicts, cycled -- and scaled to the `passes_per_unit`
    forwards a unit (volume / image) takes, + the calculate_uncertainty restatement over reduce_T samples of a 64^3 buffer
    (x reduce_vox_scale for an image of that many 64^3 volumes' worth of voxels).  Returns the cpu_baseline object."""
    import numpy as np
    import torch
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import DROPOUT_ORDER, unet3d_forward
    ncpu = os.cpu_count() or 1
    cores = min(8, ncpu)           # the thread count C2's leg finds fastest on the driver's boxes (oversubscription slows ATen's float64 conv3d)
    torch.set_num_threads(cores)
    sds = [{k: v.detach().cpu().double().requires_grad_(True) for k, v in sd.items()} for sd in state_dicts]
    g = torch.Generator().manual_seed(123)
    x = torch.randn((1, 1, 64, 64, 64), generator=g, dtype=torch.float64)
    shapes = [(8, 64), (8, 64), (16, 32), (16, 32), (32, 16), (32, 16), (64, 8), (64, 8), (64, 8), (64, 8), (64, 8),
              (32, 16), (32, 16), (16, 32), (16, 32), (8, 64), (8, 64)]
    t0 = time.perf_counter()
    sms, done = [], 0
    while done < passes_per_unit:
        masks = None
        if dropout:
            masks = {nm: torch.rand((1, c, s_, s_, s_), generator=g) > 0.5 for nm, (c, s_) in zip(DROPOUT_ORDER, shapes)}
        logits = unet3d_forward(sds[done % len(sds)], x, masks=masks)
        sms.append(torch.softmax(logits, 1).detach().numpy()[0])
        done += 1
        if time.perf_counter() - t0 > budget_s * 0.6 and done >= 2:
            break
    t_pass = (time.perf_counter() - t0) / done
    stack = np.stack((sms * reduce_T)[:reduce_T])
    t0 = time.perf_counter()
    uo.calculate_uncertainty(stack)
    t_red = (time.perf_counter() - t0) * reduce_vox_scale
    return {"value": round(1.0 / (passes_per_unit * t_pass + t_red), 6), "unit": unit, "cores": cores, "kind": "port",
            "sample": f"{done} float64 autograd-on forwards of one 64^3 volume ({t_pass:.2f} s/pass) scaled to the {passes_per_unit} "
                      f"forwards of {what} + the T={reduce_T} entropy/MI reduction ({t_red:.2f} s"
                      f"{'' if reduce_vox_scale == 1 else ', a 64^3 buffer timed, x' + str(reduce_vox_scale) + ' voxels'}); "
                      f"torch {torch.__version__} CPU, {cores} of {ncpu} host threads"}


def cpu_baseline_leg_2d(extra, state_dict, H, W, views=8, budget_s=25.0):
    """C4's CPU baseline (kind 'port'): the oracle's HRNet restatement (oracle/hrnet_oracle.py, training-mode BatchNorm as the
    reference runs it, float32 as test_2D.py does) on a BOUNDED sample -- ONE view of ONE image at the bench's size; an image
    is `views` such forwards + the softmax / entropy reduction (timed on that one view's worth and scaled)."""
    import numpy as np
    import torch
    from oracle import uncertainty_oracle as uo
    from oracle.hrnet_oracle import hrnet_forward
    ncpu = os.cpu_count() or 1
    cores = min(32, ncpu)
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().float() for k, v in state_dict.items()}
    g = torch.Generator().manual_seed(123)
    x = torch.randn((1, 3, H, W), generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        lg = hrnet_forward(extra, sd, x)
        t_view = time.perf_counter() - t0
        if t_view < budget_s / 3:          # a second, warm pass where the budget allows
            t0 = time.perf_counter()
            lg = hrnet_forward(extra, sd, x)
            t_view = time.perf_counter() - t0
    sm = torch.softmax(lg, 1).numpy()                        # (1, C, H, W)
    t0 = time.perf_counter()
    uo.calculate_uncertainty(np.concatenate([sm, sm], 0))    # two views' worth of the reference's Python-loop reduction
    t_red = (time.perf_counter() - t0) * views / 2.0
    return {"value": round(1.0 / (views * t_view + t_red), 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"one of the {views} views of one {W}x{H} image through the float32 PyTorch-CPU restatement of HighResolutionNet "
                      f"({t_view:.2f} s/view) + the reduction over two views scaled to {views} ({t_red:.2f} s); torch {torch.__version__} "
                      f"CPU, {cores} of {ncpu} host threads"}


# ------------------------------------------------------------------------------------------------------------------
def kfd_gpu_count():
    """GPUs of this node as the kernel driver lists them: /sys/class/kfd/kfd/topology/nodes/*/properties, a node with
    simd_count > 0 is a GPU (CPU nodes have 0).  Reads sysfs only: neither HIP nor amdsmi nor torch is touched, so the
    launching parent stays a process that never initialised a GPU runtime.  None when the topology is not readable
    (then the children find out: rank 0 of a short world exits non-zero)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    seen = False
    for d in nodes:
        try:
            txt = open(os.path.join(base, d, "properties")).read()
        except OSError:
            continue
        seen = True
        for ln in txt.splitlines():
            f = ln.split()
            if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                n += 1
    return n if seen else None


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run`.  This parent
    never imports torch and never touches HIP / amdsmi (the GPU count comes from sysfs): it only waits for the child and
    passes its exit code on."""
    import socket
    have = kfd_gpu_count()
    if have is not None and have < args.gpus and not os.environ.get("VX_BENCH_EMULATE_RANKS"):
        print(f"bench.py: --gpus {args.gpus} but the kfd topology of this node lists {have} GPU(s)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def timed_regions(step, flush, barrier, steps, warmup, repeats, reduce_max, min_seconds=0.0, max_regions=64):
    """W untimed steps, then `repeats` regions of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both
    sides; per region the max over ranks.  With `min_seconds`, further regions of the same K steps follow until the timed
    regions add up to that much GPU time (so that a sampler outside this process sees the GPU busy); every rank takes the
    same decision because the region times are the max over ranks.  Returns the list of region times (seconds)."""
    k = 0
    for _ in range(warmup):
        step(k)
        k += 1
    flush()
    barrier()
    times = []
    while len(times) < repeats or (sum(times) < min_seconds and len(times) < max_regions):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(k)
            k += 1
        flush()
        barrier()
        times.append(reduce_max(time.perf_counter() - t0))
    return times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=("C2", "C3", "C4", "C5"))
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = their median")
    ap.add_argument("--min-gpu-seconds", type=float, default=4.0,
                    help="C2: keep adding timed regions of --steps steps until they add up to this much GPU time")
    ap.add_argument("--volumes", type=int, default=None, help="units per GPU per step (C2: 32 volumes, C3: 16, C4: 4 images)")
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-batch64", action="store_true", help="C2: skip the side measurement at 64 volumes per step (value_at_64)")
    ap.add_argument("--hrnet-width", type=int, default=18, choices=(18, 48),
                    help="C4: HRNet-W18 (BASELINE config 4) or W48 (the width of the reference's shipped configs)")
    ap.add_argument("--storage16", action="store_true",
                    help="(default at one GPU since round 6; kept so that old command lines parse) C2: ALSO time the opt-in reduced-"
                         "precision modes (vx_config.storage16 = 1: expand_1_1's tensor stored as fp16; = 2: plus one fp16 product per fp32 "
                         "product on the full-resolution launches) and print them as a side object with their measured deviation from "
                         "the default path; `value` stays the default path")
    ap.add_argument("--no-storage16", action="store_true", help="C2: skip that side measurement")
    ap.add_argument("--graph", action="store_true", help="C2: replay the step as one captured hipGraph (GraphedPredictor)")
    ap.add_argument("--no-gather", action="store_true", help="C2, N > 1: leave the maps on the ranks that computed them (default: "
                    "every step's maps are gathered on rank 0 inside the timed region, the metric SURVEY 8d defines)")
    ap.add_argument("--gather", action="store_true", help="(default since round 3; kept so that old command lines still parse)")
    ap.add_argument("--eager", action="store_true", help="C4: eager launches instead of the captured hipGraph")
    ap.add_argument("--roofline-only", action="store_true",
                    help="C4: run ONLY the roofline leg (single-stream eager forwards with a HIP event pair around every conv, 8 "
                         "reps) and print its object: the command a rocprofv3 --kernel-trace --stats / --pmc pass wraps so that "
                         "the CSV's per-kernel averages are those of the launches the `roofline` object times (the graph replay "
                         "overlaps branch kernels on side streams, where rocprofv3's per-kernel durations are inflated)")
    ap.add_argument("--roofline-reps", type=int, default=8, help="C4 --roofline-only: timed forwards (a second profile with another count "
                    "shows which kernels of the trace belong to the model's one-off setup: their call counts do not move)")
    ap.add_argument("--pcie", action="store_true", help="also time the host-inclusive variant (pinned host input, maps copied back)")
    ap.add_argument("--detail", type=str, default=None, help="write the per-kernel breakdown JSON here")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
              "(or without a launcher: bench.py starts its own ranks)", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: values_amd has no CPU path")
    # VX_BENCH_EMULATE_RANKS=1 (tests/test_gpu_dist.py only): the N ranks share the visible GPU(s) and talk over gloo -- a functional run
    # of the whole N > 1 path (rank spawn, shards, the gather pipeline inside the timed region, max over ranks, the JSON line) on a
    # 1-GPU box.  Not a measurement: the line carries "emulated": true and its value is meaningless.
    emulate = bool(os.environ.get("VX_BENCH_EMULATE_RANKS")) and world > 1
    dev_index = local_rank % torch.cuda.device_count() if emulate else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emulate:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: {dist.get_world_size()} ranks joined, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(3)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(dt):
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    if args.config == "C2":
        line = run_c2(args, world, rank, dev, barrier, reduce_max)
    elif args.config == "C3":
        line = run_c3(args, world, rank, dev, barrier, reduce_max)
    elif args.config == "C5":
        line = run_c5(args, world, rank, dev, barrier, reduce_max)
    else:
        line = run_c4(args, world, rank, dev, barrier, reduce_max)
    if world > 1:
        dist.barrier()
    if rank == 0:
        line["ranks_joined"] = dist.get_world_size() if world > 1 else 1
        if emulate:
            line["emulated"] = True      # ranks sharing a GPU over gloo: a functional check of the N > 1 path, not a measurement
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def summarise(times, units_per_region, steps):
    med = statistics.median(times)
    return {"value": round(units_per_region / med, 3), "ms_per_step": round(med / steps * 1e3, 3),
            "repeats": {"regions": len(times), "steps_per_region": steps,
                        "value_median": round(units_per_region / med, 3),
                        "value_min": round(units_per_region / max(times), 3),
                        "value_max": round(units_per_region / min(times), 3),
                        "value_first_region": round(units_per_region / times[0], 3)}}


def run_c2(args, world, rank, dev, barrier, reduce_max):
    import torch
    from values_amd import UNet3D, predict_uncertainty
    from values_amd.dist import MapGatherPipeline

    torch.manual_seed(123)  # reference seed (configs/dropout_config.yaml:8); default torch init = random weights
    model = UNet3D(num_classes=2, do_dropout=True).to(dev)
    V, T, S = args.volumes or 32, args.T, args.size
    g = torch.Generator(device="cpu").manual_seed(123 + rank)
    x = torch.randn((V, 1, S, S, S), generator=g).to(dev)  # z-scored synthetic volumes, resident in HBM

    # the volumes are independent: no collective on the compute path.  The metric counts a volume once its
    # maps are resident on rank 0, so at N > 1 every step's maps are gathered there, the gather of step i overlapping the
    # kernels of the following steps, everything flushed before the closing barrier so that the timed region contains
    # every transfer.  --no-gather: the maps stay on their ranks (timed as a side number by default).
    gather = world > 1 and not args.no_gather
    pipe = MapGatherPipeline(world if gather else 1, rank, depth=2)

    # --graph: the step (33 forward launches + the reduction) captured once and replayed with a fresh device seed word
    gp = None
    if args.graph:
        from values_amd import GraphedPredictor
        gp = GraphedPredictor([model], tuple(x.shape), n_pred=T)

    def step(i):
        out = gp(x, seed=i) if gp is not None else predict_uncertainty([model], x, n_pred=T, seeds=[i], range_check="off")
        return pipe.submit(out)

    times = timed_regions(step, pipe.flush, barrier, args.steps, args.warmup, max(1, args.repeats), reduce_max,
                          min_seconds=args.min_gpu_seconds)
    side = None
    if gather:
        # side number: the same steps with the maps left on their ranks (what the gather costs is value vs this)
        local = MapGatherPipeline(1, rank, depth=2)

        def step_local(i):
            return local.submit(predict_uncertainty([model], x, n_pred=T, seeds=[i], range_check="off"))
        side = timed_regions(step_local, local.flush, barrier, args.steps, 1, min(3, max(1, args.repeats)), reduce_max)
    model.check_range()      # the fp16-range word keeps the running maximum over every step above

    st16 = None
    if (args.storage16 or world == 1) and not args.no_storage16 and not args.graph:
        try:
            # opt-in reduced-storage throughput mode: never `value` (it cannot meet the 1e-4 parity of the maps); what it buys and
            # what it costs, measured on the same volumes and dropout seeds as the default path
            from values_amd import _lib
            ref_out = predict_uncertainty([model], x, n_pred=T, seeds=[777], range_check="off")
            ref_maps = {k: ref_out[k].clone() for k in ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty", "mean_softmax")}
            ref_seg = ref_out["pred_seg_mean"].clone()
            with _lib.config(storage16=1):
                red = predict_uncertainty([model], x, n_pred=T, seeds=[777], range_check="off")
                dev_maps = {k: round(float((red[k] - ref_maps[k]).abs().max().item()), 7) for k in ref_maps}
                flips = int((red["pred_seg_mean"] != ref_seg).sum().item())
                # an EAGER step: a graph captured above replays the default path's kernels whatever the configuration says now
                def step16(i):
                    return pipe.submit(predict_uncertainty([model], x, n_pred=T, seeds=[i], range_check="off"))
                t16 = timed_regions(step16, pipe.flush, barrier, args.steps, 2, min(3, max(1, args.repeats)), reduce_max)
            s16 = summarise(t16, V * world * args.steps, args.steps)
            st16 = {"value": s16["value"], "ms_per_step": s16["ms_per_step"], "unit": "volumes/s",
                    "max_abs_diff_vs_default_path": dev_maps, "argmax_flips": flips, "voxels": int(ref_seg.numel()),
                    "note": "vx_config.storage16 = 1: expand_1_1 -> expand_1_2 tensor stored as fp16 (2 instead of 3 matrix products "
                            "in expand_1_2); NOT the default, NOT within the 1e-4 parity bar -- the deviation from the float64 "
                            "oracle is asserted in tests/test_gpu_unet3d.py::test_storage16_mode_reports_its_deviation..."}
            # mode 2 (round 6): besides that tensor, ONE fp16 product per fp32 product on the three full-resolution launches
            with _lib.config(storage16=2):
                red = predict_uncertainty([model], x, n_pred=T, seeds=[777], range_check="off")
                dev_maps2 = {k: round(float((red[k] - ref_maps[k]).abs().max().item()), 7) for k in ref_maps}
                flips2 = int((red["pred_seg_mean"] != ref_seg).sum().item())

                def step16b(i):
                    return pipe.submit(predict_uncertainty([model], x, n_pred=T, seeds=[i], range_check="off"))
                t16b = timed_regions(step16b, pipe.flush, barrier, args.steps, 2, min(3, max(1, args.repeats)), reduce_max)
            s16b = summarise(t16b, V * world * args.steps, args.steps)
            st16["fp16_products"] = {"value": s16b["value"], "ms_per_step": s16b["ms_per_step"], "unit": "volumes/s",
                                     "max_abs_diff_vs_default_path": dev_maps2, "argmax_flips": flips2,
                                     "note": "vx_config.storage16 = 2: mode 1 plus one fp16 product per fp32 product (activations and "
                                             "weights rounded to fp16, fp32 accumulation) on contr_1_2, upscale2 + expand_1_1 and "
                                             "expand_1_2 + head -- what BASELINE config 2 calls bf16; a side number, never `value`; "
                                             "deviation from the float64 oracle asserted in test_fp16_products_mode_reports_its_deviation_at_64"}
        except Exception as e:      # a side measurement must never take the line with it
            st16 = {"error": f"{type(e).__name__}: {e}"}

    pcie = None
    if args.pcie and world == 1:
        # host-inclusive variant: volumes start in pinned host memory, the maps end there; never reported as `value`.
        # values_amd.HostPipeline: upload and download on their own streams, step i's maps travel while step i + 1 computes
        from values_amd import HostPipeline
        xh = x.cpu().pin_memory()
        hp = HostPipeline([model], n_pred=T)
        for i in range(3):
            hp.submit(xh, seeds=[i])
        hp.flush()
        torch.cuda.synchronize()
        hsteps = max(3 * args.steps, 30)     # the pipeline is 3 steps deep: enough steps to amortise fill and drain
        t1 = time.perf_counter()
        for i in range(hsteps):
            hp.submit(xh, seeds=[100 + i])
        hp.flush()
        torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        pcie = {"volumes_per_s": round(V * hsteps / dth, 3), "steps": hsteps,
                "note": "pinned host -> device input, maps device -> pinned host, copies on their own streams "
                        "(values_amd.HostPipeline), fill and drain of the 3-step pipeline included"}

    # side object (never `value`; the bench keeps 32 volumes per step for comparability with rounds 1-4): the same step at 64
    # volumes per GPU -- the 8^3 / 4^3 layers' launches amortise over twice the samples.  Workspace 47 MB per sample:
    # 640 samples = 30 GB + 1.3 GB of logits, of the 288 GB.
    at64 = None
    if world == 1 and V == 32 and not args.no_batch64 and gp is None:
        g64 = torch.Generator(device="cpu").manual_seed(777)
        x64 = torch.randn((64, 1, S, S, S), generator=g64).to(dev)
        loc64 = MapGatherPipeline(1, rank, depth=2)

        def step64(i):
            return loc64.submit(predict_uncertainty([model], x64, n_pred=T, seeds=[i], range_check="off"))
        t64 = timed_regions(step64, loc64.flush, barrier, args.steps, 2, min(3, max(1, args.repeats)), reduce_max)
        s64 = summarise(t64, 64 * args.steps, args.steps)
        at64 = {"value": s64["value"], "ms_per_step": s64["ms_per_step"], "unit": "volumes/s", "volumes_per_gpu_per_step": 64,
                "note": "the same step at 64 volumes (640 samples) per GPU; needs ~31 GB of HBM for workspace + logits"}
        del x64
        model.check_range()

    roof, detail, cpu, lat = None, None, None, None
    if rank == 0 and not args.no_roofline:
        roof, detail = roofline_leg(model, x, T)
        if args.detail:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
            json.dump(detail, open(args.detail, "w"), indent=1)
    if rank == 0 and not args.no_latency:
        lat = latency_leg(model, x[:1], T)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(model.state_dict(), T)
    line = {"metric": "uncertainty-volumes/sec (64^3, T=10 MC-dropout)", "unit": "volumes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {S}^3 1-channel volumes, UNet3D(initial_filter_size=8, 2 classes), T={T} "
                                   "MC-dropout passes + fused softmax/entropy/MI/variance/argmax reduction",
                       "volumes_per_gpu_per_step": V, "samples_per_gpu_per_step": V * T,
                       "sharding": (f"volumes over {world} rank(s); " + ("maps gathered to rank 0 inside the timed region (RCCL, "
                                    "overlapped with the next steps' kernels)" if gather
                                    else "maps stay on the rank that computed them, no data-path collective"))
                       if world > 1 else "single GPU",
                       "dropout": "hash bit generator, new seed every step", "weights": "torch default init, seed 123"},
            "roofline": roof, "cpu_baseline": cpu}
    line.update(summarise(times, V * world * args.steps, args.steps))
    line["gpu_seconds_timed"] = round(sum(times), 3)
    if side is not None:
        ss = summarise(side, V * world * args.steps, args.steps)
        line["no_gather"] = {"value": ss["value"], "ms_per_step": ss["ms_per_step"],
                             "note": "same steps, maps left on the ranks that computed them"}
    if at64 is not None:
        line["value_at_64"] = at64
    if st16 is not None:
        line["storage16"] = st16
    if lat is not None:
        line["latency_single_volume"] = lat
    if detail is not None:
        line["launches_per_forward"] = detail["launches_per_forward"]
    if pcie is not None:
        line["pcie_inclusive"] = pcie
    return line


def latency_leg(model, x1, T, reps=30):
    """One 64^3 volume, T passes + reduction, alone on the GPU: eager launches vs one captured hipGraph replay."""
    import torch
    from values_amd import predict_uncertainty
    try:
        from values_amd.predict import GraphedPredictor
    except ImportError:
        GraphedPredictor = None

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    out = {"volumes": 1, "T": T,
           "eager_ms": round(timeit(lambda: predict_uncertainty([model], x1, n_pred=T, seeds=[1], range_check="off")), 3)}
    if GraphedPredictor is not None:
        gp = GraphedPredictor([model], x1.shape, n_pred=T)
        k = [0]

        def run():
            k[0] += 1
            gp(x1, seed=k[0])
        out["graph_ms"] = round(timeit(run), 3)
    return out


def run_c3(args, world, rank, dev, barrier, reduce_max):
    """BASELINE config 3: 64^3, 5-member deep ensemble, members sharded over the ranks (with volume blocks so that every
    rank has work), one sum-reduce of sufficient statistics to rank 0 per step."""
    import torch
    from values_amd import UNet3D
    from values_amd.dist import ensemble_uncertainty_sharded
    M, S = 5, args.size
    Vg = args.volumes or 64                       # 64 volumes x 5 members = 320 forwards per step and GPU, as config C2 runs (16: 3 545 volumes/s, 32: 4 371, 64: 5 067 on one box)
    V = Vg * world                                    # weak scaling: volumes per step grow with the ranks
    members = []
    for m in range(M):
        torch.manual_seed(123 + m)                    # seeds 123..127
        members.append(UNet3D(num_classes=2, do_dropout=False).to(dev))
    g = torch.Generator(device="cpu").manual_seed(123)
    x = torch.randn((V, 1, S, S, S), generator=g).to(dev)      # the same volumes on every rank

    def step(i):
        return ensemble_uncertainty_sharded(members, x, world, rank, n_pred=1)

    times = timed_regions(step, lambda: None, barrier, args.steps, args.warmup, max(1, args.repeats), reduce_max)
    roof = None
    if rank == 0 and not args.no_roofline:
        # one member's forward over this rank's volumes (n_pred = 1, no dropout): the launches a step issues M times
        from values_amd.dist import ensemble_work_items
        vb = max(1, max((v1 - v0) for _, v0, v1 in ensemble_work_items(M, V, world)[rank]))
        roof, _ = roofline_leg(members[0], x[:vb], 1, chunks=[(0, vb)])
        roof["note"] = f"one member's forward over a {vb}-volume block; a step launches it once per (member, block) item"
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg_passes([m.state_dict() for m in members], M, M, False, "volumes/s",
                                      f"one volume ({M} members, no dropout)")
    line = {"metric": "uncertainty-volumes/sec (64^3, 5-member deep ensemble)", "unit": "volumes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: {S}^3 volumes, {M}-member deep ensemble (n_pred=1, no dropout), (member, volume "
                                   "block) items dealt over the ranks, RCCL sum-reduce of sufficient statistics + finalize",
                       "volumes_per_step": V, "forwards_per_step": V * M,
                       "sharding": f"{M} members x volume blocks over {world} rank(s)"},
            "roofline": roof, "cpu_baseline": cpu}
    line.update(summarise(times, V * args.steps, args.steps))
    return line


def run_c5(args, world, rank, dev, barrier, reduce_max):
    """BASELINE config 5: 128^3 images, sliding 64^3 patches at half-patch overlap (27 patches per image), T = 20 MC-dropout
    passes per patch, softmax sums + count map accumulated on the device, maps of the whole image from one reduction;
    images sharded over the ranks (one step = `--volumes` images per GPU)."""
    import torch
    from values_amd import UNet3D
    from values_amd.sliding import predict_image_sliding
    torch.manual_seed(123)
    model = UNet3D(num_classes=2, do_dropout=True).to(dev)
    B, T, S, P = args.volumes or 2, 20, 128, 64
    g = torch.Generator(device="cpu").manual_seed(123 + rank)
    imgs = [torch.randn((S, S, S), generator=g).to(dev) for _ in range(B)]

    def step(i):
        out = None
        for b, im in enumerate(imgs):
            out = predict_image_sliding([model], im, patch_size=P, patch_overlap=0.5, n_pred=T, patch_batch=16,
                                        seeds=[1000 * i + b])
        return out

    times = timed_regions(step, lambda: None, barrier, args.steps, args.warmup, max(1, args.repeats), reduce_max)
    roof = None
    if rank == 0 and not args.no_roofline:
        # the forwards of one patch batch (16 patches x T = 320 samples, the launch geometry predict_image_sliding uses)
        xp = torch.stack([im[:P, :P, :P] for im in imgs[:1]] * 16)[:, None].contiguous()
        roof, _ = roofline_leg(model, xp, T, chunks=[(0, 16)])
        roof["note"] = "one patch batch (16 patches of 64^3 x T = 20 samples per launch), as predict_image_sliding issues them"
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg_passes([model.state_dict()], 27 * T, T, True, "images/s",
                                      f"one 128^3 image (27 patches x T={T})", reduce_vox_scale=8)
    line = {"metric": "uncertainty-images/sec (128^3 sliding window, T=20 MC-dropout)", "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C5: {S}^3 images, {P}^3 patches at overlap 0.5 (27 per image), T={T} MC-dropout passes per "
                                   "patch, on-device softmax accumulation + count map, one uncertainty reduction per image",
                       "images_per_gpu_per_step": B, "patch_forwards_per_step": B * 27 * T,
                       "sharding": f"images over {world} rank(s)" if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu}
    line.update(summarise(times, B * world * args.steps, args.steps))
    line["patch_volumes_per_s"] = round(line["value"] * 27, 1)       # comparable with C2's 64^3 volumes/s at T = 20
    return line


def run_c4(args, world, rank, dev, barrier, reduce_max):
    """BASELINE config 4: HRNet-W18 at 1024x512, 8 TTA views per image ({id, H, V, HV} x {clean, noisy}), images
    sharded over the ranks.  One step = B images per GPU = 8 B forwards (every view is its own batch, as test_2D.py:299-311
    runs them: training-mode BatchNorm sees one view at a time)."""
    import torch
    from values_amd.hrnet_configs import hrnet_w18_extra, hrnet_w48_extra
    from values_amd.hrnet import HighResolutionNet
    from values_amd.predict2d import GraphedPredictor2D, predict_logits_2d, process_output_2d, tta_views_8
    # images per step and GPU: 8 (64 views per forward) for W18 -- 142 / 149 / 151 images/s at 4 / 8 / 12 on one box; W48 holds ~25 GB of
    # intermediates per image of a forward: 4
    B = args.volumes or (8 if args.hrnet_width == 18 else 4)
    H, W, NC = 512, 1024, 19
    extra = hrnet_w48_extra(False) if args.hrnet_width == 48 else hrnet_w18_extra(False)
    cfg = {"MODEL": {"EXTRA": extra, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3},
           "DATASET": {"NUM_CLASSES": NC}}
    torch.manual_seed(123)
    model = HighResolutionNet(cfg).to(dev)
    g = torch.Generator(device="cpu").manual_seed(123 + rank)
    x = torch.randn((B, 3, H, W), generator=g).to(dev)
    noisy = (x + 0.05 * torch.randn((B, 3, H, W), generator=g).to(dev))
    # the 8 views of a step are built ON the device by ONE launch inside the step (vx_tta_views_2d: flips as index arithmetic,
    # channels-last at the stem's pitch) from the resident clean / noisy images -- round 3 prepared them outside the timed
    # region with torch.flip and re-laid them out (permute + zero fill) inside it
    from values_amd.data import tta_views_8_device
    from values_amd.predict2d import NhwcViews
    d8, hf, vf = tta_views_8_device(x, noisy)
    views = NhwcViews(d8, hf, vf)

    if args.roofline_only:
        roof = model.profile_forward(views.t.view(-1, H, W, 4), peak_tflops=PEAK_SPLIT16_TFLOPS, hbm_gbs=PEAK_HBM_GBS, groups=8,
                                     nhwc=True, reps=max(1, args.roofline_reps))
        # the PMC record is a mean over ALL launches of the instance (several layer shapes): it cannot be the traffic of the one
        # shape the object names -- quoted beside it, `traffic` stays null
        roof["instance_traffic_mean"] = pmc_traffic(roof["kernel"], f"traffic_c4w{args.hrnet_width}.json")
        return {"config": "C4 roofline leg only", "hrnet_width": args.hrnet_width, "images": B, "roofline": roof}

    def step_eager(i):
        # every view's softmax is taken in its forward's upsampling pass (the outputs -- softmax_pred and the maps -- are the
        # reference's; the full-resolution logits, which process_output never sees, are not written)
        tta_views_8_device(x, noisy, out=views.t)
        pr = predict_logits_2d([model], views, tta=True, softmax=True)
        return process_output_2d(None, probs=pr)

    # the product path for a fixed image geometry: the step captured once as a hipGraph (the eager walk is ~950 launches
    # from Python and host-bound), replayed per step with the views copied into the graph's inputs
    eager = None
    if not args.eager and rank == 0:     # the eager walk first (side number), its cached blocks returned before the capture
        et = timed_regions(step_eager, lambda: None, torch.cuda.synchronize, max(2, args.steps // 2), 1, 1, lambda v: v)
        eager = summarise(et, B * max(2, args.steps // 2), max(2, args.steps // 2))
        model._hold_last = None
        torch.cuda.empty_cache()
    gp = None if args.eager else GraphedPredictor2D([model], views, tta=True, keep_logits=False)

    def step(i):
        if gp is None:
            return step_eager(i)
        tta_views_8_device(x, noisy, out=gp.x[0])      # this step's views straight into the graph's input
        return gp()

    times = timed_regions(step, lambda: None, barrier, args.steps, args.warmup, max(1, args.repeats), reduce_max)
    roof = None
    if rank == 0 and not args.no_roofline and hasattr(model, "profile_forward"):
        roof = model.profile_forward(views.t.view(-1, H, W, 4), peak_tflops=PEAK_SPLIT16_TFLOPS, hbm_gbs=PEAK_HBM_GBS, groups=8,
                                     nhwc=True)
        # HBM-side bytes per launch (PMC passes around `--config C4 --roofline-only`: the same single-stream launches)
        # the PMC record is a mean over ALL launches of the instance (several layer shapes): it cannot be the traffic of the one
        # shape the object names -- quoted beside it, `traffic` stays null
        roof["instance_traffic_mean"] = pmc_traffic(roof["kernel"], f"traffic_c4w{args.hrnet_width}.json")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg_2d(extra, model.state_dict(), H, W)
    line = {"metric": f"uncertainty-images/sec (HRNet-W{args.hrnet_width}, 1024x512, 8-view TTA)", "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: HRNet-W{args.hrnet_width} (training-mode BatchNorm as the reference runs it), {W}x{H}, {NC} classes, "
                                   "8 TTA views per image + softmax / entropy / MI reduction",
                       "images_per_gpu_per_step": B, "views_per_image": 8,
                       "batching": "the 8 views of a step travel as one batch of 8 B images with BatchNorm statistics per view",
                       "launch": "one hipGraph replay per step (GraphedPredictor2D)" if gp is not None else "eager launches",
                       "sharding": f"images over {world} rank(s)" if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu}
    line.update(summarise(times, B * world * args.steps, args.steps))
    if eager is not None:
        line["eager"] = {"value": eager["value"], "ms_per_step": eager["ms_per_step"]}
    return line


if __name__ == "__main__":
    main()
