#!/usr/bin/env python3
"""Headline benchmark: uncertainty-volumes/sec at 64^3, T=10 MC-dropout (BASELINE.json config C2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one batch of V synthetic 64^3 1-channel volumes per GPU through the whole hot path:
T=10 MC-dropout forwards of UNet3D (batched on device as V*T samples, hash dropout, fp32 exact-MFMA
convs) -> logits in their pred_idx slots -> fused softmax/entropy/MI/argmax reduction -> (N>1) RCCL gather
of the per-volume maps to rank 0.  Inputs are resident in HBM before the timed region.  Volumes are
sharded over ranks (weak scaling: V per GPU fixed), no collective on the data path except that gather.

Rank 0 prints ONE JSON line; it carries `roofline` (dominant kernel, live HIP-event timing, algorithmic
FLOPs) and, at N=1, `cpu_baseline` (the oracle = PyTorch-CPU restatement of test_3D.py's float64 loop, timed
on this host on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), dense
# split-fp16 kernels: every fp32 product costs three f16 matrix products; the f16 pipe runs 16 x the fp32 rate
# (same table: F16/BF16 ~2.5 PF dense = 16 x 157.3), so the matrix roof of fp32-EQUIVALENT work is 16/3 x 157.3
PEAK_SPLIT16_TFLOPS = round(157.3 * 16 / 3, 1)
PEAK_HBM_GBS = 8000.0           # HBM3E spec


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
        t[f"expand_{lvl}_2"] = ("conv", c, c, S >> (lvl - 1))
        if lvl > 1:
            t[f"upscale{lvl}"] = ("convT", c, c // 2, S >> (lvl - 1))
    t["final"] = ("conv1x1", F, 2, S)
    return t


def kernel_name(kind, ci, co, edge, label=""):
    if kind == "conv":
        if ci == 1:
            return f"conv3d_k3_c1_kernel<{co}>"
        fp32 = int(os.environ.get("VX_CONV_FP32", "0") or 0)
        if fp32 == 0 or (fp32 == 2 and co != 8):
            # conv3d_s16.hip (default): <CB, NT, TX, TY, TZ, NW, XP, DB, EPI>, split-fp16 products on v_mfma_f32_16x16x32_f16;
            # Cout = 8 layers: x-pair packing (XP = 1) in chunks of 8 channels, a column is a voxel pair
            xp = 1 if co == 8 else 0
            cb = 8 if xp else (16 if ci % 16 == 0 else 8)
            nt = 2 if co % 32 == 0 else 1
            ex = edge // 2 if xp else edge
            if ex >= 16:   # vx_conv3d_s16_tile: large layers (H >= 32) take 16 x 8 x 4 tiles
                tile, nw = ("16,8,4" if (edge >= 32 and nt == 1) else "16,4,4"), 8
            else:
                tile, nw = ("8,8,4", 8) if ex >= 8 else ("4,4,4", 4)
            # DB = 2: x-pair layers on the large tile run the double-buffered, staggered variant
            db = 0
            if xp and tile == "16,8,4" and not os.environ.get("VX_S16_NO_DB"):
                db = 2 if ci == cb else (3 if ci == 2 * cb else 0)   # 3: the same for two chunks per tile (16 -> 8)
            # EPI: compile-time epilogue of the large-tile instances -- 0 plain (encoder: an InstanceNorm follows),
            # 1 LeakyReLU + hash dropout (decoder), 2 = 1 + fused 1x1x1 head (expand_1_2), 3 run-time (all others)
            epi = 3
            if tile == "16,8,4" and nt == 1 and not os.environ.get("VX_S16_NO_EPI"):
                epi = 0 if label.startswith("contr") else 1
                if label == "expand_1_2" and not os.environ.get("VX_NO_HEAD_FUSION"):
                    epi = 2
            return f"conv3d_k3_s16_kernel<{cb},{nt},{tile},{nw},{xp},{db},{epi}>"
        if co == 8 and ci in (8, 16):
            # conv3d_c8.hip: <chunks of 8 input channels, tile x, y, z> (v_mfma_f32_4x4x1 kernel for Cout = 8)
            tile = "32,4,4" if edge >= 32 else ("16,8,4" if edge >= 16 else "8,4,4")
            return f"conv3d_k3_c8_kernel<{ci // 8},{tile}>"
        # template instance = <CB, NT, TX, TY, TZ, NW, XP> as conv_config()/tile_config() choose it (conv3d_mfma.hip)
        nt = 2 if co % 32 == 0 else 1
        xp = 1 if co == 8 else 0
        cb = 8 if xp else (16 if ci % 16 == 0 else 8)
        ex = edge // 2 if xp else edge  # x-pair: a column is a voxel pair
        tile, nw = ("16,4,4", 8) if ex >= 16 else (("8,8,4", 8) if ex >= 8 else ("4,4,4", 4))
        return f"conv3d_k3_mfma_kernel<{cb},{nt},{tile},{nw},{xp}>"
    if kind == "convT":
        return "convT_k2s2_mfma_kernel" if ci in (16, 32, 64, 128) else "convT_k2s2_kernel"
    if kind == "conv1x1":
        return f"conv1x1_ncdhw_kernel<{ci}>"
    return kind


def launch_cost(kind, ci, co, edge, N):
    """(algorithmic FLOPs, algorithmic bytes) of one launch over N samples."""
    vox = edge ** 3
    if kind == "conv":
        return 2.0 * 27 * ci * co * vox * N, 4.0 * ((ci + co) * vox * N + 27 * ci * co)
    if kind == "convT":
        return 2.0 * ci * co * 8 * vox * N, 4.0 * ((ci + 8 * co) * vox * N + 8 * ci * co)
    if kind == "conv1x1":
        return 2.0 * ci * co * vox * N, 4.0 * (ci + co) * vox * N
    return 0.0, 0.0


def profiled_forward(model, x, n_samples, seed):
    """Per-launch milliseconds of one eager forward (HIP events on the launch stream, inside the library)."""
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
    run.drop_mode, run.seed = _lib.VX_DROP_HASH, seed
    run.logits = out.data_ptr()
    run.workspace = ws.data_ptr() + off
    run.workspace_bytes = ws_bytes
    ms = (C.c_float * 96)()
    labels = (C.c_char_p * 96)()
    n = C.c_int(0)
    _lib.check(lib.vx_unet3d_forward_profiled(C.byref(w), C.byref(run), _lib.stream_ptr(), 96, ms, labels, C.byref(n)),
               "vx_unet3d_forward_profiled")
    return [(labels[i].decode(), float(ms[i])) for i in range(n.value)]


def roofline_leg(model, x, T, reps=3):
    """Per-launch HIP-event times of the forwards one step launches: the SAME volume chunks the timed path runs
    (values_amd.predict splits a batch into chunks on two streams), one after the other on this stream -- so the
    average launch duration is the one a rocprofv3 trace of this command shows for the kernel."""
    from values_amd.predict import _volume_chunks
    tab = layer_table()
    V = x.shape[0]
    N = V * T
    chunks, _ = _volume_chunks(V, None, False, T)
    acc = {}
    per_label = {}
    for rep in range(reps + 1):
        for ci_, (v0, v1) in enumerate(chunks):
            rows = profiled_forward(model, x[v0:v1].contiguous(), T, 1000 + 16 * rep + ci_)
            if rep == 0:
                continue  # warm-up
            Vc = v1 - v0
            for label, ms in rows:
                per_label.setdefault(label, []).append(ms)
                if label in tab:
                    kind, ci, co, edge = tab[label]
                    name = kernel_name(kind, ci, co, edge, label)
                    # MC-dropout: contr_1_1 runs once per volume (its T samples share input and statistics)
                    fl, by = launch_cost(kind, ci, co, edge, Vc if label == "contr_1_1" else Vc * T)
                else:
                    name, fl, by = label.split(":")[0], 0.0, 0.0
                a = acc.setdefault(name, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                a["ms"] += ms; a["flops"] += fl; a["bytes"] += by; a["launches"] += 1
    per_label = {k: [sum(v) / reps] for k, v in per_label.items()}     # per step: the chunks' launches summed
    total_ms = sum(a["ms"] for a in acc.values()) / reps
    dom = max(acc.items(), key=lambda kv: kv[1]["ms"])
    name, a = dom
    tflops = a["flops"] / (a["ms"] * 1e-3) / 1e12
    split = name.startswith("conv3d_k3_s16")
    peak = PEAK_SPLIT16_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    roof = {"bound": "mfma", "kernel": name, "achieved": round(tflops, 3), "peak": peak,
            "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
            "frac_of_fp32_matrix_peak": round(tflops / PEAK_FP32_MFMA_TFLOPS, 4),
            "peak_note": ("fp32-equivalent matrix roof of the split-fp16 scheme: f16 dense peak (16 x 157.3 TF) / 3 "
                          "products per fp32 product; the native-fp32 matrix peak is 157.3 TF") if split else
                         "fp32 matrix peak, dense",
            "avg_launch_ms": round(a["ms"] / a["launches"], 4), "launches_per_step": a["launches"] // reps,
            "samples_per_launch": round(N / len(chunks), 1),
            "share_of_forward": round(a["ms"] / reps / total_ms, 3), "traffic": None}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            roof["traffic"] = json.load(open(tpath)).get(name)
        except Exception:
            pass
    detail = {"forward_ms_sum_of_launches": round(total_ms, 3), "samples": N,
              "kernels": {k: {"ms_per_step": round(v["ms"] / reps, 4), "launches": v["launches"] // reps,
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3) if v["flops"] else None,
                              "alg_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                          for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["ms"])},
              "layers_ms": {k: round(sum(v) / len(v), 4) for k, v in per_label.items()}}
    return roof, detail


def cpu_baseline_leg(T=10, passes=2):
    """The oracle (kind 'port': PyTorch-CPU restatement of test_3D.py:417-482, float64, autograd on, as the
    reference runs it) on a bounded sample: `passes` of the T forwards of one 64^3 volume + the full T-sample
    calculate_uncertainty restatement; extrapolated to volumes/s."""
    import numpy as np
    import torch
    from oracle import uncertainty_oracle as uo
    from oracle.unet3d_oracle import DROPOUT_ORDER, unet3d_forward
    from values_amd.formula import formula_unet3d_state_dict
    ncpu = os.cpu_count() or 1
    sd0 = {k: torch.from_numpy(v) for k, v in formula_unet3d_state_dict().items()}
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
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in formula_unet3d_state_dict().items()}
    g = torch.Generator().manual_seed(123)
    x = torch.randn((1, 1, 64, 64, 64), generator=g, dtype=torch.float64)
    shapes = [(8, 64), (8, 64), (16, 32), (16, 32), (32, 16), (32, 16), (64, 8), (64, 8), (64, 8), (64, 8), (64, 8),
              (32, 16), (32, 16), (16, 32), (16, 32), (8, 64), (8, 64)]
    t0 = time.perf_counter()
    sm = None
    for _ in range(passes):
        masks = {n: torch.rand((1, c, s, s, s), generator=g) > 0.5 for n, (c, s) in zip(DROPOUT_ORDER, shapes)}
        logits = unet3d_forward(sd, x, masks=masks)
        sm = torch.softmax(logits, 1).detach().numpy()
    t_pass = (time.perf_counter() - t0) / passes
    stack = np.repeat(sm, T, axis=0)  # (T,2,64,64,64) float64 buffer like concat_data's
    t0 = time.perf_counter()
    uo.calculate_uncertainty(stack)
    t_red = time.perf_counter() - t0
    vps = 1.0 / (T * t_pass + t_red)
    return {"value": round(vps, 5), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"{passes} of the {T} float64 MC-dropout forwards of one 64^3 volume ({t_pass:.2f} s/pass) + "
                      f"one T={T} entropy/MI reduction ({t_red:.2f} s), extrapolated to a whole volume; "
                      f"torch {torch.__version__} CPU, {cores} of {ncpu} host threads (fastest of {cand} on a 32^3 pass)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--volumes", type=int, default=32, help="64^3 volumes per GPU per step")
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="also time the host-inclusive variant (pinned host input, maps copied back)")
    ap.add_argument("--detail", type=str, default=None, help="write the per-kernel breakdown JSON here")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: values_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from values_amd import UNet3D, predict_uncertainty
    from values_amd.dist import MapGatherPipeline

    torch.manual_seed(123)  # reference seed (configs/dropout_config.yaml:8); default torch init = random weights
    model = UNet3D(num_classes=2, do_dropout=True).to(dev)
    V, T, S = args.volumes, args.T, args.size
    g = torch.Generator(device="cpu").manual_seed(123 + rank)
    x = torch.randn((V, 1, S, S, S), generator=g).to(dev)  # z-scored synthetic volumes, resident in HBM

    # maps are collected on rank 0 with the gather of step i overlapping the kernels of the following steps;
    # everything is flushed before the closing barrier, so the timed region contains every transfer
    pipe = MapGatherPipeline(world, rank, depth=2)

    def step(i):
        out = predict_uncertainty([model], x, n_pred=T, seeds=[i])
        return pipe.submit(out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    pipe.flush()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    pipe.flush()
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    pcie = None
    if args.pcie and world == 1:
        # host-inclusive variant: volumes start in pinned host memory, the five maps end there; never reported as `value`.
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
                "note": "pinned host -> device input, 5 maps device -> pinned host, copies on their own streams "
                        "(values_amd.HostPipeline), fill and drain of the 3-step pipeline included"}

    roof, detail, cpu = None, None, None
    if rank == 0 and not args.no_roofline:
        roof, detail = roofline_leg(model, x, T)
        if args.detail:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
            json.dump(detail, open(args.detail, "w"), indent=1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(T)
    if world > 1:
        dist.barrier()

    if rank == 0:
        vols = V * world * args.steps
        line = {
            "metric": "uncertainty-volumes/sec (64^3, T=10 MC-dropout)", "value": round(vols / dt, 3),
            "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {S}^3 1-channel volumes, UNet3D(initial_filter_size=8, 2 classes), T={T} "
                                   "MC-dropout passes + fused softmax/entropy/MI/argmax reduction",
                       "volumes_per_gpu_per_step": V, "samples_per_gpu_per_step": V * T,
                       "sharding": f"volumes over {world} rank(s); gather of maps to rank 0" if world > 1 else "single GPU",
                       "dropout": "hash bit generator, new seed every step", "weights": "torch default init, seed 123"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
