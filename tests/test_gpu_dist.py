"""GPU, world_size 2 on ONE device (gloo moves device tensors; RCCL refuses two ranks on one GPU): the member-sharded
ensemble's real exchange step -- values_amd.dist.ensemble_uncertainty_sharded's sum-reduce of sufficient statistics --
and the map gather, with the HIP kernels on both ranks.  On a node with at least two GPUs the same exchange runs on the
`nccl` backend (= RCCL over xGMI), one rank per device (skipped on the 1-GPU boxes of the build pool)."""
import os

import numpy as np
import pytest
import torch

from tests.test_dist_cpu import _run_world2

pytestmark = pytest.mark.gpu


def _worker_sharded(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from tests.test_gpu_unet3d import KEYS, make_model
        from values_amd import predict_uncertainty
        from values_amd.dist import ensemble_uncertainty_sharded, gather_maps, shard_range
        from tests.formula import formula_volume
        models = [make_model(seed_tag=s, do_dropout=False) for s in range(3)]
        x = torch.from_numpy(np.concatenate([formula_volume((1, 1, 16, 16, 16), tag=70 + i) for i in range(3)], 0)).float().cuda()
        sh = ensemble_uncertainty_sharded(models, x, world=world, rank=rank, n_pred=1)
        ok = True
        if rank == 0:
            one = predict_uncertainty(models, x, n_pred=1)
            ok = all((sh[k] - one[k]).abs().max().item() < 2e-6 for k in KEYS + ("mean_softmax",))
            ok = ok and torch.equal(sh["pred_seg_mean"], one["pred_seg_mean"])
        else:
            ok = sh is None
        # volume-sharded MC-dropout + gather: rank r runs its shard, rank 0 ends up with every volume's maps
        drop = make_model(do_dropout=True)
        lo, hi = shard_range(3, world, rank)
        mine = predict_uncertainty([drop], x[lo:hi], n_pred=4, seeds=[11 + rank])
        allm = gather_maps(mine, world, rank)
        if rank == 0:
            for r in range(world):
                l2, h2 = shard_range(3, world, r)
                ref = predict_uncertainty([drop], x[l2:h2], n_pred=4, seeds=[11 + r])
                ok = ok and all(torch.equal(allm[k][l2:h2], ref[k]) for k in KEYS + ("softmax_variance", "mean_softmax", "pred_seg_mean"))
        else:
            ok = ok and allm is None
        torch.cuda.synchronize()
        q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_member_sharded_ensemble_reduce_and_map_gather_world2():
    assert all(_run_world2(_worker_sharded))


def _worker_nccl(rank, world, port, q):
    """one rank per GPU, backend nccl (RCCL): member-sharded ensemble (sum-reduce) + the overlapped map gather of
    bench.py --gpus N against the single-GPU results computed on rank 0"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from tests.test_gpu_unet3d import KEYS, make_model
        from values_amd import predict_uncertainty
        from values_amd.dist import MapGatherPipeline, ensemble_uncertainty_sharded, shard_range
        from tests.formula import formula_volume
        assert dist.get_world_size() == world and str(dist.get_backend()).lower() == "nccl"
        models = [make_model(seed_tag=s, do_dropout=False) for s in range(3)]
        x = torch.from_numpy(np.concatenate([formula_volume((1, 1, 16, 16, 16), tag=70 + i) for i in range(4)], 0)).float().to(dev)
        sh = ensemble_uncertainty_sharded(models, x, world=world, rank=rank, n_pred=1)
        ok = True
        if rank == 0:
            one = predict_uncertainty(models, x, n_pred=1)
            ok = all((sh[k] - one[k]).abs().max().item() < 2e-6 for k in KEYS + ("mean_softmax",))
            ok = ok and torch.equal(sh["pred_seg_mean"], one["pred_seg_mean"])
        else:
            ok = sh is None
        # volume shards + MapGatherPipeline (async gathers, buffer reuse, flush): bit for bit the single-GPU maps
        drop = make_model(do_dropout=True)
        lo, hi = shard_range(4, world, rank)
        pipe = MapGatherPipeline(world, rank, depth=2)
        got = []
        for step in range(5):
            r = pipe.submit(predict_uncertainty([drop], x[lo:hi], n_pred=4, seeds=[100 * step + rank]))
            if r is not None:
                got.append({k: v.clone() for k, v in r.items()})
        got += [{k: v.clone() for k, v in r.items()} for r in pipe.flush() if r is not None]
        if rank == 0:
            ok = ok and len(got) == 5
            for step, res in enumerate(got):
                for r in range(world):
                    l2, h2 = shard_range(4, world, r)
                    ref = predict_uncertainty([drop], x[l2:h2], n_pred=4, seeds=[100 * step + r])
                    ok = ok and all(torch.equal(res[k][l2:h2], ref[k])
                                    for k in KEYS + ("softmax_variance", "mean_softmax", "pred_seg_mean"))
        else:
            ok = ok and len(got) == 0
        torch.cuda.synchronize()
        q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_nccl_backend_two_gpus_sharded_ensemble_and_gather_pipeline():
    """SURVEY section 4, pyramid level 4: the RCCL paths themselves (dist.reduce / async dist.gather on `nccl`)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on one node (the build pool's boxes have one); covered under gloo above")
    assert all(_run_world2(_worker_nccl))
