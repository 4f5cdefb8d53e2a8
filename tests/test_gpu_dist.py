"""GPU, world_size 2 on ONE device (gloo moves device tensors; RCCL refuses two ranks on one GPU): the member-sharded
ensemble's real exchange step -- values_amd.dist.ensemble_uncertainty_sharded's sum-reduce of sufficient statistics --
and the map gather, with the HIP kernels on both ranks."""
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
        from values_amd.formula import formula_volume
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
