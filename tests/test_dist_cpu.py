"""CPU, world_size 2, gloo: the N>1 path of bench.py (shard volumes, gather maps on rank 0)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from values_amd.dist import gather_maps, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_maps(lo, hi):
    v = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1)
    sp = (4, 4, 4)
    return {"pred_entropy": v.expand(-1, *sp) + 0.1, "aleatoric_uncertainty": v.expand(-1, *sp) + 0.2,
            "epistemic_uncertainty": v.expand(-1, *sp) + 0.3,
            "mean_softmax": torch.stack([v.expand(-1, *sp) + 0.4, v.expand(-1, *sp) + 0.5], 1),
            "pred_seg_mean": (torch.arange(lo, hi).view(-1, 1, 1, 1).expand(-1, *sp) % 2).to(torch.uint8)}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_vol = 6
    lo, hi = shard_range(n_vol, world, rank)
    for _ in range(2):  # twice: the receive buffers are cached
        res = gather_maps(_fake_maps(lo, hi), world, rank)
    if rank == 0:
        ref = _fake_maps(0, n_vol)
        ok = all(torch.equal(res[k], ref[k]) for k in ref)
        q.put(ok)
    else:
        q.put(res is None)
    dist.barrier()
    dist.destroy_process_group()


def _worker_pipeline(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from values_amd.dist import MapGatherPipeline
    pipe = MapGatherPipeline(world, rank, depth=2)
    pipe._use_all_gather = bool(int(os.environ.get("VX_TEST_ALL_GATHER", "0")))   # the fallback path of a backend without gather
    got = []
    n_vol = 4
    for step in range(5):    # more steps than buffers: every send / receive buffer is reused
        lo, hi = shard_range(n_vol, world, rank)
        maps = _fake_maps(lo + 10 * step, hi + 10 * step)
        r = pipe.submit(maps)
        for v in maps.values():
            v.zero_() if v.is_contiguous() else None    # the caller may reuse its tensors right away
        if r is not None:
            got.append({k: v.clone() for k, v in r.items()})   # views of a receive buffer: valid until it is reused
    got += [{k: v.clone() for k, v in r.items()} for r in pipe.flush() if r is not None]   # (no submit in between)
    if rank == 0:
        ok = len(got) == 5
        for step, res in enumerate(got):
            ref = {}
            parts = [_fake_maps(*[x + 10 * step for x in shard_range(n_vol, world, r)]) for r in range(world)]
            for k in parts[0]:
                ref[k] = torch.cat([p[k] for p in parts], 0)
            ok = ok and all(torch.equal(res[k], ref[k]) for k in ref)
        q.put(ok)
    else:
        q.put(len(got) == 0)
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(target):
    """spawn two ranks; a rendezvous that fails (the free port was taken in between, a slow host) is retried on a new port"""
    import queue
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = []
        for _ in range(240):
            try:
                res.append(q.get(timeout=1))
            except queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
            if len(res) == len(procs):
                break
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if len(res) == len(procs) and all(p.exitcode == 0 for p in procs):
            return res
    raise AssertionError(f"world-2 gloo run failed three times (last results {res}, exit codes {[p.exitcode for p in procs]})")


@pytest.mark.parametrize("all_gather", ["0", "1"])
def test_map_gather_pipeline_world2_gloo(all_gather, monkeypatch):
    """the overlapped gather of bench.py --gpus N: order of results, buffer reuse, flush (and the all_gather fallback)"""
    monkeypatch.setenv("VX_TEST_ALL_GATHER", all_gather)
    assert all(_run_world2(_worker_pipeline))


def test_map_gather_pipeline_world1():
    from values_amd.dist import MapGatherPipeline
    pipe = MapGatherPipeline(1, 0, depth=2)
    outs = [pipe.submit({"step": i}) for i in range(4)]
    assert outs[:2] == [None, None] and outs[2] == {"step": 0} and outs[3] == {"step": 1}
    assert pipe.flush() == [{"step": 2}, {"step": 3}]


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_maps_world2_gloo():
    assert all(_run_world2(_worker))


def test_ensemble_work_items_cover_every_member_and_volume_once():
    from values_amd.dist import ensemble_work_items
    for members, vols, world in [(5, 8, 8), (5, 1, 8), (3, 4, 2), (2, 7, 1), (8, 8, 8), (10, 3, 4)]:
        per_rank = ensemble_work_items(members, vols, world)
        assert len(per_rank) == world
        seen = {}
        for items in per_rank:
            for m, lo, hi in items:
                for v in range(lo, hi):
                    assert (m, v) not in seen
                    seen[(m, v)] = 1
        assert len(seen) == members * vols
        busy = sum(1 for items in per_rank if items)
        assert busy == min(world, len(seen) if vols == 1 else world) or busy >= min(world, members)


def test_ensemble_work_items_are_balanced_over_the_ranks():
    """config C3 (5 members, 8 ranks): every rank carries the same number of (member, volume block) items and the same
    number of member-volumes (round 2 dealt 10 items over 8 ranks: 62 % ideal efficiency by construction)"""
    from values_amd.dist import ensemble_work_items
    for members, vols, world in [(5, 128, 8), (5, 16, 8), (5, 8, 8), (5, 32, 4), (5, 32, 2), (3, 16, 8), (5, 64, 1)]:
        per_rank = ensemble_work_items(members, vols, world)
        counts = [len(it) for it in per_rank]
        work = [sum(hi - lo for _, lo, hi in it) for it in per_rank]
        assert max(counts) == min(counts), (members, vols, world, counts)
        assert max(work) == min(work), (members, vols, world, work)


def test_bench_parent_counts_gpus_without_a_gpu_runtime():
    """bench.py --gpus N without a launcher: the parent that starts the ranks must not import torch or touch HIP /
    amdsmi -- it reads the kfd topology from sysfs (None where there is none, as in the build container)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.kfd_gpu_count(); "
            "assert n is None or n >= 0; assert 'torch' not in sys.modules, 'the launching parent imported torch'; "
            "print('ok', n)") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr


def test_gather_maps_world1_is_identity():
    m = _fake_maps(0, 3)
    assert gather_maps(m, 1, 0) is m
