"""world_size-2 gloo tests (CPU) of the data-parallel path: the gradient exchange over the flat arena equals the mean of the
per-rank gradients whatever order regions are reported in, and rank 0's parameters are broadcast at start (the collectives
DistributedDataParallel performs at reference train.py:174-178)."""
import os
import sys

import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from clip_lite_amd.runtime import Arena
    from clip_lite_amd.utils import distributed as D
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Conv2d(3, 8, 3), torch.nn.Linear(5, 300))
    arena = Arena(net.named_parameters(), torch.device("cpu"), lowp=False)

    class M:                                           # what broadcast_parameters expects of a model
        _rt = type("R", (), {"arena": arena})()
        buffers = staticmethod(lambda: [])
        parameters = staticmethod(lambda: net.parameters())
    D.broadcast_parameters(M)
    ref0 = [torch.empty_like(arena.flat_p) for _ in range(world)]
    tdist.all_gather(ref0, arena.flat_p)
    assert all(torch.equal(ref0[0], r) for r in ref0), "parameters differ across ranks after broadcast"
    assert D.get_world_size() == world and D.get_rank() == rank and D.is_master_process() == (rank == 0)

    ex = D.GradientExchange(arena, bucket_elems=256)
    g = torch.Generator().manual_seed(7 + rank)
    arena.flat_g.copy_(torch.randn(arena.total, generator=g))
    mine = arena.flat_g.clone()
    allg = [torch.empty_like(mine) for _ in range(world)]
    tdist.all_gather(allg, mine)
    # report regions back-to-front in uneven pieces, as a backward pass does; leave one region unreported
    cuts = [arena.total, arena.total - 100, 700, 300]
    for hi, lo in zip(cuts[:-1], cuts[1:]):
        ex.region_ready(lo, hi)
    scale = ex.finish()
    assert scale == 1.0 / world
    assert torch.allclose(arena.flat_g * scale, sum(allg) / world, atol=1e-6)
    # second step: nothing reported at all -> the whole arena is exchanged once
    arena.flat_g.copy_(mine)
    assert torch.allclose(arena.flat_g * ex.finish(), sum(allg) / world, atol=1e-6)
    # third step: deferred mode (hard negatives / augmented views make the encoders run several times per step, so a module's "ready" report
    # is premature): reports are ignored, more gradient arrives afterwards, finish() still reduces every element exactly once
    ex.defer = True
    arena.flat_g.copy_(mine * 0.5)
    ex.region_ready(300, arena.total)
    arena.flat_g.add_(mine * 0.5)                       # a second backward call adds to the same region after the report
    assert torch.allclose(arena.flat_g * ex.finish(), sum(allg) / world, atol=1e-6)
    ex.defer = False
    t = {"a": torch.tensor(float(rank))}
    D.average_across_processes(t)
    assert abs(t["a"].item() - (world - 1) / 2) < 1e-6
    D.synchronize()
    tdist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_gradient_exchange_world2_gloo(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
