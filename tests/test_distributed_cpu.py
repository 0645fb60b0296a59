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


def _mesh_worker(rank, world, port, tmp):
    """The mesh form of the exchange (all-to-all, local sum, all-gather) must give what the all-reduce gives — on spans that do not divide
    by the world size, on a span shorter than one chunk per rank (some ranks own nothing), through region_ready / reduce_span / reduce_all."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from clip_lite_amd.runtime import Arena
    from clip_lite_amd.utils import distributed as D
    net = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Conv2d(3, 8, 3), torch.nn.Linear(5, 301))
    arena = Arena(net.named_parameters(), torch.device("cpu"), lowp=False)
    g = torch.Generator().manual_seed(70 + rank)
    mine = torch.randn(arena.total, generator=g)
    allg = [torch.empty_like(mine) for _ in range(world)]
    tdist.all_gather(allg, mine)
    want = sum(allg)
    ex = D.GradientExchange(arena, bucket_elems=128, algorithm="mesh")
    arena.flat_g.copy_(mine)
    for hi, lo in zip([arena.total, arena.total - 101, 701, 299], [arena.total - 101, 701, 299, 0]):
        ex.region_ready(lo, hi)
    assert ex.finish() == 1.0 / world
    assert torch.allclose(arena.flat_g, want, atol=1e-5), (arena.flat_g - want).abs().max()
    arena.flat_g.copy_(mine)
    ex.reduce_span(0, 100)                 # shorter than world x 64: the last ranks own no chunk of it
    ex.reduce_span(100, 1000)
    ex.reduce_span(1000, arena.total)
    ex.wait()
    assert torch.allclose(arena.flat_g, want, atol=1e-5)
    arena.flat_g.copy_(mine)
    ex.reduce_all(chunk_elems=777)
    assert torch.allclose(arena.flat_g, want, atol=1e-5)
    every = [torch.empty_like(mine) for _ in range(world)]
    tdist.all_gather(every, arena.flat_g)
    assert all(torch.equal(every[0], e) for e in every), "ranks hold different sums after the mesh exchange"
    tdist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_mesh_exchange_equals_allreduce_world3_gloo(tmp_path):
    port = 31500 + os.getpid() % 2000
    mp.spawn(_mesh_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(3))


# ---------------------------------------------------------------------------------------------------------------------------------
# Two-rank data-parallel TRAIN STEPS, end to end on the product's own update path (reference train.py:174-178,211-226): each rank
# back-propagates its own shard into the flat gradient arena, GradientExchange sums the arenas (gloo), FusedSGD clips by the global norm
# of the MEAN gradient and applies SGD(momentum, wd, per-name lr) + Lookahead with the schedule's multiplier — through the same
# clite_sumsq / clite_sgd_step kernels as on the GPU, here in their wave-simulator build (tests/simlib.py). The oracle is the reference
# optimiser stack (oracle/ref_model.py: torch.optim.SGD + Lookahead restatement, pinned by tests/golden/optim.npz) fed the average of
# the per-shard gradients, each shard run on its own (local BatchNorm statistics, local mean loss) — SURVEY.md §8(e)'s semantics.
def _tiny_net():
    torch.manual_seed(5)
    return torch.nn.ModuleDict({
        "image_encoder": torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 8)),
        "text_encoder": torch.nn.Sequential(torch.nn.Linear(10, 8), torch.nn.Tanh()),
        "loss": torch.nn.Linear(8, 1),
    })


def _tiny_loss(net, x, t):
    a, b = net["image_encoder"](x), net["text_encoder"](t)
    o = net["loss"](a * b).squeeze(-1)
    return torch.nn.functional.softplus(-o).mean() + torch.nn.functional.softplus(torch.roll(o, 1)).mean()      # local mean, local negatives


def _shard(step, rank, n=6):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randn(n, 12, generator=g), torch.randn(n, 10, generator=g)


STEPS, WARM, TOTAL, CLIP = 7, 2, 20, 0.05      # clipping active on some steps, Lookahead (k = 3) syncs twice, warm-up then cosine


def _dp_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    import simlib
    from clip_lite_amd import hip
    from clip_lite_amd.optim import FusedSGD, Lookahead
    from clip_lite_amd.optim.lr_scheduler import LinearWarmupCosineAnnealingLR
    from clip_lite_amd.runtime import Arena
    from clip_lite_amd.utils import distributed as D
    simlib.build_sim()
    hip._lib, hip._allow_host_tensors = hip._bind(C.CDLL(simlib.SIM_SO)), True      # the same csrc/optim_ops.hip, compiled for the host
    net = _tiny_net()
    if rank == 1:                                   # rank 1 starts from different parameters: the broadcast must repair that
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.5)
    arena = Arena(net.named_parameters(), torch.device("cpu"), lowp=False)

    class M:
        _rt = type("R", (), {"arena": arena})()
        buffers = staticmethod(lambda: [])          # BatchNorm buffers stay rank-local (DDP would broadcast rank 0's every forward)
        parameters = staticmethod(lambda: net.parameters())
    D.broadcast_parameters(M)
    groups = [{"params": [p], "lr": 0.2 if "image_encoder" in n else 1e-3, "weight_decay": 1e-4} for n, p in net.named_parameters()]
    opt = Lookahead(FusedSGD(groups, momentum=0.9), k=3, alpha=0.5)
    sched = LinearWarmupCosineAnnealingLR(opt, total_steps=TOTAL, warmup_steps=WARM)
    ex = D.GradientExchange(arena, bucket_elems=64)
    lo_img, hi_img = arena.index["image_encoder.0.weight"][0], arena.index["text_encoder.0.weight"][0]
    for step in range(STEPS):
        opt.zero_grad()
        _tiny_loss(net, *_shard(step, rank)).backward()            # autograd accumulates into the arena views (p.grad is flat_g)
        ex.region_ready(hi_img, arena.total)                       # heads + text encoder first, as backward finishes them ...
        ex.region_ready(lo_img, hi_img)                            # ... then the image encoder
        opt.optimizer.grad_prescale = ex.finish()                  # SUM over ranks now, 1/world inside the update kernel
        opt.clip_grad_norm(CLIP)
        opt.step()
        sched.step()
    allp = [torch.empty_like(arena.flat_p) for _ in range(world)]
    tdist.all_gather(allp, arena.flat_p)
    assert torch.equal(allp[0], allp[1]), "ranks diverged"
    if rank == 0:
        torch.save({n: p.detach().clone() for n, p in net.named_parameters()}, os.path.join(tmp, "dp_params.pt"))
    tdist.destroy_process_group()
    open(os.path.join(tmp, f"dp_ok{rank}"), "w").write("ok")


def test_two_rank_train_steps_match_shardwise_oracle(tmp_path):
    port = 31500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "dp_ok0").exists() and (tmp_path / "dp_ok1").exists()
    got = torch.load(tmp_path / "dp_params.pt")
    # the oracle: one process, shard by shard
    sys.path.insert(0, ROOT)
    from oracle import ref_model as O
    world = 2
    net = _tiny_net()
    replicas = [_tiny_net() for _ in range(world)]                 # per-rank BatchNorm buffers; parameters are tied to `net` below
    opt = O.build_optimizer(net.named_parameters(), cnn_lr=0.2, trans_lr=1e-3, lr=1e-3, k=3, alpha=0.5)
    for step in range(STEPS):
        mult = O.lr_multiplier("cosine", step, TOTAL, WARM, 0.0)
        for g in opt.param_groups:
            g["lr"] = g["initial_lr"] * mult
        mean = [torch.zeros_like(p) for p in net.parameters()]
        for r in range(world):
            replicas[r].load_state_dict({k: v for k, v in net.state_dict().items() if "running" not in k and "num_batches" not in k}, strict=False)
            grads = torch.autograd.grad(_tiny_loss(replicas[r], *_shard(step, r)), list(replicas[r].parameters()))
            for m, g in zip(mean, grads):
                m += g / world
        for p, m in zip(net.parameters(), mean):
            p.grad = m
        torch.nn.utils.clip_grad_norm_(list(net.parameters()), CLIP)
        opt.step()
    for n, p in net.named_parameters():
        assert torch.allclose(got[n], p.detach(), rtol=1e-5, atol=1e-6), (n, (got[n] - p).abs().max().item())


# ---------------------------------------------------------------------------------------------------------------------------------
# Pre-flight of the driver's 8-GPU run: world 8 over gloo, both exchange forms, on the REAL gradient regions of
# ResNet-50 + BERT-base + heads — the spans train_loop.TrainStep hands to GradientExchange.reduce_span in the captured step (heads, text
# encoder, image chain segments [layer4, layer3] / [layer2] / [layer1, stem]); the text encoder's 109.5 M elements do not divide by 8 x 64, so
# the mesh form runs with uneven splits and a padded gather. Gradients are small multiples of 1/8, so every summation order gives
# the same bits: the result must EQUAL the analytic sum on every rank, for both forms.
def _real_regions():
    sys.path.insert(0, ROOT)
    from clip_lite_amd.encoder import ImageEncoder, TextEncoder
    from clip_lite_amd.loss import JSDInfoMaxLoss
    from clip_lite_amd.model import VLInfoModel
    from clip_lite_amd.runtime import Arena
    import contextlib
    import io
    with torch.device("meta"), contextlib.redirect_stdout(io.StringIO()):
        M = VLInfoModel(TextEncoder(mode="train_sbert", num_hidden_layers=12), ImageEncoder("resnet50"), JSDInfoMaxLoss(2048, 768, "dot", 0.1, True, True),
                        "train_sbert", True)
    order, index, total = Arena.layout(M.named_parameters(), M.text_encoder.strans.contiguous_groups("text_encoder.strans."))
    A = Arena.__new__(Arena)
    A.names, A.index, A.total, A._pending = order, index, total, []
    pre = "image_encoder.img_encoder."
    img, l2, l3 = A.region("image_encoder."), A.region(pre + "layer2."), A.region(pre + "layer3.")
    spans = [A.region("loss."), A.region("text_encoder."), (l3[0], img[1]), (l2[0], l3[0]), (img[0], l2[0])]      # the captured step's hand-over order
    return A, spans


def _fill(flat, rank, world=None):
    """flat[i] = a[i] + rank * b[i] with a = ((7 i) mod 257 - 128) / 8 and b = (i mod 5 - 2) / 4, in slabs (no 1.2 GB index tensor); with `world`
    given, the sum over ranks instead: world * a + world (world - 1) / 2 * b. Every value is a small multiple of 1/8: sums are exact in f32."""
    step = 1 << 24
    ca, cb = (1.0, float(rank)) if world is None else (float(world), world * (world - 1) / 2.0)
    for s in range(0, flat.numel(), step):
        e = min(flat.numel(), s + step)
        idx = torch.arange(s, e, dtype=torch.int64)
        a = (((idx * 7) % 257) - 128).to(torch.float32) / 8
        b = ((idx % 5) - 2).to(torch.float32) / 4
        flat[s:e] = ca * a + cb * b


def _world8_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from clip_lite_amd.utils import distributed as D
    A, spans = _real_regions()
    assert A.total == 156157056 and sum(hi - lo for lo, hi in spans) == A.total          # the five spans tile the whole arena
    assert any((hi - lo) % (world * 64) for lo, hi in spans), "no uneven mesh split in this layout: the test would not cover the padded gather"
    A.flat_g = torch.empty(A.total, dtype=torch.float32)
    want = torch.empty(A.total, dtype=torch.float32)
    _fill(want, 0, world)
    # the mesh form — this package's own chunking, uneven splits and padded gather — on all five spans at full size; the all-reduce form (one
    # library call per span, nothing of ours to size) on the two image segments that are cheap over gloo's loopback TCP (the CPU suite's time
    # budget: the full-size all-reduce of 625 MB across 8 gloo processes alone takes ~90 s here)
    for algorithm, which in (("mesh", spans), ("allreduce", spans[3:])):
        ex = D.GradientExchange(A, algorithm=algorithm)
        assert ex.world == world and ex.rank == rank
        _fill(A.flat_g, rank)
        for lo, hi in which:
            ex.reduce_span(lo, hi)
        ex.wait()
        for lo, hi in which:
            assert torch.equal(A.flat_g[lo:hi], want[lo:hi]), (algorithm, lo, hi, (A.flat_g[lo:hi] - want[lo:hi]).abs().max().item())
    tdist.barrier()
    tdist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_world8_both_exchange_forms_on_the_real_resnet50_bert_regions_gloo(tmp_path):
    port = 33500 + os.getpid() % 2000
    mp.spawn(_world8_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(8))
