"""Process-group helpers with the reference's names and no-distributed fallbacks (reference utils/distributed.py:15-171), on
RCCL over xGMI, plus the gradient exchange that replaces DistributedDataParallel (reference train.py:174-178).

One process per GPU. `launch` spawns them like the reference (mp.spawn, tcp rendezvous); when the processes were started by
`torch.distributed.run` (RANK/WORLD_SIZE in the environment) it joins that rendezvous instead. Backend "nccl" is RCCL on ROCm.

Gradient exchange: pure data parallel — every rank computes local BatchNorm statistics and local roll-by-one negatives
(reference loss.py:214-216), so the only collective is the mean of the per-rank gradients. `GradientExchange` all-reduces
contiguous regions of the flat gradient arena on a side HIP stream as soon as the backward executor reports them complete
(loss heads first, then the text encoder layer by layer, then the ResNet stages), so the exchange overlaps the rest of
backward; the 1/world_size factor is folded into the fused update kernel. DDP's per-forward buffer broadcast and the
find_unused_parameters bitmap are dropped: BatchNorm buffers are rank-local by design and every parameter is used.
"""
import os
from typing import Callable, Dict, Tuple, Union

import torch
from torch import distributed as dist
from torch import multiprocessing as mp


def _backend():
    return "nccl" if torch.cuda.is_available() else "gloo"


def launch(job_fn: Callable, num_machines: int = 1, num_gpus_per_machine: int = 1, machine_rank: int = 0,
           dist_url: str = "tcp://127.0.0.1:23456", args=()):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        # started by torch.distributed.run: one process per GPU already exists — never spawn from here, whatever --num-gpus-per-machine says
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        if int(os.environ["WORLD_SIZE"]) > 1:
            dist.init_process_group(backend=_backend(), init_method="env://")
            synchronize()
        return job_fn(*args)
    assert torch.cuda.is_available(), "No GPU visible: cannot launch distributed processes."
    world_size = num_machines * num_gpus_per_machine
    if world_size > 1:
        mp.spawn(_job_worker, nprocs=num_gpus_per_machine,
                 args=(job_fn, world_size, num_gpus_per_machine, machine_rank, dist_url, args), daemon=False)
    else:
        torch.cuda.set_device(0)
        job_fn(*args)


def _job_worker(local_rank: int, job_fn: Callable, world_size: int, num_gpus_per_machine: int, machine_rank: int, dist_url: str, args: Tuple):
    global_rank = machine_rank * num_gpus_per_machine + local_rank
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=_backend(), init_method=dist_url, world_size=world_size, rank=global_rank)
    synchronize()
    job_fn(*args)


def synchronize() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_master_process() -> bool:
    return get_rank() == 0


def average_across_processes(t: Union[torch.Tensor, Dict[str, torch.Tensor]]):
    if dist.is_available() and dist.is_initialized():
        if isinstance(t, torch.Tensor):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t /= get_world_size()
        elif isinstance(t, dict):
            for k in t:
                dist.all_reduce(t[k], op=dist.ReduceOp.SUM)
                t[k] /= get_world_size()


def gpu_mem_usage() -> int:
    return torch.cuda.max_memory_allocated() // 1048576 if torch.cuda.is_available() else 0


def broadcast_parameters(model) -> None:
    """What DDP's constructor does once (reference train.py:176-178): rank 0's parameters and buffers become everyone's."""
    if get_world_size() == 1:
        return
    rt = getattr(model, "_rt", None)
    if rt is not None:
        dist.broadcast(rt.arena.flat_p, src=0)
        rt.arena.refresh_lowp()
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    for b in model.buffers():
        dist.broadcast(b, src=0)


class GradientExchange:
    """Mean-of-gradients over the flat f32 gradient arena, overlapped with backward on a side stream.

    `region_ready(lo, hi)` is called (from the autograd thread) by the backward executors when flat_g[lo:hi] is final; regions
    are coalesced into buckets of at least `bucket_elems` and all-reduced (SUM) asynchronously. `finish()` makes the compute
    stream wait for the exchange and sets the optimizer's gradient pre-scale to 1/world_size."""

    def __init__(self, arena, bucket_elems: int = 16 * 1024 * 1024, group=None, algorithm: str = "allreduce"):
        """algorithm: "allreduce" — one RCCL all-reduce (SUM) per region (RCCL's rings / trees over the xGMI links); "mesh" — the
        direct form for a fully connected node: all-to-all (rank w receives chunk w of every rank's region: all 7 links of a GPU carry
        1/8 of the region at once), a local fixed-order sum (clite_sum_slices), all-gather of the reduced chunks. Per link the mesh form
        moves 2 x region / world instead of a ring's 2 x (world - 1) / world x region over one link direction."""
        assert algorithm in ("allreduce", "mesh")
        self.arena, self.bucket_elems, self.group, self.algorithm = arena, bucket_elems, group, algorithm
        self.world = get_world_size()
        self.rank = get_rank()
        self._scratch = {}
        self.on_gpu = arena.flat_g.is_cuda
        self.stream = torch.cuda.Stream() if self.on_gpu else None
        self._pending = []      # [lo, hi) regions not yet sent
        self._works = []
        self._covered = []     # [lo, hi) regions already handed to RCCL this step
        self.defer = False     # True: ignore region_ready (modules run several times per step); finish() reduces the whole arena

    def _buffer(self, name, n, like):
        t = self._scratch.get(name)
        if t is None or t.numel() < n:
            t = self._scratch[name] = torch.empty(n, dtype=like.dtype, device=like.device)
        return t

    def _mesh_sum(self, buf):
        """buf (a slice of flat_g) <- sum over ranks, by all-to-all / local sum / all-gather. Enqueued on the current stream (RCCL) or run
        blocking (gloo: the CPU tests)."""
        W, r, n = self.world, self.rank, buf.numel()
        chunk = ((n + W - 1) // W + 63) // 64 * 64                       # 256-byte chunks; only the tail chunks are short or empty
        sizes = [max(0, min(chunk, n - w * chunk)) for w in range(W)]
        mine = sizes[r]
        recv = self._buffer("recv", W * chunk, buf)[:W * chunk]
        dist.all_to_all_single(recv[:W * mine], buf, output_split_sizes=[mine] * W, input_split_sizes=sizes, group=self.group)
        red = self._buffer("red", chunk, buf)[:chunk]
        if mine:
            if self.on_gpu:
                from .. import hip
                hip.sum_slices(recv, W, mine, mine, red)
            else:
                torch.sum(recv[:W * mine].view(W, mine), dim=0, out=red[:mine])
        if n == W * chunk:
            dist.all_gather_into_tensor(buf, red, group=self.group)
        else:
            full = self._buffer("full", W * chunk, buf)[:W * chunk]
            dist.all_gather_into_tensor(full, red, group=self.group)
            buf.copy_(full[:n])               # chunk w sits at w * chunk in both layouts; only the tail is padding

    def _sum(self, buf):
        """Start the sum over ranks of `buf` on the current stream; returns a work handle or None."""
        if self.algorithm == "mesh":
            self._mesh_sum(buf)
            return None
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _send(self, lo, hi):
        buf = self.arena.flat_g[lo:hi]
        if self.on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                w = self._sum(buf)
        else:
            w = self._sum(buf)
        if w is not None:
            self._works.append(w)
        self._covered.append((lo, hi))

    def region_ready(self, lo, hi):
        if self.world == 1 or hi <= lo or self.defer:
            return
        # merge with an adjacent pending region when possible
        if self._pending and self._pending[-1][0] == hi:
            self._pending[-1][0] = lo
        elif self._pending and self._pending[-1][1] == lo:
            self._pending[-1][1] = hi
        else:
            self._pending.append([lo, hi])
        while self._pending and self._pending[0][1] - self._pending[0][0] >= self.bucket_elems:
            lo0, hi0 = self._pending.pop(0)
            self._send(lo0, hi0)

    def reduce_span(self, lo, hi, after=None):
        """All-reduce (SUM) flat_g[lo:hi] on the exchange's side stream once the work already enqueued on stream `after` (default:
        the current stream) is done; returns immediately. `wait()` later orders the caller's stream after every such reduction."""
        if self.world == 1 or hi <= lo:
            return
        buf = self.arena.flat_g[lo:hi]
        if self.on_gpu:
            self.stream.wait_stream(after if after is not None else torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                w = self._sum(buf)
        else:
            w = self._sum(buf)
        if w is not None:
            self._works.append(w)

    def wait(self):
        for w in self._works:
            w.wait()
        if self.on_gpu and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._works = []

    def reduce_all(self, chunk_elems: int = 64 * 1024 * 1024):
        """Non-overlapped form used between the two hipGraphs of a captured step: all-reduce (SUM) the whole gradient arena on
        the caller's stream, in chunks of at most `chunk_elems` (256 MB of f32) so RCCL pipelines the rings. The update kernel
        applies 1/world_size."""
        if self.world == 1:
            return
        total = self.arena.total
        for lo in range(0, total, chunk_elems):
            w = self._sum(self.arena.flat_g[lo:min(total, lo + chunk_elems)])
            if w is not None:
                w.wait()

    def finish(self):
        """Flush pending regions, then exchange every part of the arena that was never reported, so each element is reduced
        exactly once per step whatever the executors announced. Returns the factor that turns the SUM into the mean."""
        if self.world == 1:
            return 1.0
        self.arena.join()       # backward executors may have written gradients on other streams than the caller's
        for lo, hi in self._pending:
            self._send(lo, hi)
        self._pending = []
        pos = 0
        for lo, hi in sorted(self._covered):
            if lo < pos:          # a region summed twice would silently double its gradients
                raise RuntimeError(f"GradientExchange: gradient region [{lo}, {hi}) was handed over twice in one step (overlaps [.., {pos}))")
            if lo > pos:
                self._send(pos, lo)
            pos = max(pos, hi)
        if pos < self.arena.total:
            self._send(pos, self.arena.total)
        for w in self._works:
            w.wait()
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._works, self._covered = [], []
        return 1.0 / self.world
