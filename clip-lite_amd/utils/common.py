"""Job setup shared by the entry points (reference utils/common.py:14-159): the infinite batch iterator, seeding / log sinks /
config dump, and the common command-line flags (same names and defaults). loguru is not available in the target image, so
logging goes through the stdlib `logging` module with the reference's sinks (stdout on the master process, a per-rank file
when world_size > 1)."""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

from . import distributed as dist

logger = logging.getLogger("clip_lite_amd")


class _Stager:
    """Host -> device hand-over of a batch on a dedicated copy stream (pinned source buffers make the copies asynchronous DMA), so the
    77 MB of fp32 images of a 128-pair batch cross PCIe while the previous step computes instead of in front of it."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def stage(self, batch):
        if self.stream is None:
            return {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}, None
        with torch.cuda.stream(self.stream):
            out = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def hand_over(self, staged):
        out, ev = staged
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in out.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)        # allocated on the copy stream, consumed on the compute stream
        return out


def cycle(dataloader, device, start_iteration: int = 0, type: str = "normal", prefetch: int = 2):
    """Yield batches forever, moving every tensor to `device`; a DistributedSampler is re-seeded with the running iteration
    at each pass over the data (reference utils/common.py:22-37: same order of batches, same shuffle seeds). Unlike the reference's
    `.to(device)` in front of each step, up to `prefetch` batches are staged ahead on a copy stream; prefetch = 0 restores
    the copy-then-compute order."""
    from collections import deque
    stager = _Stager(device)
    pending = deque()
    loaded = start_iteration          # batches taken from the loader so far = the reference's `iteration` at its set_epoch calls
    while True:
        sampler = getattr(dataloader, "sampler", None)
        if isinstance(sampler, torch.utils.data.DistributedSampler):
            logger.info(f"Beginning new epoch, setting shuffle seed {loaded}")
            sampler.set_epoch(loaded)
            if type == "clusters":
                sampler.dataset.update_iter(loaded)
        for batch in dataloader:
            pending.append(stager.stage(batch))
            loaded += 1
            if len(pending) > prefetch:
                yield stager.hand_over(pending.popleft())


def common_setup(_C, _A: argparse.Namespace, job_type: str = "pretrain"):
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(_C.RANDOM_SEED)
    random.seed(_C.RANDOM_SEED)
    np.random.seed(_C.RANDOM_SEED)
    out_dir = _A.checkpoints_dir + _C.RUN_ID
    os.makedirs(out_dir, exist_ok=True)
    _C.dump(os.path.join(out_dir, f"{job_type}_config.yaml"))
    logger.handlers.clear()
    logger.setLevel(logging.INFO)
    logger.propagate = False
    if world > 1:
        fh = logging.FileHandler(os.path.join(out_dir, f"log-rank{rank}.txt"))
        fh.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
        logger.addHandler(fh)
    if dist.is_master_process():
        sh = logging.StreamHandler(sys.stdout)
        sh.setFormatter(logging.Formatter("%(asctime)s: %(message)s"))
        logger.addHandler(sh)
    logger.info(f"Rank of current process: {rank}. World size: {world}")
    logger.info(str(_C))
    logger.info("Command line args:")
    for arg in vars(_A):
        logger.info("{:<20}: {}".format(arg, getattr(_A, arg)))


def common_parser(description: str = "") -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=description)
    parser.add_argument("--config", metavar="FILE", help="Path to a pretraining config file.")
    parser.add_argument("--config-override", nargs="*", default=[], help="A list of key-value pairs to modify pretraining config params.")
    parser.add_argument("--checkpoints-dir", default="saves/checkpoints", help="Path to a directory to serialize checkpoints and save job logs.")
    group = parser.add_argument_group("Compute resource management arguments.")
    group.add_argument("--cpu-workers", type=int, default=4, help="Number of CPU workers per GPU to use for data loading.")
    group.add_argument("--num-machines", type=int, default=1, help="Number of machines used in distributed training.")
    group.add_argument("--num-gpus-per-machine", type=int, default=8, help="Number of GPUs per machine with IDs as (0, 1, 2 ...).")
    group.add_argument("--machine-rank", type=int, default=0, help="Rank of the machine, integer in [0, num_machines).")
    group.add_argument("--dist-url", default="tcp://127.0.0.1:23456", help="URL of the master process in distributed training.")
    return parser


class GradScaler:
    """Loss-scaler object with torch.cuda.amp.GradScaler's surface, kept so the train loop and the checkpoint layout
    ({"model","optimizer","scheduler","scaler","iteration"}, reference checkpointing.py:80-94) stay those of the reference.
    bf16 keeps fp32's exponent range, so no loss scaling happens: scale() is the identity."""

    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def scale(self, loss):
        return loss

    def unscale_(self, optimizer):
        pass

    def step(self, optimizer, *a, **k):
        return optimizer.step(*a, **k)

    def update(self):
        pass

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass
