"""Checkpoint files with the reference's layout (reference utils/checkpointing.py:12-222): a dict of `state_dict()`s keyed by the
names the manager was built with (`model`, `optimizer`, `scheduler`, `scaler`) plus `iteration`, saved as
`checkpoint_{iteration}.pth`; `climax_step` saves the model only; `load` restores whatever keys it finds and returns the
iteration. Tensors are moved to the CPU before serialisation so files do not depend on the GPU arena's flat storage."""
import copy
import pathlib
from typing import Any, Dict, List, Optional

import torch

from . import distributed as dist


def _unwrap(obj):
    return obj.module if hasattr(obj, "module") and hasattr(obj.module, "state_dict") else obj


def _to_cpu(x):
    if torch.is_tensor(x):
        return x.detach().to("cpu").contiguous().clone()
    if isinstance(x, dict):
        return type(x)((k, _to_cpu(v)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cpu(v) for v in x)
    return x


class CheckpointManager(object):
    def __init__(self, serialization_dir: str = "/tmp", keep_recent: int = 1000, **checkpointables: Any):
        self.serialization_dir = pathlib.Path(serialization_dir)
        self.keep_recent = keep_recent
        self.checkpointables = copy.copy(checkpointables)
        self._best_metric: float = -1e-12
        self._best_ckpt: Dict[str, Any] = {}
        self._recent_iterations: List[int] = []

    def _state_dict(self, only: Optional[str] = None):
        return {k: _to_cpu(_unwrap(v).state_dict()) for k, v in self.checkpointables.items() if only is None or k == only}

    def _save(self, state, iteration):
        state["iteration"] = iteration
        torch.save(state, self.serialization_dir / f"checkpoint_{iteration}.pth")
        self._recent_iterations.append(iteration)
        if len(self._recent_iterations) > self.keep_recent:
            self.remove_earliest_checkpoint()

    def step(self, iteration: int, metric: Optional[float] = None):
        state = self._state_dict()
        if metric is not None and metric > self._best_metric:
            self._best_metric = metric
            self._best_ckpt = copy.copy(state)
            self._best_ckpt["iteration"] = iteration
        self._save(state, iteration)
        if self._best_metric != -1e-12:
            torch.save(self._best_ckpt, self.serialization_dir / "checkpoint_best.pth")

    def climax_step(self, iteration: int):
        self._save(self._state_dict("model"), iteration)

    def remove_earliest_checkpoint(self):
        earliest = self._recent_iterations.pop(0)
        (self.serialization_dir / f"checkpoint_{earliest}.pth").unlink()

    def load(self, checkpoint_path: str):
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        iteration = checkpoint.pop("iteration", -1)
        missing = []
        for key, obj in self.checkpointables.items():
            if key in checkpoint:
                _unwrap(obj).load_state_dict(checkpoint[key])
            else:
                missing.append(key)
        if missing:
            print(f"Rank {dist.get_rank()}: checkpointables not found in {checkpoint_path}: {missing}")
        return iteration
