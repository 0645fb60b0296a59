"""Learning-rate schedules of the reference (optim/lr_scheduler.py): linear warmup followed by no decay (:19-70), multi-step
decay (:73-112), linear decay (:115-152) or cos^2 annealing (:155-202). Same class names, constructor arguments and
``_lr_multiplier``; they drive any optimizer with ``param_groups`` through torch's LambdaLR."""
import bisect
import math
from typing import List

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


def _inner(optimizer):
    """LambdaLR insists on a torch Optimizer; unwrap a Lookahead-style wrapper."""
    return optimizer if isinstance(optimizer, Optimizer) else optimizer.optimizer


class LinearWarmupNoDecayLR(LambdaLR):
    def __init__(self, optimizer, total_steps: int, warmup_steps: int, last_epoch: int = -1):
        assert warmup_steps < total_steps, "Warmup steps should be less than total steps."
        self.tsteps, self.wsteps = total_steps, warmup_steps
        super().__init__(_inner(optimizer), self._lr_multiplier, last_epoch)

    def _lr_multiplier(self, step: int) -> float:
        multiplier = step / float(max(1, self.wsteps)) if step < self.wsteps else 1
        return max(0, multiplier)


class LinearWarmupMultiStepLR(LambdaLR):
    def __init__(self, optimizer, total_steps: int, warmup_steps: int, milestones: List[int], gamma: float = 0.1, last_epoch: int = -1):
        self.wsteps, self.milestones, self.gamma = warmup_steps, milestones, gamma
        self.milestones_so_far = 0
        assert milestones == sorted(milestones), "milestones must be increasing"
        assert milestones[0] > warmup_steps, "first milestone must be after warmup"
        assert milestones[-1] < total_steps, "last milestone must be less than total steps"
        super().__init__(_inner(optimizer), self._lr_multiplier, last_epoch)

    def _lr_multiplier(self, step: int) -> float:
        if step < self.wsteps:
            multiplier = step / float(max(1, self.wsteps))
        else:
            multiplier = self.gamma ** bisect.bisect_right(self.milestones, step)
        return max(0, multiplier)


class LinearWarmupLinearDecayLR(LambdaLR):
    def __init__(self, optimizer, total_steps: int, warmup_steps: int, last_epoch: int = -1):
        assert warmup_steps < total_steps, "Warmup steps should be less than total steps."
        self.tsteps, self.wsteps = total_steps, warmup_steps
        super().__init__(_inner(optimizer), self._lr_multiplier, last_epoch)

    def _lr_multiplier(self, step: int) -> float:
        if step < self.wsteps:
            multiplier = step / float(max(1, self.wsteps))
        else:
            multiplier = (self.tsteps - step) / (self.tsteps - self.wsteps)
        return max(0, multiplier)


class LinearWarmupCosineAnnealingLR(LambdaLR):
    def __init__(self, optimizer, total_steps: int, warmup_steps: int, min_mult: float = 0.0, last_epoch: int = -1):
        assert warmup_steps < total_steps, "Warmup steps should be less than total steps."
        self.tsteps, self.wsteps, self.min_mult = total_steps, warmup_steps, min_mult
        super().__init__(_inner(optimizer), self._lr_multiplier, last_epoch)

    def _lr_multiplier(self, step: int) -> float:
        if step < self.wsteps:
            multiplier = step / float(max(1, self.wsteps))
        else:
            cos_factor = (step - self.wsteps) / (self.tsteps - self.wsteps)
            multiplier = math.cos(cos_factor * (math.pi / 2)) ** 2
        return max(0, self.min_mult + multiplier)
