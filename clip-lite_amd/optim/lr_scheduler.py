"""Learning-rate schedules behind ``OPTIM.LR_DECAY_NAME`` (reference optim/lr_scheduler.py:19-202; selected by
factories.py:505-531): a linear warm-up from 0 followed by one of four decay shapes.

The schedule is ONE scalar per step — every parameter group's rate is its base rate times the same multiplier — and the fused update
kernel takes exactly that scalar (``hp[0]`` of ``clite_sgd_step``, uploaded by ``FusedSGD.upload_hp``). So instead of four LambdaLR
subclasses there is one pure function, :func:`warmup_multiplier`, a table of decay shapes, and a small stateful driver that keeps the
reference's public surface (class names, constructor arguments, ``step()``, ``get_last_lr()``, ``_lr_multiplier``,
``state_dict()``/``load_state_dict()`` with torch's ``last_epoch`` / ``base_lrs`` keys so reference checkpoints resume).
"""
import math
from typing import Callable, Dict, List, Sequence


def _decay_none(progress: float, step: int, shape: dict) -> float:
    return 1.0


def _decay_multistep(progress: float, step: int, shape: dict) -> float:
    passed = sum(1 for m in shape["milestones"] if m <= step)       # milestones already reached at `step`
    return shape["gamma"] ** passed


def _decay_linear(progress: float, step: int, shape: dict) -> float:
    return 1.0 - progress


def _decay_cosine(progress: float, step: int, shape: dict) -> float:
    return math.cos(0.5 * math.pi * progress) ** 2 + shape.get("min_mult", 0.0)


# decay shape after the warm-up, as a function of progress = (step - warmup) / (total - warmup) in [0, 1)
DECAY_SHAPES: Dict[str, Callable[[float, int, dict], float]] = {
    "none": _decay_none, "multistep": _decay_multistep, "linear": _decay_linear, "cosine": _decay_cosine,
}


def warmup_multiplier(kind: str, step: int, total: int, warmup: int, **shape) -> float:
    """Multiplier on every base learning rate at optimisation step `step` (step 0 => 0: the reference's first update is a no-op).
    During warm-up `step / warmup` (plus ``min_mult`` for the cosine shape, whose floor the reference adds in both phases,
    optim/lr_scheduler.py:193-202); afterwards ``DECAY_SHAPES[kind]``. Never negative."""
    if step < warmup:
        value = step / float(max(1, warmup)) + (shape.get("min_mult", 0.0) if kind == "cosine" else 0.0)
    else:
        value = DECAY_SHAPES[kind]((step - warmup) / float(total - warmup), step, shape)
    return max(0.0, value)


class WarmupSchedule:
    """Stateful driver: counts steps and writes ``base_lr * multiplier`` into the optimizer's param groups (where FusedSGD reads the
    multiplier back as ``lr / initial_lr`` when it uploads the update kernel's hyper-parameters). Works on any object exposing
    ``param_groups`` — a torch optimizer or the Lookahead wrapper."""
    kind: str = ""

    def __init__(self, optimizer, total_steps: int, warmup_steps: int, last_epoch: int = -1, **shape):
        if not warmup_steps < total_steps:
            raise AssertionError(f"{type(self).__name__}: warmup_steps ({warmup_steps}) must be below total_steps ({total_steps})")
        self.optimizer = optimizer
        self.total_steps, self.warmup_steps, self.shape = total_steps, warmup_steps, shape
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lrs: List[float] = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self._step_count = 0
        self.step()               # like torch's schedulers: construction applies step `last_epoch + 1`

    def _lr_multiplier(self, step: int) -> float:
        return warmup_multiplier(self.kind, step, self.total_steps, self.warmup_steps, **self.shape)

    def _apply(self):
        mult = self._lr_multiplier(self.last_epoch)
        self._last_lr = [b * mult for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g["lr"] = lr

    def step(self):
        self.last_epoch += 1
        self._step_count += 1
        self._apply()

    def get_last_lr(self) -> List[float]:
        return list(self._last_lr)

    def state_dict(self) -> dict:
        """torch's LambdaLR layout plus this class's own keys, so that the reference's scheduler (a LambdaLR subclass: optim/lr_scheduler.py:11-202)
        can resume a checkpoint written here: LambdaLR.load_state_dict pops "lr_lambdas" (one entry per lambda; None = "not a picklable object,
        keep the constructed one") and copies every other key onto the instance — hence the reference's attribute names tsteps / wsteps
        (and min_mult / milestones / gamma where the schedule has them) beside ours."""
        st = {"last_epoch": self.last_epoch, "_step_count": self._step_count, "base_lrs": list(self.base_lrs), "_last_lr": list(self._last_lr),
              "kind": self.kind, "total_steps": self.total_steps, "warmup_steps": self.warmup_steps, "shape": dict(self.shape),
              "lr_lambdas": [None], "tsteps": self.total_steps, "wsteps": self.warmup_steps}
        st.update({k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in self.shape.items()})
        return st

    def load_state_dict(self, state: dict):
        """Accepts this class's own dict and a torch LambdaLR one (a checkpoint written by the reference): the position is `last_epoch`."""
        self.last_epoch = int(state["last_epoch"])
        self._step_count = int(state.get("_step_count", self.last_epoch + 1))
        if "base_lrs" in state and len(state["base_lrs"]) == len(self.base_lrs):
            self.base_lrs = list(state["base_lrs"])
        self._apply()


def _check_milestones(milestones: Sequence[int], warmup_steps: int, total_steps: int):
    ms = list(milestones)
    if not ms or ms != sorted(ms):
        raise AssertionError("milestones must be a non-empty increasing list")
    if not (warmup_steps < ms[0] and ms[-1] < total_steps):
        raise AssertionError("milestones must lie strictly between warmup_steps and total_steps")
    return ms


class LinearWarmupNoDecayLR(WarmupSchedule):
    kind = "none"

    def __init__(self, optimizer, total_steps: int, warmup_steps: int, last_epoch: int = -1):
        super().__init__(optimizer, total_steps, warmup_steps, last_epoch)


class LinearWarmupMultiStepLR(WarmupSchedule):
    kind = "multistep"

    def __init__(self, optimizer, total_steps: int, warmup_steps: int, milestones: List[int], gamma: float = 0.1, last_epoch: int = -1):
        super().__init__(optimizer, total_steps, warmup_steps, last_epoch,
                         milestones=_check_milestones(milestones, warmup_steps, total_steps), gamma=gamma)


class LinearWarmupLinearDecayLR(WarmupSchedule):
    kind = "linear"

    def __init__(self, optimizer, total_steps: int, warmup_steps: int, last_epoch: int = -1):
        super().__init__(optimizer, total_steps, warmup_steps, last_epoch)


class LinearWarmupCosineAnnealingLR(WarmupSchedule):
    kind = "cosine"

    def __init__(self, optimizer, total_steps: int, warmup_steps: int, min_mult: float = 0.0, last_epoch: int = -1):
        super().__init__(optimizer, total_steps, warmup_steps, last_epoch, min_mult=min_mult)
