"""Small runtime helpers used by the train loop (reference utils/base.py:80-143 Timer; make_directory)."""
import os
import time
from typing import Optional


def make_directory(path: str) -> None:
    os.makedirs(path, exist_ok=True)


class Timer(object):
    """Wall-clock per iteration and an ETA from a moving window (same fields and strings as the reference's Timer)."""

    def __init__(self, start_from: int = 1, total_iterations: Optional[int] = None, window_size: int = 20):
        self.current_iter = start_from - 1
        self.total_iters = total_iterations
        self._start_time = time.time()
        self._times = [0.0] * window_size

    def tic(self) -> None:
        self._start_time = time.time()

    def toc(self) -> None:
        self._times = self._times[1:] + [time.time() - self._start_time]
        self.current_iter += 1

    @property
    def stats(self) -> str:
        return f"Iter {self.current_iter} | Time: {self._times[-1]:.3f} sec | ETA: {self.eta_hhmm}"

    @property
    def eta_sec(self) -> float:
        if not self.total_iters:
            return 0.0
        return sum(self._times) / len(self._times) * (self.total_iters - self.current_iter)

    @property
    def eta_hhmm(self) -> str:
        if not self.total_iters:
            return "N/A"
        s = int(self.eta_sec)
        return f"{s // 3600}h {((s % 3600) // 60):02d}m"
