"""Update path on the HIP kernels: global-norm clipping + SGD(momentum, weight decay, one group per tensor) + Lookahead
in one pass over the flat parameter arena (csrc/optim_ops.hip).

Mirrors the reference objects: torch.optim.SGD built with one param group per parameter (reference factories.py:464-482)
wrapped in optim/lookahead.py:35-101's Lookahead(k, alpha). `state_dict()` keeps torch.optim.SGD's layout
(`momentum_buffer` per parameter), and — like the reference (lookahead.py:68-78) — the slow weights are not serialized but
re-seeded from the fast weights on load.
"""
import ctypes as C
from typing import Any, Callable, Dict

import torch

from .. import hip

CHUNK = 8192     # elements per workgroup of the update kernel


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False))
        plist = [p for g in self.param_groups for p in g["params"]]
        arenas = {id(p._clite[0]) for p in plist if hasattr(p, "_clite")}
        if len(arenas) != 1 or any(not hasattr(p, "_clite") for p in plist):
            raise RuntimeError("FusedSGD: parameters must live in one device arena — move the model to the GPU (model.to(device)) "
                               "before building the optimizer, as reference train.py:136-138 does")
        self.arena = plist[0]._clite[0]
        dev = self.arena.device
        self.flat_v = torch.zeros(self.arena.total, device=dev, dtype=torch.float32)
        self.flat_slow = self.arena.flat_p.clone()
        self.hp = torch.zeros(8, device=dev, dtype=torch.float32)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.sumsq_partials = torch.zeros(1024, device=dev, dtype=torch.float32)      # clite_sumsq's fixed-order first stage
        self.max_norm = 0.0
        self.grad_prescale = 1.0
        self._items = None
        self._hp_ring = None
        self._items_key = None
        self.before_step: Callable = None     # hook: e.g. wait for the gradient all-reduce stream

    # -- work-item table (one entry per <= CHUNK slice of one tensor) ------------------------------------------------
    def _base_lrs(self):
        return tuple((g.get("initial_lr", g["lr"]), g["weight_decay"], bool(g["params"][0].requires_grad)) for g in self.param_groups)

    def _build_items(self):
        key = self._base_lrs()
        if key == self._items_key:
            return
        rows = []
        for g, (lr, wd, on) in zip(self.param_groups, key):
            if not on:
                continue
            for p in g["params"]:
                o, n = self.arena.index[p._clite[1]]
                n4 = (n + 3) // 4 * 4
                for s in range(0, n4, CHUNK):
                    rows.append((o + s, min(CHUNK, n4 - s), lr, wd))
        # frozen tensors: no work item, so the update kernel never zeroes their slice of flat_g — yet several backward kernels write
        # gradients unconditionally (BatchNorm dgamma/dbeta, LayerNorm, embeddings). Their (merged) spans are zeroed explicitly.
        spans = []
        for g, (_, _, on) in zip(self.param_groups, key):
            if on:
                continue
            for p in g["params"]:
                o, n = self.arena.index[p._clite[1]]
                if spans and spans[-1][1] == o:
                    spans[-1][1] = o + n
                else:
                    spans.append([o, o + n])
        self._frozen_spans = [tuple(x) for x in spans]
        rows.sort()                      # by arena offset: a [lo, hi) element range of the arena is then a contiguous run of items (launch(span=...))
        arr = (hip.OptimItem * len(rows))()
        for i, (s, c, lr, wd) in enumerate(rows):
            arr[i].start, arr[i].count, arr[i].lr, arr[i].wd = s, c, lr, wd
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._items = host.to(self.arena.device)
        self._n_items = len(rows)
        self._item_starts = [r[0] for r in rows]
        self._items_key = key

    def _lr_mult(self):
        for g in self.param_groups:
            base = g.get("initial_lr", None)
            if base:
                return g["lr"] / base
        return 1.0

    # -- reference train.py:222-223 ---------------------------------------------------------------------------------
    def clip_grad_norm(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ over every parameter: the squared norm is reduced now, the scaling is applied
        inside the update kernel. Returns the (device) total norm of the pre-scaled gradients."""
        self.max_norm = float(max_norm)
        self.arena.join()
        self._build_items()
        self.zero_frozen()           # stale / unconditional gradients of frozen tensors must not enter the norm
        self.sumsq.zero_()
        self.sumsq_spans()
        return self.sumsq.sqrt() * self.grad_prescale

    def norm_spans(self):
        """The spans the squared gradient norm is summed over, in the FIXED order every path - eager, one graph, per-phase graphs - adds them into
        `sumsq` (so that all of them derive bit-identical clip factors): what is final before the image encoder's last weight-gradient groups (the text
        encoder, then layer3 .. the loss heads), then the late span (stem, layer1, layer2). The captured per-phase step sums the early spans on the main
        stream while the side stream finishes the late span's gradients (train_loop.TrainStep: norm_early); one span when the model has no such layout."""
        if getattr(self, "_norm_spans", None) is None:
            A = self.arena
            spans = [(0, A.total)]
            try:
                img, l3 = A.region("image_encoder."), A.region("image_encoder.img_encoder.layer3.")
                if img[0] < l3[0] < img[1]:
                    spans = [sp for sp in ((0, img[0]), (l3[0], A.total), (img[0], l3[0])) if sp[1] > sp[0]]
            except AssertionError:
                pass
            self._norm_spans = spans
        return self._norm_spans

    def sumsq_spans(self, part=None):
        """sumsq += the squared norm of the spans of `part`: None all, "early" all but the last, "late" the last (a single-span layout is all "late")."""
        spans = self.norm_spans()
        sel = spans if part is None else (spans[:-1] if part == "early" else spans[-1:])
        for lo, hi in sel:
            hip.sumsq(self.arena.flat_g[lo:hi], hi - lo, self.sumsq, self.sumsq_partials)

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are zeroed by the update kernel itself; this only clears a backward that was not followed by step()."""
        self.arena.join()
        if getattr(self, "_dirty", True):
            self.arena.flat_g.zero_()
        self._dirty = True

    @torch.no_grad()
    def step(self, closure=None, lookahead_sync=False, alpha=1.0):
        if self.before_step is not None:
            self.before_step()
        self.upload_hp(lookahead_sync, alpha)
        self.launch()
        self.max_norm = 0.0
        return None

    # step() = upload_hp() + launch(). A captured hipGraph of the train step contains only launch() (and the sumsq of
    # clip_grad_norm); the host uploads this step's hyper-parameters into `hp` before every replay.
    def upload_hp(self, lookahead_sync=False, alpha=1.0, max_norm=None):
        """Host -> device: (lr multiplier of the schedule, momentum, clip norm, Lookahead sync flag, alpha, gradient pre-scale)."""
        mom = self.param_groups[0]["momentum"]
        mn = self.max_norm if max_norm is None else float(max_norm)
        vals = [self._lr_mult(), mom, mn, 1.0 if lookahead_sync else 0.0, alpha, self.grad_prescale, 0.0, 0.0]
        if not self.hp.is_cuda:
            self.hp.copy_(torch.tensor(vals))
            return
        # asynchronous upload from a small ring of pinned slots: a slot is rewritten only after the copy that read it has executed
        if self._hp_ring is None:
            self._hp_ring = [(torch.empty(8, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(8)]
            self._hp_used = [False] * 8
            self._hp_next = 0
        i = self._hp_next
        self._hp_next = (i + 1) % len(self._hp_ring)
        host, ev = self._hp_ring[i]
        if self._hp_used[i]:
            ev.synchronize()
        host.copy_(torch.tensor(vals))
        self.hp.copy_(host, non_blocking=True)
        ev.record()
        self._hp_used[i] = True

    def zero_frozen(self):
        for lo, hi in getattr(self, "_frozen_spans", ()):
            self.arena.flat_g[lo:hi].zero_()

    @torch.no_grad()
    def launch(self, span=None):
        """span = (lo, hi): update only the tensors whose arena offset lies in [lo, hi) — the captured step updates the image encoder at the end
        of a step and the rest at the start of the next one (train_loop.TrainStep, defer_update); every tensor exactly once per step."""
        self.arena.join()
        self._build_items()
        self.zero_frozen()
        first, n = 0, self._n_items
        if span is not None:
            import bisect
            first = bisect.bisect_left(self._item_starts, span[0])
            n = bisect.bisect_left(self._item_starts, span[1]) - first
        if n > 0:
            hip.sgd_step(self.arena.flat_p, self.arena.flat_g, self.flat_v, self.flat_slow, self.arena.flat_lp,
                         C.c_void_p(self._items.data_ptr() + first * C.sizeof(hip.OptimItem)), n, self.hp, self.sumsq)
        self.arena._tr_stale = True          # the bf16 weights changed: the transposed copies of the dgrad GEMMs are out of date
        self._dirty = False

    # -- torch.optim.SGD-compatible checkpoint layout -----------------------------------------------------------------
    def state_dict(self):
        self.arena.flush_pending()          # a deferred share of the last update (TrainStep defer_update) must land before momentum is read
        state, groups = {}, []
        for i, g in enumerate(self.param_groups):
            p = g["params"][0]
            o, n = self.arena.index[p._clite[1]]
            state[i] = {"momentum_buffer": self.arena._torch_view(self.flat_v, o, n, p).clone()}
            groups.append({k: v for k, v in g.items() if k != "params"} | {"params": [i]})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self.arena.flush_pending()
        for i, g in enumerate(self.param_groups):
            saved = sd["param_groups"][i]
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
            p = g["params"][0]
            o, n = self.arena.index[p._clite[1]]
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None and st.get("momentum_buffer") is not None:
                self.arena._torch_view(self.flat_v, o, n, p).copy_(st["momentum_buffer"])
        self._items_key = None


class FusedAdamW(FusedSGD):
    """torch.optim.AdamW (reference factories.py:439, `OPTIM.OPTIMIZER_NAME: adamw`, torch's default betas / eps) on the same arena, work items, clipping,
    Lookahead hook and captured-step protocol as FusedSGD (clite_adamw_step). `state_dict()` keeps torch.optim.AdamW's layout (step, exp_avg, exp_avg_sq
    per parameter). The step count - the bias corrections' exponent - advances once per upload_hp(), i.e. once per optimizer step however many span
    launches the captured step splits it into."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        torch.optim.Optimizer.__init__(self, params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        plist = [p for g in self.param_groups for p in g["params"]]
        arenas = {id(p._clite[0]) for p in plist if hasattr(p, "_clite")}
        if len(arenas) != 1 or any(not hasattr(p, "_clite") for p in plist):
            raise RuntimeError("FusedAdamW: parameters must live in one device arena - move the model to the GPU (model.to(device)) before building the "
                               "optimizer, as reference train.py:136-138 does")
        self.arena = plist[0]._clite[0]
        dev = self.arena.device
        self.flat_v = torch.zeros(self.arena.total, device=dev, dtype=torch.float32)           # exp_avg
        self.flat_v2 = torch.zeros(self.arena.total, device=dev, dtype=torch.float32)          # exp_avg_sq
        self.flat_slow = self.arena.flat_p.clone()
        self.hp = torch.zeros(12, device=dev, dtype=torch.float32)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.sumsq_partials = torch.zeros(1024, device=dev, dtype=torch.float32)
        self.max_norm = 0.0
        self.grad_prescale = 1.0
        self._items = None
        self._hp_ring = None
        self._items_key = None
        self.before_step = None
        self.steps = 0

    def upload_hp(self, lookahead_sync=False, alpha=1.0, max_norm=None):
        self.steps += 1
        b1, b2 = self.param_groups[0]["betas"]
        mn = self.max_norm if max_norm is None else float(max_norm)
        vals = [self._lr_mult(), b1, mn, 1.0 if lookahead_sync else 0.0, alpha, self.grad_prescale, b2, self.param_groups[0]["eps"],
                1.0 - b1 ** self.steps, 1.0 - b2 ** self.steps, 1.0 - b1, 1.0 - b2]
        if not self.hp.is_cuda:
            self.hp.copy_(torch.tensor(vals))
            return
        if self._hp_ring is None:
            self._hp_ring = [(torch.empty(12, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(8)]
            self._hp_used = [False] * 8
            self._hp_next = 0
        i = self._hp_next
        self._hp_next = (i + 1) % len(self._hp_ring)
        host, ev = self._hp_ring[i]
        if self._hp_used[i]:
            ev.synchronize()
        host.copy_(torch.tensor(vals))
        self.hp.copy_(host, non_blocking=True)
        ev.record()
        self._hp_used[i] = True

    @torch.no_grad()
    def launch(self, span=None):
        self.arena.join()
        self._build_items()
        self.zero_frozen()
        first, n = 0, self._n_items
        if span is not None:
            import bisect
            first = bisect.bisect_left(self._item_starts, span[0])
            n = bisect.bisect_left(self._item_starts, span[1]) - first
        if n > 0:
            hip.adamw_step(self.arena.flat_p, self.arena.flat_g, self.flat_v, self.flat_v2, self.flat_slow, self.arena.flat_lp,
                           C.c_void_p(self._items.data_ptr() + first * C.sizeof(hip.OptimItem)), n, self.hp, self.sumsq)
        self.arena._tr_stale = True
        self._dirty = False

    def state_dict(self):
        self.arena.flush_pending()
        state, groups = {}, []
        for i, g in enumerate(self.param_groups):
            p = g["params"][0]
            o, n = self.arena.index[p._clite[1]]
            state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": self.arena._torch_view(self.flat_v, o, n, p).clone(),
                        "exp_avg_sq": self.arena._torch_view(self.flat_v2, o, n, p).clone()}
            groups.append({k: v for k, v in g.items() if k != "params"} | {"params": [i]})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self.arena.flush_pending()
        for i, g in enumerate(self.param_groups):
            saved = sd["param_groups"][i]
            for k, v in saved.items():
                if k != "params":
                    g[k] = tuple(v) if k == "betas" else v
            p = g["params"][0]
            o, n = self.arena.index[p._clite[1]]
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None:
                if st.get("exp_avg") is not None:
                    self.arena._torch_view(self.flat_v, o, n, p).copy_(st["exp_avg"])
                if st.get("exp_avg_sq") is not None:
                    self.arena._torch_view(self.flat_v2, o, n, p).copy_(st["exp_avg_sq"])
                if st.get("step") is not None:
                    self.steps = int(float(st["step"]))
        self._items_key = None


class Lookahead(object):
    r"""Reference optim/lookahead.py:8-127 on top of FusedSGD: every ``k``-th step ``p = alpha*p + (1-alpha)*slow; slow = p`` —
    executed inside the same update kernel."""

    def __init__(self, optimizer: FusedSGD, k: int = 5, alpha: float = 0.8):
        self.optimizer = optimizer
        self.k = k
        self.alpha = alpha
        self._k_counter = 0
        self.optimizer.flat_slow.copy_(self.optimizer.arena.flat_p)     # slow weights start as a copy of the parameters

    def __getstate__(self):
        return {"optimizer": self.optimizer, "alpha": self.alpha, "k": self.k, "_k_counter": self._k_counter}

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    @property
    def state(self):
        return self.optimizer.state

    def clip_grad_norm(self, max_norm):
        return self.optimizer.clip_grad_norm(max_norm)

    def zero_grad(self):
        self.optimizer.zero_grad()

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, state_dict: Dict[str, Any]):
        self.optimizer.load_state_dict(state_dict)
        self.optimizer.flat_slow.copy_(self.optimizer.arena.flat_p)      # reference lookahead.py:73-78

    def advance(self) -> bool:
        """Count one fast step; True when this step ends with the slow-weight synchronisation (reference lookahead.py:88-101)."""
        self._k_counter += 1
        sync = self._k_counter >= self.k
        if sync:
            self._k_counter = 0
        return sync

    def step(self, closure: Callable = None):
        return self.optimizer.step(closure, lookahead_sync=self.advance(), alpha=self.alpha)

    def load_slow_weights(self):
        a = self.optimizer.arena
        a.flush_pending()          # a deferred share of the last update must land on the FAST weights, not on top of the slow ones
        self._backup = a.flat_p.clone()
        a.flat_p.copy_(self.optimizer.flat_slow)
        a.refresh_lowp()

    def restore_fast_weights(self):
        a = self.optimizer.arena
        a.flush_pending()
        a.flat_p.copy_(self._backup)
        del self._backup
        a.refresh_lowp()
