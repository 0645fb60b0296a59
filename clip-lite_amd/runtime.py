"""Device-side runtime of the train step: flat parameter/gradient arenas and per-step kernel state.

Layout in HBM (MI355X, 288 GB): every trainable tensor of the model lives in ONE flat f32 buffer `flat_p` (master
weights), with same-offset twins `flat_g` (gradients, written in place by the wgrad kernels' float atomics — no per-tensor
.grad allocation, no autograd accumulation pass) and, in bf16 mode, `flat_lp` (the bf16 copy the forward/backward kernels
read; refreshed by the fused update kernel). Conv weights are stored [K][R][S][C] (what the implicit-GEMM loaders gather)
and exposed to PyTorch as channels_last [K][C][R][S] views, so `state_dict()` keeps the reference's torchvision shapes
(reference encoder.py:84-94 key map). The gradient all-reduce (reference train.py:174-178, DDP) runs on slices of `flat_g`.
"""
import torch

from . import hip

ALIGN = 64  # elements; keeps every tensor 256-byte aligned and QKV groups exactly contiguous


def _kernel_shape(p):
    """Shape of the tensor as the kernels see it (conv weights KCRS -> KRSC)."""
    if p.dim() == 4:
        K, Cc, R, S = p.shape
        return (K, R, S, Cc)
    return tuple(p.shape)


class Arena:
    @staticmethod
    def layout(named_params, contiguous_groups=()):
        """(order, index, total) of the flat buffers for these (name, tensor) pairs: named_parameters() order with every contiguous group pulled
        together at its first member, every tensor padded to ALIGN elements. Needs shapes only (meta tensors do): the data-parallel pre-flight
        test lays out the real ResNet-50 + BERT arena this way without allocating it."""
        named = list(named_params)
        by_name = dict(named)
        order, placed = [], set()
        group_of = {}
        for g in contiguous_groups:
            g = [n for n in g if n in by_name]
            for n in g:
                group_of[n] = g
        for n, _ in named:
            if n in placed:
                continue
            for m in group_of.get(n, [n]):
                if m not in placed:
                    order.append(m)
                    placed.add(m)
        index, off = {}, 0
        for n in order:
            numel = by_name[n].numel()
            index[n] = (off, numel)
            off += (numel + ALIGN - 1) // ALIGN * ALIGN
        return order, index, off

    def __init__(self, named_params, device, lowp, contiguous_groups=()):
        named = list(named_params)
        by_name = dict(named)
        order, self.index, off = self.layout(named, contiguous_groups)
        self.device, self.lowp = torch.device(device), lowp
        self._pending = []
        # a train step that left part of its update for later (train_loop.TrainStep defer_update) registers its finish() here; every reader of the
        # parameters outside the step (state_dict, load_state_dict, refresh_lowp, an eval forward) calls flush_pending() first
        self.pending_update = None
        self.total = off
        self.flat_p = torch.zeros(off, device=device, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=device, dtype=torch.float32)
        self.flat_lp = torch.zeros(off, device=device, dtype=torch.bfloat16) if lowp else None
        self.flat_lpT, self._tr, self._tr_stale, self._tr_event, self._tr_shape = None, None, True, None, {}
        self.params = {}
        self.names = order
        for n in order:
            p = by_name[n]
            o, numel = self.index[n]
            ks = _kernel_shape(p)
            src = p.detach().to(device=device, dtype=torch.float32)
            if p.dim() == 4:
                src = src.permute(0, 2, 3, 1)
            self.flat_p[o:o + numel].view(ks).copy_(src)
            p.data = self._torch_view(self.flat_p, o, numel, p)
            p.grad = self._torch_view(self.flat_g, o, numel, p)
            p._clite = (self, n)
            self.params[n] = p
        self.refresh_lowp()

    @staticmethod
    def _torch_view(flat, o, numel, p):
        if p.dim() == 4:
            K, Cc, R, S = p.shape
            return flat[o:o + numel].view(K, R, S, Cc).permute(0, 3, 1, 2)
        return flat[o:o + numel].view(p.shape)

    # cross-stream ordering ---------------------------------------------------------------------------------------
    # The backward executors write parameter gradients straight into flat_g on whatever stream autograd runs them on (the two
    # encoders use different streams), and no AccumulateGrad node exists to make autograd join those streams at the end of
    # backward(). So each executor leaves an event here and every reader of the arena (gradient norm, update kernel, gradient
    # exchange) waits for the outstanding events on its own stream first.
    def note_stream_work(self):
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pending.append(ev)
            return ev
        return None

    def join(self):
        if self._pending:
            cur = torch.cuda.current_stream(self.device)
            for ev in self._pending:
                cur.wait_event(ev)
            self._pending = []

    # kernel-side views -------------------------------------------------------------------------------------------
    def w(self, p):
        """Weight as the kernels read it: compute-dtype copy, kernel layout, contiguous."""
        _, n = p._clite
        o, numel = self.index[n]
        src = self.flat_lp if self.lowp else self.flat_p
        return src[o:o + numel].view(_kernel_shape(p))

    def w32(self, p):
        _, n = p._clite
        o, numel = self.index[n]
        return self.flat_p[o:o + numel].view(_kernel_shape(p))

    def g(self, p):
        """f32 gradient buffer in kernel layout (wgrad kernels accumulate here)."""
        _, n = p._clite
        o, numel = self.index[n]
        return self.flat_g[o:o + numel].view(_kernel_shape(p))

    def span(self, params, grad=False, lowp=True):
        """One contiguous kernel-side view over adjacent tensors (e.g. BERT q/k/v weights)."""
        offs = [self.index[p._clite[1]] for p in params]
        for (o, n), (o2, _) in zip(offs[:-1], offs[1:]):
            assert o + n == o2, "tensors are not contiguous in the arena"
        o, end = offs[0][0], offs[-1][0] + offs[-1][1]
        src = self.flat_g if grad else (self.flat_lp if (self.lowp and lowp) else self.flat_p)
        return src[o:end]

    def region(self, prefix):
        """[lo, hi) element range of the flat buffers covered by the tensors whose name starts with `prefix` (they are adjacent:
        the arena follows named_parameters() order, which lists one top-level module after the other)."""
        offs = [self.index[n] for n in self.names if n.startswith(prefix)]
        if not offs:
            return (0, 0)
        lo = min(o for o, _ in offs)
        hi = max((o + n + ALIGN - 1) // ALIGN * ALIGN for o, n in offs)
        assert sum((n + ALIGN - 1) // ALIGN * ALIGN for _, n in offs) == hi - lo, f"tensors under {prefix!r} are not contiguous"
        return (lo, min(hi, self.total))

    def flush_pending(self):
        """Complete a deferred share of the last update (no-op unless a TrainStep(defer_update=True) is mid-way): before anything reads or
        overwrites parameters, momentum or the bf16 copies from outside the step."""
        if self.pending_update is not None:
            self.pending_update()

    def refresh_lowp(self):
        """Re-derive the bf16 copy from the f32 masters (after load_state_dict or any out-of-band weight edit)."""
        self.flush_pending()
        if self.lowp:
            hip.cast_bf16(self.flat_p, self.flat_lp, self.total)
        self._tr_stale = True

    # transposed bf16 copies for the input-gradient GEMMs (include/clite.h: clite_transpose_weights) ----------------------------------
    def register_transposed(self, weights, groups=()):
        """weights: parameters whose dgrad reads a transposed copy — Linear [N][K] -> [K][N], conv (kernel layout [K][R][S][C]) -> [C][R][S][K].
        groups: lists of adjacent 2-D weights with equal K that one GEMM reads as a single [sum N][K] matrix (BERT's q/k/v): transposed as one
        matrix, so the copy is the [K][sum N] operand of the fused input gradient. Same offsets as the bf16 arena."""
        if not self.lowp:
            return
        import ctypes as C
        items, tile, done = [], 0, set()

        def add(off, rows, cols, src_ld, dst_ld, batch, sb, db):
            nonlocal tile
            it = hip.TransposeItem(off, off, rows, cols, src_ld, dst_ld, batch, sb, db, tile)
            items.append(it)
            tile += batch * ((rows + 63) // 64) * ((cols + 63) // 64)

        for g in groups:
            ps = [p for p in g]
            offs = [self.index[p._clite[1]] for p in ps]
            if any(p.dim() != 2 or p.shape[1] != ps[0].shape[1] for p in ps) or any(o + n != o2 for (o, n), (o2, _) in zip(offs[:-1], offs[1:])):
                continue
            rows, cols = sum(p.shape[0] for p in ps), ps[0].shape[1]
            if rows % 8 or cols % 8:
                continue
            add(offs[0][0], rows, cols, cols, rows, 1, 0, 0)
            self._tr_shape[tuple(id(p) for p in ps)] = (offs[0][0], (cols, rows))
            done.update(id(p) for p in ps)
        for p in weights:
            if id(p) in done:
                continue
            o, numel = self.index[p._clite[1]]
            if p.dim() == 2 and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0:
                N, K = p.shape
                add(o, N, K, K, N, 1, 0, 0)
                self._tr_shape[id(p)] = (o, (K, N))
            elif p.dim() == 4 and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0:
                K, Cc, R, S = p.shape
                add(o, K, Cc, R * S * Cc, R * S * K, R * S, Cc, K)
                self._tr_shape[id(p)] = (o, (Cc, R, S, K))
        if not items:
            return
        arr = (hip.TransposeItem * len(items))(*items)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._tr = (host.to(self.device), len(items), tile)
        self.flat_lpT = torch.zeros(self.total, device=self.device, dtype=torch.bfloat16)
        self._tr_stale = True

    def ensure_transposed(self, force=False, capturing=False):
        """Bring the transposed copies up to date with the bf16 arena on the current stream (one grouped launch), or — when another stream has
        already done so since the weights last changed — order the current stream behind that launch. Inside a stream capture only `force`
        does anything: the captured step places the launch itself (train_loop.TrainStep) and orders its phases explicitly."""
        if self._tr is None:
            return
        if capturing and not force:
            return
        if self._tr_stale or force:
            dev, n, tiles = self._tr
            hip.transpose_weights(self.flat_lp, self.flat_lpT, dev, n, tiles)
            self._tr_stale = False
            if not capturing and self.device.type == "cuda":
                self._tr_event = torch.cuda.Event()
                self._tr_event.record(torch.cuda.current_stream(self.device))
        elif self._tr_event is not None:
            torch.cuda.current_stream(self.device).wait_event(self._tr_event)

    def has_wt(self, *ps):
        return self._tr is not None and ((id(ps[0]) if len(ps) == 1 else tuple(id(p) for p in ps)) in self._tr_shape)

    def wt(self, *ps):
        """Transposed bf16 copy: one weight -> [K][N] / [C][R][S][K]; several adjacent ones (a registered group) -> [K][sum N]."""
        o, shape = self._tr_shape[id(ps[0]) if len(ps) == 1 else tuple(id(p) for p in ps)]
        n = 1
        for d in shape:
            n *= d
        return self.flat_lpT[o:o + n].view(shape)


class StepState:
    """Per-forward kernel state: dropout seed of this step and a bump allocator of dropout sites. With `indirect` the seed
    is the device address of a uint64 (hip.SEED_INDIRECT is OR-ed into every site), which is how a captured hipGraph of the
    step sees a fresh seed at every replay."""

    def __init__(self, seed, training, indirect=False):
        self.seed, self.training = seed, training
        self._site = 0
        self._flag = hip.SEED_INDIRECT if indirect else 0

    def site(self):
        self._site += 1
        return self._site | self._flag


class ZeroPool:
    """Bump allocator over pre-zeroed f32 chunks: the many small accumulators of a step (BatchNorm statistics, loss sums) come
    out of a few large memsets instead of one tiny fill kernel each. A chunk is handed out once and never reused."""

    def __init__(self, device, chunk=4 * 1024 * 1024):
        self.device, self.chunk = device, chunk
        self.buf, self.pos = None, 0

    def take(self, n):
        n = (n + 63) // 64 * 64
        if self.buf is None or self.pos + n > self.buf.numel():
            self.buf = torch.zeros(max(self.chunk, n), device=self.device, dtype=torch.float32)
            self.pos = 0
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out


STAT_REPLICAS = 8


class DeviceRuntime:
    """Everything a kernel-driving executor needs: device, compute dtype, the parameter arena, dropout seeding and the
    BatchNorm `num_batches_tracked` counters (kept in one int64 tensor per owner so a step bumps them with one add)."""

    def __init__(self, model, device, lowp, contiguous_groups=(), seed=0):
        self.device = torch.device(device)
        self.lowp = bool(lowp)
        self.dt = hip.BF16 if lowp else hip.F32
        self.tdtype = torch.bfloat16 if lowp else torch.float32
        # exact-f32 parity mode: BatchNorm variance by a second, centered pass (PyTorch-grade accuracy); the bf16 production mode
        # keeps the single-pass statistics that come for free out of the conv epilogue
        self.precise_bn = not lowp
        self.arena = Arena(model.named_parameters(), self.device, lowp, contiguous_groups)
        # input gradients as forward-form GEMMs on transposed weight copies (bf16 mode; Arena.register_transposed): every conv weight with
        # >= 8 input channels and every Linear weight; BERT's adjacent q/k/v weights as one matrix
        self.transposed_dgrad = bool(lowp)
        if lowp:
            by_name = dict(model.named_parameters())
            # only the two encoders' backward executors read the copies (bert._dgrad, resnet._wd); the heads' small GEMMs keep clite_gemm_nn on
            # the [out][in] weights, so their matrices (and the word embeddings, which have no in_features) are not re-transposed every step
            tw = [m.weight for top in ("image_encoder", "text_encoder") if hasattr(model, top) for m in getattr(model, top).modules()
                  if isinstance(getattr(m, "weight", None), torch.nn.Parameter) and hasattr(m.weight, "_clite")
                  and (hasattr(m, "in_features") or hasattr(m, "in_channels"))]
            tg = [[by_name[n] for n in g] for g in contiguous_groups if all(n in by_name for n in g) and by_name[g[0]].dim() == 2]
            self.arena.register_transposed(tw, tg)
        self.base_seed = int(seed)
        self._zpools = {}          # one pre-zeroed pool per launch stream (the two encoders run on different streams)
        self.side_stream = None    # the text encoder's stream (model.py), created on first use
        self.exchange = None
        self._spans = {}
        import os
        self.s2_classes = True             # 3x3/stride-2 dgrads as four parity-class GEMMs (attribute: tools flip it for A/B runs)
        self.fuse_bn_backward = True       # BatchNorm-backward reductions inside the dgrad epilogues (resnet.py)
        self.defer_head_wgrads = True      # captured step: the loss heads' Linear weight gradients ride in the encoders' grouped launches (train_loop.py)
        self.compact_shortcut = True       # stride-2 shortcuts' input gradients kept compact, added by the main branch's BatchNorm-backward dgrad (resnet.py; bf16)
        self.fused_heads = True            # the MI projection blocks' non-GEMM work in four kernels: 5 + 5 launches per block instead of 10 + 9 (loss.mi_block_forward / _backward; csrc/heads_fused.hip; bf16, <= 128 rows)
        self.bn_fold = True                # block-output BatchNorm backward folded into conv3's input + weight gradients: no apply pass, no dy (resnet.py _fold_wanted; bf16)
        self.bn_fold_min_rows = 300000     # ... where the tensors have at least this many rows (56 x 56 at batch 128: 401 408). Same-box A/B of the captured step: off 14.73 ms,
                                           # layer1 14.54, layers 1 + 2 14.75, + layer3 15.5 — below 56 x 56 the doubled K of the input gradient and the Gram
                                           # members of the side stream's groups cost what the apply pass saves
        self.stat_replicas_fixed = False   # True: 8 statistics replicas everywhere (new_stats; tools/ab_runtime.py stat_replicas_fixed=1 for the A/B)
        self.fp8 = False                   # image-encoder forward convs on OCP e4m3 operands, quantised by their producers (BASELINE configs[4]; fp8.py); bf16 mode only
        self.fp8_wgrad = True              # (with fp8 + fp8_dgrad) weight gradients of the convs whose two operands already exist in fp8, on the block-scaled MFMA (resnet.py wgrad)
        self.fp8_text = False              # ... and BERT's QKV / FFN1 / FFN2 forward linears, quantised by the LayerNorm forward and FFN1's epilogue (fp8.Fp8Text)
        self.zero_chunk = 512 * 1024       # floats per chunk of the zero pool (ZeroPool): every captured phase starts a fresh pool, and most need a few KB of statistics -
                                           # a 16 MB chunk per phase cost ~14 fills of 10 - 20 us per step
        self.stem_tail_deferred = True     # bn1's backward + conv1's weight gradient with the collected weight gradients (off the dependent chain) when the caller defers them
        self.stem_wgrad_patch = True       # conv1's weight gradient on the patch-resident kernel (hip.stem_wgrad_patch; bf16 mode)
        self.stem_pooled_stats = True      # bn1's backward reductions from pooled-size operands in the epilogue of layer1's first input gradient (resnet.py; bf16 mode)
        self.fp8_dgrad = False             # ... and the input gradients of the image encoder's 3 x 3 (>= 128 channels) / late 1 x 1 convs inside a block: e5m2 gradient x e4m3
                                           # transposed weights (clite_conv_dgrad_fp8), the gradient quantised by the producing bn_bwd_apply (needs fp8 = True)
        self.fp8_nets = {}                 # id(ResNet) -> fp8.Fp8Forward
        self.group_wgrad = bool(lowp)      # weight gradients of a backward pass as grouped launches (hip.WgradGroup / clite_wgrad_group)
        if lowp and self.device.type == "cuda":
            hip.patch_workspace(self.device)          # scratch of the patch-resident 3 x 3 weight gradient: must exist before any stream capture
        self._aux_streams, self._aux_busy, self._aux_keep, self.overlap_wgrad = {}, set(), {}, False
        self.steps = 0
        self.seed_dev, self._capturing, self._slots, self.graph_slots, self._graph_next = None, False, 0, 0, -1
        # flatten num_batches_tracked buffers per top-level owner
        self.counters = {}
        owners = {}
        for name, mod in model.named_modules():
            if "num_batches_tracked" in getattr(mod, "_buffers", {}):
                owners.setdefault(name.split(".")[0], []).append(mod)
        for owner, mods in owners.items():
            flat = torch.zeros(len(mods), dtype=torch.long, device=self.device)
            for i, m in enumerate(mods):
                flat[i] = m._buffers["num_batches_tracked"].to(self.device)
                m._buffers["num_batches_tracked"] = flat[i]
            self.counters[owner] = flat
        for mod in model.modules():
            mod._clite_rt = self

    @property
    def zpool(self):
        key = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        pool = self._zpools.get(key)
        if pool is None:
            pool = self._zpools[key] = ZeroPool(self.device, self.zero_chunk)
        return pool

    def gemm_ws(self, M, N):
        """Zeroed f32 [M][N] split-K workspace for a GEMM of few output tiles (clite_epilogue.splitk_ws), or None where the library would
        not split anyway (many tiles, exact-f32 mode)."""
        if not self.lowp or ((M + 127) // 128) * ((N + 127) // 128) > 48 or N % 8:
            return None
        return self.zpool.take(M * N)

    def new_stats(self, Cc, rows=None):
        """Zeroed replicated accumulator for per-channel statistics of a [rows][Cc] tensor. The replicas exist to spread the producers' float
        atomics (one per workgroup and column: same-address atomics retire at ~90 per microsecond), and every READER pays for them: each
        workgroup of bn_apply / bn_bwd_apply / the BatchNorm-backward dgrad sums all replicas of its columns in its prologue — thousands of
        workgroups reading the same few lines: 3.5 us per launch at 8 replicas (bn_apply on 25088 x 256: 10.9 us, 7.6 with one), which is most
        of what the ~150 small BatchNorm launches of a step cost beyond their bytes. So the count follows the number of producing workgroups.
        Same-box A/B on the captured step (8 everywhere / 1-2-4-8 by rows / this rule / 1 everywhere): 16.29 / 15.87 / 15.63 / 15.73 ms."""
        R = STAT_REPLICAS
        # bf16 mode only: the exact-f32 mode's second (centered) variance pass adds its deviations from up to 1024 workgroups per column — there 8
        # replicas are faster (same-box A/B, bench --f32: 67.0 ms with 8, 70.7 with this rule); the deterministic mode deals one reduction workgroup
        # per replica (det.h)
        if rows is not None and self.lowp and not self.stat_replicas_fixed and not hip.is_deterministic():
            R = 1 if rows <= 65536 else 2 if rows <= 262144 else 4
        return hip.Stats(self.zpool.take(R * 3 * Cc), R, Cc)

    # -- weight gradients off the critical path ---------------------------------------------------------------------------
    # In backward only dgrad -> BN backward -> dgrad ... is a dependency chain; every weight-gradient GEMM (conv wgrad, linear
    # dW/db) just needs dy and the saved input and feeds nothing but the gradient arena. They are enqueued on an auxiliary stream
    # so they can fill the CUs that the chain's small-grid / latency-bound kernels leave idle (in a captured step: parallel
    # branches of the hipGraph). `join_aux` orders the caller's stream after them. OFF by default (`overlap_wgrad`): measured on
    # MI355X at batch 128 with the text encoder already running beside the ResNet, the extra branch slows the step (25.1 ->
    # 28.6 ms): the split-K wgrad grids contend with the dgrad/BN chain instead of filling gaps.
    def aux_launch(self, fn, *tensors):
        if self.device.type != "cuda" or not self.overlap_wgrad:
            fn()
            return
        cur = torch.cuda.current_stream(self.device)
        if self.side_stream is not None and cur == self.side_stream:
            # the text encoder's stream is itself a forked branch; forking a second level from it makes hipStreamEndCapture
            # crash (ROCm 7.2), and BERT's backward is not the critical path of the step anyway: keep its dW GEMMs in line
            fn()
            return
        st = self._aux_for(cur)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            fn()
        # transient operands (dy) must not be recycled by the allocator before the aux stream has read them: keep them referenced
        # until join_aux has ordered the launching stream after the aux stream
        self._aux_keep.setdefault(cur.cuda_stream, []).extend(t for t in tensors if t is not None)
        self._aux_busy.add(cur.cuda_stream)

    def _aux_for(self, cur):
        """The auxiliary stream (created outside any capture: begin_capture calls this first)."""
        st = self._aux_streams.get("main")
        if st is None:
            st = self._aux_streams["main"] = torch.cuda.Stream(device=self.device)
        return st

    def join_aux(self):
        if self.device.type != "cuda":
            return
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream in self._aux_busy:
            cur.wait_stream(self._aux_for(cur))
            self._aux_busy.discard(cur.cuda_stream)
            self._aux_keep.pop(cur.cuda_stream, None)

    def grads_ready(self, module):
        """Tell the gradient exchange (data parallel) that every parameter gradient under `module` is final for this step,
        so its region of the flat gradient arena can be all-reduced while the rest of backward still runs."""
        ex = getattr(self, "exchange", None)
        if ex is None:
            return
        self.join_aux()          # the region's weight gradients were written on the auxiliary stream
        span = self._spans.get(id(module))
        if span is None:
            offs = [self.arena.index[p._clite[1]] for p in module.parameters() if hasattr(p, "_clite")]
            span = (min(o for o, _ in offs), max((o + n + ALIGN - 1) // ALIGN * ALIGN for o, n in offs)) if offs else (0, 0)
            self._spans[id(module)] = span
        ex.region_ready(*span)

    def bump_counters(self, owner, n=1):
        if owner in self.counters:
            self.counters[owner] += n

    # -- dropout / prior-noise seeds ---------------------------------------------------------------------------------------
    # Eager: the k-th forward executor of the process gets seed base*1000003 + k by value. Captured (hipGraph) steps read the
    # same sequence from `seed_dev`: slot j of the captured step holds the seed of its j-th executor, and the graph's last node
    # advances every slot by the number of slots, so eager and replayed steps draw identical masks.
    SEED_SLOTS = 16

    def next_step(self, training):
        if self._capturing:
            j = self._slots
            self._slots += 1
            assert j < self.SEED_SLOTS
            return StepState(self.seed_dev.data_ptr() + 8 * j, training, indirect=True)
        self.steps += 1
        return StepState(self.base_seed * 1000003 + self.steps, training)

    def new_side_stream(self):
        """The text encoder's stream. Measured and rejected (round 2, MI355X, 18.5 ms step): a high-priority side stream, a high-priority main
        stream (18.5 / 18.6 ms: HIP stream priorities do not change which stream's workgroups get the CUs), and a CU-masked side stream
        (hipExtStreamCreateWithCUMask with 64-160 CUs wrapped as an ExternalStream: 21.6 ms for every mask size — graph replays on it no
        longer overlapped the main stream at all)."""
        return torch.cuda.Stream(device=self.device)

    def begin_capture(self):
        if self.seed_dev is None:
            self.seed_dev = torch.zeros(self.SEED_SLOTS, device=self.device, dtype=torch.int64)
        if self.side_stream is None:
            self.side_stream = self.new_side_stream()
        self._aux_for(torch.cuda.current_stream(self.device))
        self._capturing, self._slots = True, 0
        self._saved_zpools, self._zpools = self._zpools, {}     # chunks taken during capture live in the graph's memory pool

    def end_capture(self):
        """Call inside the capture, after the last kernel of the step: advances the device seeds for the next replay."""
        self.seed_dev.add_(self._slots)
        self._capturing = False
        self.graph_slots = self._slots
        self._zpools = self._saved_zpools

    def abort_capture(self):
        self._capturing = False
        self._zpools = self._saved_zpools

    def sync_graph_seeds(self):
        """Before a replay: make the device seeds continue the host sequence (no-op while only replays advance it)."""
        if self._graph_next != self.steps:
            vals = [self.base_seed * 1000003 + self.steps + 1 + j for j in range(self.SEED_SLOTS)]
            self.seed_dev.copy_(torch.tensor(vals, dtype=torch.int64))
        self.steps += self.graph_slots
        self._graph_next = self.steps
