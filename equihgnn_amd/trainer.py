"""Training-step harness equal to the reference's LitModel step (main.py:49-63,137-151):
forward -> nn.MSELoss -> backward -> gradient average across ranks -> Adam(lr, wd).

Data parallelism: one process per GPU, each rank steps its own molecule batch;
the only exchange is ONE all-reduce of a flat gradient buffer per step (RCCL over xGMI when the
backend is "nccl", gloo on CPU for the tests).  Parameter gradients are views into that flat
buffer, so there is no pack/unpack copy, and parameters whose gradient the model never
produces (the reference's dead branches, which is why main.py:281 needs
``find_unused_parameters``) are simply left out: their ``.grad`` stays ``None`` on every rank.
"""
from __future__ import annotations

import inspect
from typing import Callable, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class FlatGradients:
    """One contiguous fp32 buffer holding every live parameter's gradient."""

    def __init__(self, params):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        w = _world()
        if w > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / w)


class TrainStep:
    """``loss = step(data)``; ``data.y`` is the target.  Model-agnostic (the gloo tests drive it
    with the CPU oracle, the GPU path with the HIP models)."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, weight_decay: float = 0.0,
                 broadcast_from_rank0: bool = True, on_batch: Optional[Callable] = None):
        self.model = model
        self.lr, self.wd = lr, weight_decay
        self.on_batch = on_batch
        self.flat: Optional[FlatGradients] = None
        self.opt: Optional[torch.optim.Optimizer] = None
        if broadcast_from_rank0 and _world() > 1:  # DDP broadcasts rank 0's parameters at wrap time
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0)

    def _first_step(self, data):
        """Discover which parameters receive a gradient, then lay out the flat buffer."""
        for p in self.model.parameters():
            p.grad = None
        loss = F.mse_loss(self.model(data), data.y)
        loss.backward()
        live = [p for p in self.model.parameters() if p.grad is not None]
        first = [p.grad.clone() for p in live]
        self.flat = FlatGradients(live)
        for p, g in zip(live, first):
            p.grad.copy_(g)
        fused = live[0].is_cuda
        self.opt = torch.optim.Adam(live, lr=self.lr, weight_decay=self.wd, fused=fused)
        return loss

    def step(self, data) -> torch.Tensor:
        if self.on_batch is not None:
            self.on_batch(data)
        if self.flat is None:
            loss = self._first_step(data)
        else:
            self.flat.zero_()
            loss = F.mse_loss(self.model(data), data.y)
            loss.backward()
        self.flat.all_reduce_mean()
        self.opt.step()
        return loss.detach()

    def sync_buffers(self):
        """DDP's per-forward buffer broadcast (BatchNorm running stats of ``mhnnm``)."""
        if _world() > 1:
            for b in self.model.buffers():
                dist.broadcast(b.data, src=0)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad=False) over ONE flat fp32 parameter through eqh_adam_step: a grid-stride
    kernel instead of the multi-tensor kernel's one block per 64 Ki elements (12 us against 46 us for the
    2.6 M parameters of egnn_equihnns).  Learning rate and step counter live in device memory, so a
    captured hipGraph follows a scheduler (``sync_lr`` before each replay); ``grad_scale`` folds the
    1 / world_size of the gradient average into the update."""

    def __init__(self, param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__([param], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0
        self.zero_grad_in_step = False      # clear the flat gradient inside the update kernel (GraphedTrainStep)
        self.zero_also = None               # callable -> a second buffer the update clears (the owner's MergedScratch)
        self._lr_dev = None
        st = self.state[param]
        st["exp_avg"] = torch.zeros_like(param.data)
        st["exp_avg_sq"] = torch.zeros_like(param.data)
        st["step_block"] = torch.zeros(2, dtype=torch.int64, device=param.device)   # {step, ticket}
        st["lr"] = torch.tensor([float(lr)], dtype=torch.float32, device=param.device)
        self._lr_dev = float(lr)

    def sync_lr(self):
        """Mirror param_groups[0]['lr'] (what schedulers write) into the device scalar; outside capture."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_dev:
            p = self.param_groups[0]["params"][0]
            self.state[p]["lr"].fill_(lr)
            self._lr_dev = lr

    @torch.no_grad()
    def step(self, closure=None):
        from . import hip, ops
        g = self.param_groups[0]
        p = g["params"][0]
        st = self.state[p]
        b1, b2 = g["betas"]
        zs = self.zero_also() if (self.zero_grad_in_step and self.zero_also is not None) else None
        hip.check(hip.lib().eqh_adam_step(ops._ptr(p.data), ops._ptr(p.grad), ops._ptr(st["exp_avg"]),
                                          ops._ptr(st["exp_avg_sq"]), p.numel(), ops._ptr(st["lr"]), b1, b2, g["eps"],
                                          g["weight_decay"], self.grad_scale, ops._ptr(st["step_block"]),
                                          1 if self.zero_grad_in_step else 0,
                                          ops._ptr(zs) if zs is not None else None, zs.numel() if zs is not None else 0,
                                          ops._stream(p.device)), "eqh_adam_step")


def with_next(batches):
    """(batch, next batch or None) pairs of an iterable: the one batch of look-ahead ``GraphedTrainStep.step(data, next_data)``
    uses to build the next batch's index beside the current step."""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


class CollectiveCaptureRefused(RuntimeError):
    """The process-group backend could not record a collective into a hipGraph (raised by GraphedTrainStep._capture_pass)."""


class GraphedTrainStep:
    """TrainStep with the launch-bound part captured in hipGraphs.

    forward + MSE + backward (≈700 kernel launches for egnn_equihnns, including the per-batch index
    build), the gradient all-reduce and the fused Adam update are captured ONCE per static shape bucket
    as ONE hipGraph and replayed: with the "nccl" backend (RCCL) the collective is a node of that graph
    (``collective_mode == "in_graph"``; RCCL kernels are stream-capturable), so a multi-rank step is one
    graph launch exactly like the single-rank one.  Where the collective cannot be captured (gloo, or
    a capture that RCCL refuses) the step falls back to graph A -> eager all-reduce -> graph B
    (``"split"``) -- on EVERY rank: the ranks settle the form with one eager all-reduce (MIN of a success flag) at their
    first capture, and only a failing collective call triggers the fallback (kernel / argument errors re-raise).  Batches must be padded to bucket extents with ``batch.pad_batch`` (exact for models
    without batch statistics); the loss is taken over the real molecules only.

    Parameters live in ONE flat buffer (``pflat``; every ``nn.Parameter`` is a view into it) and so
    do their gradients (``gflat``, same element order): the weight gradients of ``ops.linear`` are
    accumulated straight into its head by the GEMMs, the gradients autograd allocates inside the
    captured backward (``grad=None`` before the capture, so the first contribution is written, not
    added) are packed behind them by one batched copy at the end of the first graph.  The all-reduce
    and the fused Adam therefore each see a single tensor.
    """

    def __init__(self, model: nn.Module, lr: float = 1e-4, weight_decay: float = 0.0,
                 broadcast_from_rank0: bool = True, collective: bool = True, keep_grads: bool = False,
                 force_collective: bool = False, graph_collective: bool = True):
        """``collective=False``: a rank-local trainer inside a multi-rank job (no broadcast, no all-reduce) -- what a
        measurement on ONE rank needs while the other ranks wait at a barrier.  ``keep_grads``: leave the gradients of the
        last step readable in ``p.grad`` after ``step()`` (tests); by default the update kernel clears the flat gradient
        buffer as it consumes it -- optimizer.zero_grad() without the fill launch at the head of every replayed step.
        ``force_collective``: take the multi-rank code path (broadcasts, all-reduce, 1 / world scaling) even when the
        process group has ONE rank -- how the path is exercised on a one-GPU box.  ``graph_collective=False``: never
        capture the collective (always graph A -> eager all-reduce -> graph B)."""
        self.model, self.lr, self.wd = model, lr, weight_decay
        self.collective = collective
        self.keep_grads = keep_grads
        self.force_collective = bool(force_collective) and collective and dist.is_available() and dist.is_initialized()
        self.graph_collective = graph_collective
        self.collective_mode = "none"       # "none" | "in_graph" | "split": what the captured steps do (set by _capture)
        self.capture_error = None           # why an in-graph capture fell back to "split", if it did
        self.in_graph_backends = ("nccl",)  # backends whose collectives are stream-capturable (RCCL)
        # Index prefetch: the per-batch index (three CSR sorts, kNN + its transpose: ~100 us of a 1.1 ms egnn_equihnns step)
        # depends on the batch only, not on the parameters.  step(data, next_data) builds next_data's index on a side
        # stream (its own small hipGraph) WHILE the step graph of `data` runs, and the step graph then starts at the
        # embedding.  EQH_NO_INDEX_PREFETCH=1: the index stays at the head of the step graph (rounds 1-5).
        #
        # Whether that pays depends on the model (round 6, same box, ms per step with / without): egnn_equihnns 1.122 / 1.181 at
        # batch 256 and 5.24 / 5.54 on PCQM batches of 1024, faformer_equihnns 18.9 / 19.0 -- but equiformer_equihnns 5.20 / 4.95,
        # mhnnm 0.872 / 0.810 at batch 32, egnn_equihnn 1.85 / 1.79: the staged -> live copy, the signal and the rendezvous of the
        # two streams cost a step more than a short index build saves, and neither the index graph's own time (93 us for both
        # EGNN models) nor the step's length predicts the sign.  So the trainer MEASURES (``prefetch_policy == "auto"``): the
        # first captured steps run in the built-ahead form, a window of them is timed, the same number in the in-step form, and
        # the faster form stays (``calibrating``, ``calibration``; the two forms are numerically the same step).
        # EQH_NO_INDEX_PREFETCH=1 / EQH_INDEX_PREFETCH=1 (or assigning ``index_prefetch``) pin a form.
        import os
        self.prefetch_policy = ("off" if os.environ.get("EQH_NO_INDEX_PREFETCH") else
                                "on" if os.environ.get("EQH_INDEX_PREFETCH") else "auto")
        self._prefetch_on = self.prefetch_policy != "off"
        self._cal = None                    # calibration state (auto policy): see _calibrate
        self._alt_slots = None              # the captured steps of the form that is not running, while calibrating
        self.calibration = None             # {"built_ahead_ms": .., "in_step_ms": .., "chosen": ..} once decided
        self.index_build_us = None          # the index graph alone, timed at the first capture (_capture_index)
        self.signal_in_model = not os.environ.get("EQH_PREFETCH_AT_HEAD")    # start the next index at the model's ops.signal_point()
        # which of the marked points releases it.  Measured at the BASELINE batch (same box, ms per step): index at the head
        # of the step graph 1.137; built ahead and released at the step's head 1.138 (it then competes with the chip-filling
        # front-end kernels), behind the EGNN edge kernel 1.112, before the read-out head 1.098, before the closing
        # reductions 1.160 (too late: it spills into the next step)
        # (round 6, second scan: before the conv stack's last application 1.075 / 1.076 against 1.079-1.083 before the read-out
        # head and 1.10 before its first or second application.)  A comma list: the first point the model's step reaches.
        self.signal_at = os.environ.get("EQH_PREFETCH_SIGNAL", "conv_last,readout")
        self._signal = None                 # ops.StepSignal: posted by a node of every step graph with index prefetch
        self.prefetch_hits = 0              # steps whose index had been built ahead
        self.prefetch_misses = 0            # steps that had to build it first (no next_data was given for them)
        self._mode_agreed = False           # the ranks have settled on one collective_mode (first capture, _agree_on_mode)
        from . import ops
        self.scratch = ops.MergedScratch()  # accumulators of the merged weights: owned here, part of the captured graphs
        self.bflat = []                     # flat buffers (one per dtype) behind the model's buffers (multi-rank BatchNorm)
        self.live = None
        self.opt = None
        self.slots = {}
        self.wflat = None
        self.gflat = None
        self.pflat = None
        self.gb_params = []
        self.others = []
        self.other_slots = []
        self.n_w = 0
        self.fused_head = "head" in inspect.signature(model.forward).parameters
        self._has_buffers = self._multi() and any(True for _ in model.buffers())
        # (--normalization bn: BatchNorm inside the MLPs takes its training statistics over the REAL rows of the padded static
        # batch -- HyperIndex.pad_masks, layers.MLP._norm -- so those models replay under hipGraph like the others)
        if broadcast_from_rank0 and self._multi():
            for t in list(model.parameters()):
                dist.broadcast(t.data, src=0)
            self._flatten_buffers()
            self.sync_buffers()

    def _w(self) -> int:
        return _world() if self.collective else 1

    def _multi(self) -> bool:
        """Whether this trainer takes the multi-rank path (collectives, 1 / world gradient scale)."""
        return self._w() > 1 or self.force_collective

    def _flatten_buffers(self):
        """Re-seat the model's buffers (BatchNorm running statistics, batch counters) as views into ONE flat tensor per
        dtype, so that DDP's per-forward buffer broadcast is one collective per dtype instead of one per buffer."""
        if self.bflat:
            return
        groups = {}
        for b in self.model.buffers():
            groups.setdefault(b.dtype, []).append(b)
        for dt, bs in groups.items():
            pad = lambda k: (k + 3) // 4 * 4
            flat = torch.zeros(sum(pad(b.numel()) for b in bs), dtype=dt, device=bs[0].device)
            off = 0
            for b in bs:
                k = b.numel()
                flat[off:off + k].copy_(b.data.reshape(-1))
                b.data = flat[off:off + k].view_as(b)
                off += pad(k)
            self.bflat.append(flat)

    def close(self):
        """Detach this trainer from the model: the persistent gradient accumulators the kernels add into
        (``param._eqh_gbuf``) belong to the trainer, and a later backward outside it (TrainStep, a user loop, another
        GraphedTrainStep) must hand its gradients to autograd again."""
        for p in self.model.parameters():
            if hasattr(p, "_eqh_gbuf"):
                del p._eqh_gbuf
        self.gb_params, self.slots, self.live = [], {}, None
        self._alt_slots, self._cal = None, None

    def _buffer_snapshot(self):
        return [b.detach().clone() for b in self.model.buffers()]

    def _buffer_restore(self, snap):
        """The probe, warm-up and capture passes run the model in training mode without being optimiser steps: put the
        BatchNorm running statistics / batch counters (mhnnm, egnn_equihnnm) back, so that they move once per step
        as in the reference and in TrainStep."""
        with torch.no_grad():
            for b, s0 in zip(self.model.buffers(), snap):
                b.copy_(s0)

    def sync_buffers(self):
        """DDP's per-forward buffer broadcast from rank 0 (BatchNorm running statistics): one collective per dtype."""
        if self._multi():
            self._flatten_buffers()
            for flat in self.bflat:
                dist.broadcast(flat, src=0)

    @staticmethod
    def _key(b):
        return (b.x.shape[0], b.edge_attr.shape[0], b.edge_index0.shape[0], b.y.shape[0])

    def _loss(self, data):
        nb = getattr(data, "num_real_graphs", None) or data.y.shape[0]
        if getattr(data, "_index_is_live", False):
            data._hyper_index.reset_lazy()      # (built ahead by the index graph and refreshed before this pass: see _capture_index)
        elif hasattr(data, "_hyper_index"):
            data._hyper_index = None
        from . import ops
        if self.fused_head and data.y.is_cuda:
            # pool + output MLP + MSE and their backward in one launch; the gradient that loss.backward()
            # feeds in is the implicit 1, so the head's parameter gradients go straight to the accumulators
            return self.model(data, head=(data.y, nb, True))
        out = self.model(data)[:nb]
        return ops.mse_loss(out, data.y[:nb]) if out.is_cuda else F.mse_loss(out, data.y[:nb])

    def _fwd_bwd(self, data):
        for p in self.model.parameters():
            p.grad = None
        if self.wflat is not None:
            self.wflat.zero_()
        if self.scratch.buf is not None:
            self.scratch.buf.zero_()
        return self._loss_backward(data)

    def _loss_backward(self, data):
        """forward + backward with the accumulating gradient reductions of the fused kernels deferred into
        one launch (ops.defer_begin / defer_flush; the fused readout head produces its parameter gradients
        during the forward pass, so the window opens before it), then the packing of the remaining gradients."""
        from . import ops
        dev = data.y.device
        defer = self.gflat is not None and dev.type == "cuda"
        if defer:
            # accumulators of merged weights: this trainer's persistent scratch, which its update kernel clears (no fill
            # launch per step); with keep_grads a fresh zero-filled slab per window
            ops.defer_begin(dev, scratch=None if self.keep_grads else self.scratch)
        try:
            loss = self._loss(data)
            unit = getattr(self, "_unit", None)         # (loss.backward() would fill a fresh ones_like(loss) every step)
            if unit is None or unit.device != loss.device or unit.dtype != loss.dtype:
                unit = self._unit = torch.ones((), dtype=loss.dtype, device=loss.device)
            loss.backward(unit)
        finally:
            if defer:
                ops.defer_flush(dev)
        self._join()
        self._gather_grads()
        return loss

    def _join(self):
        from . import ops
        if self.gb_params:
            ops.join_wgrad_stream(self.gb_params[0].device)

    def _gather_grads(self):
        """After a backward: the weights the GEMMs accumulated in place already sit in the head of
        ``gflat``; the gradients autograd allocated (biases, norms, embeddings, ...) are packed behind
        them by one batched copy (the padding between slots stays zero, which Adam maps to a zero update)."""
        for p in self.gb_params:
            p.grad = p._eqh_gbuf
        if self.gflat is None:
            return
        if self.others:
            from . import ops
            ops.copy_many(self.other_slots, [p.grad for p in self.others])

    def _setup_grad_buffers(self, data, live):
        """Give every weight that is used ONLY through ops.linear, and every bias / LayerNorm vector used
        ONLY through the fused kernels, a persistent gradient accumulator (the head of one flat buffer,
        zeroed by a single fill per step): the kernels add into it, so a layer applied L times costs no
        autograd add kernels.  A probe pass drops any parameter that autograd still produces a gradient
        for (i.e. that is also used some other way)."""
        from . import ops
        cand = [p for p in live if (id(p) in ops.LINEAR_PARAMS and p.dim() == 2) or id(p) in ops.ACC_PARAMS]
        dev, dt = live[0].device, live[0].dtype
        if cand:
            probe = torch.zeros(sum(p.numel() for p in cand), dtype=dt, device=dev)
            off = 0
            for p in cand:
                p._eqh_gbuf = probe[off:off + p.numel()].view_as(p)
                off += p.numel()
            for p in self.model.parameters():
                p.grad = None
            self._loss(data).backward()
            self._join()
            for p in cand:
                if p.grad is not None:
                    del p._eqh_gbuf
        self.gb_params = [p for p in cand if hasattr(p, "_eqh_gbuf")]
        gb = {id(p) for p in self.gb_params}
        self.others = [p for p in live if id(p) not in gb]
        order = self.gb_params + self.others
        pad = lambda k: (k + 63) // 64 * 64   # every view starts 256-byte aligned (the kernels want 16)
        n = sum(pad(p.numel()) for p in order)
        self.n_w = sum(pad(p.numel()) for p in self.gb_params)
        # one flat buffer each for parameters and gradients, same element order: the optimiser, the
        # all-reduce and the zero-fill all see ONE tensor instead of ~100
        self.gflat = torch.zeros(n, dtype=dt, device=dev)
        pflat = torch.zeros(n, dtype=dt, device=dev)
        self.other_slots = []
        off = 0
        for p in order:
            k = p.numel()
            pflat[off:off + k].copy_(p.data.reshape(-1))
            p.data = pflat[off:off + k].view_as(p)
            if id(p) in gb:
                p._eqh_gbuf = self.gflat[off:off + k].view_as(p)
            else:
                self.other_slots.append(self.gflat[off:off + k].view_as(p))
            off += pad(k)
        self.wflat = self.gflat[:self.n_w] if self.n_w else None
        self.pflat = torch.nn.Parameter(pflat)
        self.pflat.grad = self.gflat

    def _bootstrap(self, data):
        """Eager first step: discovers the live parameters, lays out the flat buffers and creates
        the capturable fused Adam (also performs every lazy one-time initialisation of the HIP
        library and of the GEMM libraries before anything is captured)."""
        from . import ops
        for p in self.model.parameters():      # accumulators of an earlier trainer on this model are stale
            if hasattr(p, "_eqh_gbuf"):
                del p._eqh_gbuf
        ops.LINEAR_PARAMS.clear()
        ops.ACC_PARAMS.clear()
        self.gflat = None
        snap = self._buffer_snapshot()
        loss = self._fwd_bwd(data)
        self.live = [p for p in self.model.parameters() if p.grad is not None]
        self._setup_grad_buffers(data, self.live)
        self._buffer_restore(snap)             # discovery + probe passes were not steps
        loss = self._fwd_bwd(data)
        # Adam is elementwise: one update over the flat tensor equals the per-parameter updates
        self.opt = FlatAdam(self.pflat, lr=self.lr, weight_decay=self.wd)
        self.opt.zero_grad_in_step = not self.keep_grads
        self.opt.zero_also = lambda: self.scratch.buf
        if self._multi():
            dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM)     # (eager: also creates the communicator before any capture)
            self.opt.grad_scale = 1.0 / self._w()     # the average is folded into the update
        self.opt.step()
        return loss.detach()

    def _captured_all_reduce(self):
        """The gradient all-reduce as a node of the step graph.  A backend that cannot be stream-captured raises here
        (RCCL: ``DistBackendError``, a RuntimeError); that -- and nothing else -- is what makes a capture fall back to the
        split form.  (A method so that the two-rank tests can stand in a backend that refuses on one rank.)"""
        dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM)

    def _capture_pass(self, static, in_graph: bool):
        """One capture of the step for the static batch.  ``in_graph``: buffer broadcast, all-reduce and update are nodes of
        the one graph; else the graph ends after the backward pass and the update is a second graph (g_opt).  Raises
        ``CollectiveCaptureRefused`` when -- and only when -- a captured collective call failed; an error of a kernel launch
        or of an argument check anywhere else in the pass propagates as itself (the split form would hit it too)."""
        from . import ops
        multi = self._multi()
        for p in self.model.parameters():
            p.grad = None
        g_bwd = torch.cuda.CUDAGraph()
        tl = ops.TIMELINE            # bench.py's in-graph kernel timing: only the captured pass is recorded
        if tl is not None:
            tl.reset()
        refused = None
        try:
            # thread_local: the RCCL watchdog thread may query events while this thread captures
            with torch.cuda.graph(g_bwd, capture_error_mode="thread_local"):
                if tl is not None:
                    for _ in range(4):
                        tl.pair("stamp_pair")
                if in_graph and self._has_buffers:
                    try:
                        self.sync_buffers()
                    except RuntimeError as exc:
                        refused = exc
                if refused is None:
                    if self.wflat is not None and self.keep_grads:
                        self.wflat.zero_()
                    sig = self._signal if getattr(static, "_index_is_live", False) else None
                    if sig is not None:
                        # the post that releases the NEXT batch's index build: where the model marks it (ops.signal_point:
                        # behind the EGNN edge kernel, when the chip stops being full), else at the head of the step
                        sig.armed, sig.at = True, self._signal_name
                        ops.SIGNAL = sig
                        if not (self.signal_in_model and self._signal_reached):
                            sig.post()
                    try:
                        loss = self._loss_backward(static)
                    finally:
                        if sig is not None:
                            ops.SIGNAL = None
                            if sig.armed:       # (cannot happen after the probe; a step without its post would stall the next index build until the wait's timeout)
                                sig.post()
                    if in_graph:
                        try:
                            self._captured_all_reduce()
                        except RuntimeError as exc:
                            refused = exc
                    if refused is None and (in_graph or not multi):
                        self.opt.step()     # nothing the host must do between backward and update: one graph, one launch
        except RuntimeError:
            if refused is None:
                raise
            # (the refused collective also invalidated the capture: ending it raised again -- the refusal is the cause)
        if refused is not None:
            raise CollectiveCaptureRefused(f"{type(refused).__name__}: {refused}") from refused
        g_opt = None
        if multi and not in_graph:
            g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_opt, capture_error_mode="thread_local"):
                self.opt.step()
        return g_bwd, g_opt, loss

    def _agree_on_mode(self, mode: str) -> str:
        """All ranks take the SAME form of the step: "in_graph" only if every rank captured the collective.  One eager
        all-reduce (MIN) of a success flag, at the FIRST capture of the run -- every rank reaches it at its second step,
        after the eager bootstrap step, whatever its batches look like -- and the outcome then binds every later capture
        (``_capture`` raises if a later bucket cannot follow it: a rank that changed form alone would leave the others
        waiting in a collective it never enters)."""
        flag = torch.tensor([1 if mode == "in_graph" else 0], dtype=torch.int32, device=self.gflat.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return "in_graph" if int(flag.item()) == 1 else "split"

    def _capture(self, static):
        multi = self._multi()
        # autograd graphs of earlier eager passes that are only kept alive by reference cycles hold AccumulateGrad nodes bound
        # to the stream they ran on; a capture that meets one is invalidated (and capture_end of an invalidated capture
        # segfaults on ROCm 7.2): collect them first (a capture is rare and costs ~0.3 s anyway)
        import gc
        gc.collect()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        snap = self._buffer_snapshot()
        from . import ops

        class _Probe:       # which signal points does this model's step reach (ops.signal_point)?
            armed = False

            def __init__(self):
                self.seen = set()
        probe = _Probe()
        with torch.cuda.stream(side):  # warm-up on a side stream, as graph capture requires
            for _ in range(2):
                ops.SIGNAL = probe
                try:
                    self._fwd_bwd(static)
                finally:
                    ops.SIGNAL = None
            self._buffer_restore(snap)
        torch.cuda.current_stream().wait_stream(side)
        # the first of the preferred points this model reaches (conv-stack models: before the last application; models without
        # the stack: before the read-out head; else the head of the step)
        self._signal_name = next((n for n in self.signal_at.split(",") if n in probe.seen), None)
        self._signal_reached = self._signal_name is not None
        self.scratch.freeze()          # its address is about to become part of a graph
        prefetch = self._capture_index(static) if self._prefetch_on else None
        mode = "none"
        if multi:
            mode = "in_graph" if (self.graph_collective and dist.get_backend() in self.in_graph_backends
                                  and self.collective_mode != "split") else "split"
        first = multi and not self._mode_agreed
        cap = None
        if mode == "in_graph":
            try:
                cap = self._capture_pass(static, True)
            except CollectiveCaptureRefused as exc:    # the split form is always available -- if every rank takes it
                if not first:
                    raise RuntimeError("GraphedTrainStep: the ranks agreed on the in-graph all-reduce at the first capture "
                                       f"and this rank can no longer capture it ({exc})") from exc
                self.capture_error = str(exc)
                torch.cuda.synchronize()
                self._buffer_restore(snap)
                mode = "split"
            if mode == "split":
                # the abandoned pass's CUDAGraph sits in the exception's traceback (a reference cycle): destroy it NOW --
                # a hipGraphDestroy issued by the collector in the middle of the next capture aborts the process
                gc.collect()
        if first:
            agreed = self._agree_on_mode(mode)
            self._mode_agreed = True
            if agreed != mode:         # this rank captured the collective, another one could not: drop the graph, follow
                self.capture_error = "another rank could not capture the collective"
                cap, mode = None, agreed
                self._buffer_restore(snap)
                gc.collect()
        if cap is None:
            cap = self._capture_pass(static, False)
        g_bwd, g_opt, loss = cap
        self.collective_mode = mode
        if self.wflat is not None and not self.keep_grads:
            self.wflat.zero_()       # what the captured update leaves behind after every replay: zeros to accumulate into
            if self.scratch.buf is not None:
                self.scratch.buf.zero_()
        return {"static": static, "bwd": g_bwd, "opt": g_opt, "loss": loss, "mode": mode, "prefetch": prefetch}

    def _capture_index(self, static):
        """Set up the index prefetch of one bucket, or return None when the model's index cannot be built ahead (no index
        on the batch, or a neighbour search on coordinates other than the batch's own ``pos``).

        ``staged``: a second packed copy of the static batch that a step's successor is written into; ``g_index``: a hipGraph
        that builds the staged batch's HyperIndex (CSR sorts, kNN, transposed kNN CSR) -- replayed on ``stream`` beside the
        running step; ``live``: a clone of that index, installed on the static batch, which is ALL the step graph reads;
        ``dsts / srcs``: the tensor pairs (index tensors + the batch's flat buffer) that ONE batched copy at the head of
        every step moves from staged to live (3 MB, ~4.4 us).  One step graph per bucket as before (two step graphs over two
        sets of live buffers would save the copy and double the captures and the graphs' memory)."""
        from .index import HyperIndex
        ixw = getattr(static, "_hyper_index", None)
        flat = getattr(static, "_flat", None)
        if ixw is None or flat is None or not static.pos.is_cuda:
            return None
        want = (static.pos.data_ptr(), tuple(static.pos.shape), static.pos.dtype)
        keys = list(ixw._knn.keys())
        if any(ixw._knn_pos.get(k) != want for k in keys) or any(static.pos.shape[0] - 1 < k for k, m in keys if m == 1):
            return None
        staged = static.packed()
        staged.num_real_graphs = getattr(static, "num_real_graphs", None)
        from . import ops
        if self._signal is None:
            self._signal = ops.StepSignal(static.pos.device)
        stream = getattr(self, "_prefetch_stream", None)
        if stream is None:
            stream = self._prefetch_stream = torch.cuda.Stream()

        def build():
            staged._hyper_index = None
            ix = HyperIndex.from_batch(staged)
            for k, m in keys:
                ix.knn(staged.pos, k, m)
            return ix
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            build()                                  # (warm-up on the stream the graph will be replayed on)
        torch.cuda.current_stream().wait_stream(stream)
        g_index = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_index, capture_error_mode="thread_local"):
            ix_staged = build()
        # the index graph alone (reported by bench.py; a pure function of the staged batch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            g_index.replay()
            e0.record(stream)
            for _ in range(3):
                g_index.replay()
            e1.record(stream)
        e1.synchronize()
        self.index_build_us = e0.elapsed_time(e1) * 1e3 / 3
        live, dsts, srcs = ix_staged.live_clone(static)
        static._hyper_index, static._index_is_live = live, True
        as_words = lambda t: t.view(torch.float32) if t.dtype != torch.float32 else t
        nf = flat.numel() // 4 * 4
        dsts = [as_words(t).reshape(-1) for t in dsts] + [flat[:nf].view(torch.float32)]
        srcs = [as_words(t).reshape(-1) for t in srcs] + [staged._flat[:nf].view(torch.float32)]
        # (ops.StreamEvent: no system-scope fence -- a torch.cuda.Event recorded at the head of every step writes the L2 back and
        # invalidates it; EQH_TORCH_EVENTS=1 restores those for A/B runs)
        import os
        light = not os.environ.get("EQH_TORCH_EVENTS")
        free = ops.StreamEvent() if light else torch.cuda.Event()
        free.record(torch.cuda.current_stream())
        ready = ops.StreamEvent() if light else None
        return {"staged": staged, "g_index": g_index, "ix_staged": ix_staged, "live": live, "dsts": dsts, "srcs": srcs,
                "stream": stream, "free": free, "ready": None, "ready_ev": ready, "holds": None, "light": light}

    @property
    def index_prefetch(self) -> bool:
        """Whether newly captured steps take the built-ahead form.  Assigning it pins the form (no calibration)."""
        return self._prefetch_on

    @index_prefetch.setter
    def index_prefetch(self, on: bool):
        self._prefetch_on = bool(on)
        self.prefetch_policy = "on" if on else "off"
        self._cal = None
        if self._alt_slots is not None:
            self._alt_slots = None
            import gc
            gc.collect()

    @property
    def calibrating(self) -> bool:
        """True until the auto policy has timed both forms of the step and kept one (callers that time steps themselves --
        bench.py -- run steps until this clears before their own warm-up)."""
        return self.prefetch_policy == "auto" and self.calibration is None

    CAL_WARM, CAL_STEPS = 3, 12          # replays before a timed window, steps in it

    def _calibrate(self, key):
        """Auto policy, called at the head of every graphed step until the decision: per form, one capture step, CAL_WARM
        replays, then CAL_STEPS steps between two device synchronisations (wall clock: the two streams overlap, so no single
        stream's events see a step).  A step of another bucket restarts the window."""
        import gc
        import time
        c = self._cal
        if c is None:
            c = self._cal = {"form": "built_ahead", "key": key, "n": 0, "t": {}, "restarts": 0}
        if key != c["key"]:
            c["key"], c["n"] = key, 0
            c["restarts"] += 1
            if c["restarts"] > 16:          # buckets alternate too fast to time a window: keep the form that is running
                self._decide(None)
            return
        c["n"] += 1
        first_timed = 1 + self.CAL_WARM + 1
        if c["n"] == first_timed:
            torch.cuda.synchronize()
            c["t0"] = time.perf_counter()
        elif c["n"] == first_timed + self.CAL_STEPS:
            torch.cuda.synchronize()
            c["t"][c["form"]] = (time.perf_counter() - c["t0"]) / self.CAL_STEPS * 1e3
            if c["form"] == "built_ahead":  # now the other form: every bucket is captured again, the graphs so far are kept aside
                self._alt_slots, self.slots = self.slots, {}
                self._prefetch_on = False
                c["form"], c["n"] = "in_step", 1
            else:
                self._decide(c["t"])

    def _decide(self, t):
        import gc
        ahead = t is not None and t["built_ahead"] < 0.99 * t["in_step"]
        if t is None:
            ahead = self._prefetch_on
        elif ahead:                          # back to the graphs of the first window
            self.slots, self._alt_slots = self._alt_slots, None
            self._prefetch_on = True
        self._alt_slots = None
        gc.collect()                         # (dropped graphs are destroyed now, not inside a later capture)
        self.calibration = {"built_ahead_ms": None if t is None else round(t["built_ahead"], 4),
                            "in_step_ms": None if t is None else round(t["in_step"], 4),
                            "steps_per_window": self.CAL_STEPS, "chosen": "built_ahead" if ahead else "in_step"}
        self._cal = None

    @staticmethod
    def _token(data):
        """Identity of a batch's CONTENTS: loaders hand the same device buffers out again with new molecules (and bump
        ``_generation``, fit.BucketedLoader)."""
        return (id(data), getattr(data, "_generation", 0))

    def _stage(self, pf, data, after_signal=None):
        """Write ``data`` into the bucket's staged batch and build its index there, on the prefetch stream: ordered behind
        the head copy of the last step that read the staged buffers (``free``), not behind the step itself -- and, with
        ``after_signal`` (the count the running step's signal node brings the counter to), held back until that step has
        reached its signal point."""
        st = pf["staged"]
        with torch.cuda.stream(pf["stream"]):
            if pf["light"]:
                pf["free"].wait(pf["stream"])
            else:
                pf["stream"].wait_event(pf["free"])
            if after_signal is not None:
                self._signal.wait(after_signal)
            lay = getattr(data, "_layout", None)
            if lay is not None and lay == getattr(st, "_layout", None):
                st._flat.copy_(data._flat, non_blocking=True)
            else:
                for f in data.__dataclass_fields__:
                    v = getattr(data, f)
                    if torch.is_tensor(v):
                        getattr(st, f).copy_(v, non_blocking=True)
            pf["g_index"].replay()
            if pf["light"]:
                ready = pf["ready_ev"]
                ready.record(pf["stream"])
            else:
                ready = torch.cuda.Event()
                ready.record(pf["stream"])
        pf["ready"], pf["holds"] = ready, self._token(data)

    def _refresh(self, pf):
        """Head of a step: staged -> live (index tensors and the batch itself) in one launch on the step's stream."""
        from . import ops
        cur = torch.cuda.current_stream()
        if pf["light"]:
            pf["ready"].wait(cur)
            ops.copy_many(pf["dsts"], pf["srcs"])
            pf["free"].record(cur)
        else:
            cur.wait_event(pf["ready"])
            ops.copy_many(pf["dsts"], pf["srcs"])
            pf["free"] = torch.cuda.Event()
            pf["free"].record(cur)

    def step(self, data, next_data=None) -> torch.Tensor:
        """One training step on ``data``.  ``next_data``: the batch of the NEXT call, if the caller has it (a loader with one
        batch of look-ahead, ``with_next``): its per-batch index is then built beside this step instead of at the head of
        the next one.  Numerically the two forms are the same step."""
        if self.live is None:
            return self._bootstrap(data)
        key = self._key(data)
        if self.prefetch_policy == "auto" and self.calibration is None:
            self._calibrate(key)
        slot = self.slots.get(key)
        if slot is None:
            if hasattr(data, "packed"):
                static = data.packed()       # one flat buffer: refreshed by ONE copy per step
            else:
                static = type(data)(**{f: (getattr(data, f).clone() if torch.is_tensor(getattr(data, f))
                                           else getattr(data, f)) for f in data.__dataclass_fields__})
            static.num_real_graphs = getattr(data, "num_real_graphs", None)
            slot = self.slots[key] = self._capture(static)
        st = slot["static"]
        pf = slot.get("prefetch")
        if pf is not None:
            if pf["holds"] != self._token(data):
                self.prefetch_misses += 1
                pf["stream"].wait_stream(torch.cuda.current_stream())     # (data may have been produced on this stream)
                self._stage(pf, data)
            else:
                self.prefetch_hits += 1
            self._refresh(pf)
            pf["holds"] = None
        else:
            lay = getattr(data, "_layout", None)
            if lay is not None and lay == getattr(st, "_layout", None):
                st._flat.copy_(data._flat, non_blocking=True)
            else:
                for f in data.__dataclass_fields__:
                    v = getattr(data, f)
                    if torch.is_tensor(v):
                        getattr(st, f).copy_(v, non_blocking=True)
        self.opt.sync_lr()
        if self._has_buffers and slot["mode"] != "in_graph":
            self.sync_buffers()
        slot["bwd"].replay()
        if pf is not None:
            self._signal.posted += 1      # (one post per replay of a step graph with a live index)
        if slot["opt"] is not None:
            dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM)
            slot["opt"].replay()
        if next_data is not None:
            nslot = self.slots.get(self._key(next_data))
            npf = nslot.get("prefetch") if nslot is not None else None
            if npf is not None:
                self._stage(npf, next_data, after_signal=self._signal.posted if pf is not None else None)
        return slot["loss"].detach()


class GraphedEvalStep:
    """``model(data)`` in eval mode under ``no_grad`` -- the validation / test passes the reference runs every epoch
    (main.py:65-87,89-132) -- captured ONCE per static shape bucket as a forward-only hipGraph and replayed: an eager forward
    pass is bound by Python launch issue (4-6 x the replayed time), exactly like the eager training step.  Batches must be
    padded to bucket extents (``batch.pad_batch`` / ``fit.BucketedLoader``); the returned predictions are a STATIC tensor,
    valid until the next call (rows past ``num_real_graphs`` are padding).  The graphs read the parameters and buffers in
    place, so they follow the optimiser and ``load_state_dict`` without re-capture.  A captured graph holds the ADDRESSES of
    the parameters and buffers: every call compares them with the ones recorded at capture, and when they have moved (a
    GraphedTrainStep's bootstrap re-seats ``p.data`` into its flat buffer; ``model.to`` / a re-seated BatchNorm buffer) the
    stale graphs are dropped and the bucket is captured again instead of replaying reads of freed storage."""

    def __init__(self, model: nn.Module):
        self.model = model
        self.slots = {}
        self.addresses = None
        self.recaptures = 0

    def _addresses(self):
        return tuple(t.data_ptr() for t in self.model.parameters()) + tuple(t.data_ptr() for t in self.model.buffers())

    def close(self):
        self.slots = {}

    def _capture(self, static):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                static._hyper_index = None
                self.model(static)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        static._hyper_index = None
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = self.model(static)
        return {"static": static, "graph": g, "out": out}

    @torch.no_grad()
    def __call__(self, data) -> torch.Tensor:
        if self.model.training:
            raise RuntimeError("GraphedEvalStep: put the model in eval mode first (the captured graphs are eval-mode forwards)")
        key = GraphedTrainStep._key(data)
        addr = self._addresses()
        if addr != self.addresses:
            if self.slots:
                self.recaptures += 1
            self.slots = {}
            self.addresses = addr
        slot = self.slots.get(key)
        if slot is None:
            if hasattr(data, "packed"):
                static = data.packed()
            else:
                static = type(data)(**{f: (getattr(data, f).clone() if torch.is_tensor(getattr(data, f))
                                           else getattr(data, f)) for f in data.__dataclass_fields__})
            static.num_real_graphs = getattr(data, "num_real_graphs", None)
            slot = self.slots[key] = self._capture(static)
        st = slot["static"]
        lay = getattr(data, "_layout", None)
        if lay is not None and lay == getattr(st, "_layout", None):
            st._flat.copy_(data._flat, non_blocking=True)
        else:
            for f in data.__dataclass_fields__:
                v = getattr(data, f)
                if torch.is_tensor(v):
                    getattr(st, f).copy_(v, non_blocking=True)
        slot["graph"].replay()
        return slot["out"]
