"""Parameter-gradient plumbing: persistent accumulators, deferred (batched) weight / bias gradients and slab
reductions, gradient fan-in.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import ctypes
import os

import torch

from .. import hip
from ._base import (_DEFER, _f32c, _ptr, _rows_ld, _stream, _workspace)
from .products import (GemmProblem, USE_X6, X6_DEEP_ROWS, X6_WGRAD_ROWS, gemm, gemm_batch, gemm_out_ok, gemm_supported)


class GradFan:
    """Collector for the gradient of a tensor that is added, unchanged, to the input of several bias_relu_ln layers (the
    layer-independent term of conv.py:179-180 over the L applications of the shared conv): the LayerNorm backward kernels
    sum it (hg_bias_relu_ln_bwd_acc), and _FanSource hands the sum to the tensor's producer."""

    def __init__(self):
        self.buf, self.n = None, 0


class _FanSource(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, fan):
        ctx.fan = fan
        ctx.set_materialize_grads(False)
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        fan = ctx.fan
        if fan.buf is None:          # no consumer used the collector
            return g, None
        return (fan.buf if g is None else fan.buf + g), None


def fanout(t):
    """(t as a new autograd leaf-of-this-node, GradFan): consumers that register with the GradFan (linear_add(..., fan=),
    bias_relu_ln(..., fan=)) deliver their gradient of ``t`` through it instead of through autograd's adds."""
    fan = GradFan()
    return _FanSource.apply(t, fan), fan


WGRAD_ON_SIDE_STREAM = False
_WGRAD_STREAMS = {}


def wgrad_stream(device):
    key = torch.device(device).index
    if key not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return _WGRAD_STREAMS[key]


def join_wgrad_stream(device):
    if WGRAD_ON_SIDE_STREAM and _WGRAD_STREAMS:
        torch.cuda.current_stream(device).wait_stream(wgrad_stream(device))


class MergedScratch:
    """Persistent accumulators of the merged weights (ops.merged_weight) for ONE owner -- a GraphedTrainStep: carved in
    call order from one buffer that lives across steps and is cleared by the update kernel (eqh_adam_step's zero_also)
    instead of a fresh zero-filled slab, i.e. a fill launch, per step.  The buffer's address is baked into the owner's
    captured graphs, so it may grow only until ``freeze()`` (the owner's first capture); after that running out of room
    is an error, never a reallocation."""

    def __init__(self):
        self.buf, self.cur, self.frozen = None, 0, False

    def freeze(self):
        self.frozen = True

    def take(self, n, device):
        need = self.cur + (n + 63) // 64 * 64
        if self.buf is None or self.buf.numel() < need:
            if self.frozen or torch.cuda.is_current_stream_capturing():
                raise RuntimeError("merged-weight scratch of a captured trainer cannot grow (it is part of the graph)")
            old = self.buf
            self.buf = torch.zeros(max(2 * need, 1 << 18), dtype=torch.float32, device=device)
            if old is not None:
                self.buf[:old.numel()].copy_(old)
        assert self.buf.device == torch.device(device), "one MergedScratch per device"
        out = self.buf[self.cur:self.cur + n]
        self.cur = need
        return out


def defer_begin(device, scratch=None):
    """Start recording the accumulating gradient reductions issued on the current stream of ``device``
    (eqh_defer_begin); they all run in one launch at defer_flush().  ``scratch``: the caller's MergedScratch for the
    accumulators of merged weights during this window (the caller keeps it clear); without one they come from a fresh
    zero-filled slab per window."""
    hip.check(hip.lib().eqh_defer_begin(_stream(device)), "eqh_defer_begin")
    _DEFER["active"] = True
    _DEFER["merged"] = []       # (a window that ended in an exception must not leak its records into this one)
    _DEFER["zslab"] = None
    _DEFER["scratch"] = scratch
    if scratch is not None:
        scratch.cur = 0


def wgrad_batch(entries):
    """hg_wgrad_batch_f32: ``entries`` = [(dy [K,O], x [K,I], alpha, into [O,I] view)], all of one O x I;
    every product is ADDED to its destination, products with the same destination in list order."""
    if not entries:
        return
    O, I = entries[0][3].shape
    order = {}
    for en in entries:   # group by destination, keep first-seen order
        order.setdefault((en[3].data_ptr(), en[3].stride(0)), []).append(en)
    flat = [en for grp in order.values() for en in grp]
    n = len(flat)
    # (operands may be column blocks of wider row-major matrices -- the halves of a [rows, 2 C] hidden activation: used in place)
    dl, xl = [_rows_ld(en[0]) for en in flat], [_rows_ld(en[1]) for en in flat]
    dys, xs = [t for t, _ in dl], [t for t, _ in xl]
    vp, i64, f32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_float * n
    dev = dys[0].device
    L = hip.lib()
    ws_bytes = L.hg_wgrad_batch_workspace_bytes(n, O, I)
    ws = _workspace(ws_bytes, dev)
    _DEFER["keep"].extend(dys + xs)
    hip.check(L.hg_wgrad_batch_f32(n, vp(*[t.data_ptr() for t in dys]), vp(*[t.data_ptr() for t in xs]),
                                   i64(*[t.shape[0] for t in dys]), O, I, f32(*[float(en[2]) for en in flat]),
                                   vp(*[en[3].data_ptr() for en in flat]), i64(*[en[3].stride(0) for en in flat]), 1,
                                   _ptr(ws), ws_bytes, _stream(dev), i64(*[ld for _, ld in dl]), i64(*[ld for _, ld in xl])),
              "hg_wgrad_batch_f32")


def colsum_batch(entries):
    """hg_colsum_batch_f32: ``entries`` = [(x [R,C], rowptr or None, weight_mode, into [C], scale)]; every sum
    is ADDED to its destination, in one launch."""
    n = len(entries)
    if n == 0:
        return
    vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
    dev = entries[0][0].device
    R = i64(*[en[0].shape[0] for en in entries])
    C = i32(*[en[0].shape[1] for en in entries])
    L = hip.lib()
    ws_bytes = L.hg_colsum_batch_workspace_bytes(n, R, C)
    ws = _workspace(max(ws_bytes, 16), dev)
    _DEFER["keep"].extend(en[0] for en in entries)
    hip.check(L.hg_colsum_batch_f32(n, vp(*[en[0].data_ptr() for en in entries]), i64(*[en[0].stride(0) for en in entries]),
                                    vp(*[(en[1].data_ptr() if en[1] is not None else None) for en in entries]),
                                    i32(*[int(en[2]) for en in entries]),
                                    (ctypes.c_float * n)(*[float(en[4]) for en in entries]), R, C,
                                    vp(*[en[3].data_ptr() for en in entries]),
                                    _ptr(ws), ws_bytes, _stream(dev)), "hg_colsum_batch_f32")


def defer_flush(device):
    """Run everything that was deferred: first the weight and bias gradients (one batched launch per shape /
    one for all column sums; their slab reductions are themselves deferred), then all slab reductions."""
    pending = _DEFER["wgrad"]
    _DEFER["wgrad"] = []
    sums = _DEFER["colsum"]
    _DEFER["colsum"] = []
    try:
        by_shape = {}
        x6_mid = []
        for en in pending:
            by_shape.setdefault(tuple(en[3].shape), []).append(en)
        for group in by_shape.values():
            # Deep reductions into a small output (FAFormer's per-edge / per-frame Linears: [256 x 2 M] . [2 M x 128])
            # go one by one to the single-product split-K kernel, which cuts K into as many chunks as there are idle
            # CUs (the batched kernel splits K three ways, right for the ~5 k-row products of the conv layers; the
            # library has no split-K choice for such shapes: 2.1 ms for 129 GFLOP)
            deep = [en for en in group if en[0].shape[0] >= X6_DEEP_ROWS]
            rest = [en for en in group if en[0].shape[0] < X6_DEEP_ROWS]
            with torch.no_grad():
                for dy2, x2, alpha, into in deep:
                    if USE_X6 and gemm_supported(dy2, x2, True, False) and gemm_out_ok(into):
                        # split-K on the bf16 matrix cores: 140-150 TFLOP/s against 75 (hg_wgrad_f32) / 60-70 (library)
                        _DEFER["keep"].extend((dy2, x2))
                        gemm(dy2, x2, trans_a=True, trans_b=False, d=into, out=into, alpha=alpha)
                    else:
                        wgrad(dy2, x2, alpha, into=into)
            # ~10^4-row products (FAFormer's atom-level Linears): whatever their shapes, up to eight of them share one x6
            # launch whose split-K plan fills the chip per product (the library runs a [256 x 15 k].[15 k x 128] product on
            # 8 tiles: 100 us for 1 GFLOP; hg_wgrad_batch_f32 reaches 75 TFLOP/s on the fp32 MFMA)
            mid = [en for en in rest if USE_X6 and en[0].shape[0] >= X6_WGRAD_ROWS and gemm_supported(en[0], en[1], True, False)
                   and gemm_out_ok(en[3])]
            if mid:
                x6_mid.extend(mid)
                rest = [en for en in rest if not any(en is m for m in mid)]
            big = rest and ((rest[0][3].shape[0] + 127) // 128) * ((rest[0][3].shape[1] + 127) // 128) * len(rest) >= 12
            if len(rest) >= 3 or (big and BATCH_LONE_WGRAD):
                # (a lone LARGE product -- the [2176 x 256] gradient of the EGNN's first edge Linear: 34 tiles of 128 x 128 -- fills
                # the chip in the batched bf16 x 3 kernel with a deeper split of K)
                wgrad_batch(rest)
            else:   # too few products of this shape to fill the chip together: the library GEMM is faster
                with torch.no_grad():
                    for dy2, x2, alpha, into in rest:
                        into.addmm_(dy2.t(), x2, alpha=alpha)
        with torch.no_grad():
            for i in range(0, len(x6_mid), 8):
                chunk = x6_mid[i:i + 8]
                for dy2, x2, _, _ in chunk:
                    _DEFER["keep"].extend((dy2, x2))
                gemm_batch([GemmProblem(dy2, x2, True, False, None, into, alpha, 1.0, False, into) for dy2, x2, alpha, into in chunk])
        from ._base import signal_point
        signal_point("tail")
        colsum_batch(sums)
    finally:
        _DEFER["active"] = False
        _DEFER["scratch"] = None      # the window is over: a later window of another caller must not carve from this one
        hip.check(hip.lib().eqh_defer_flush(_stream(device)), "eqh_defer_flush")
        _DEFER["keep"].clear()
    # merged weights (ops.merged_weight): their accumulated gradients are complete now; one backward through each
    # weight-level product hands them on to the parameters
    merged = _DEFER["merged"]
    _DEFER["merged"] = []
    _DEFER["zslab"] = None
    for outs, accs in merged:
        torch.autograd.backward(outs, accs)


def copy_many(dsts, srcs):
    """dst[i].copy_(src[i]) for lists of contiguous fp32 device tensors, in one launch (eqh_copy_many)."""
    n = len(dsts)
    if n == 0:
        return
    srcs = [_f32c(t) for t in srcs]
    vp, i64 = ctypes.c_void_p * n, ctypes.c_int64 * n
    for d, t in zip(dsts, srcs):
        assert d.is_contiguous() and d.numel() == t.numel() and d.dtype == torch.float32
    hip.check(hip.lib().eqh_copy_many(n, vp(*[t.data_ptr() for t in srcs]), vp(*[d.data_ptr() for d in dsts]),
                                      i64(*[d.numel() for d in dsts]), _stream(dsts[0].device)), "eqh_copy_many")


def colsum(x, rowptr=None, weight_mode: int = 0, into=None, scale: float = 1.0):
    """scale * sum_r w_r x[r, :] for a 2-D fp32 matrix through hg_colsum_f32 (bias gradients).  ``rowptr`` +
    ``weight_mode`` (1: [row non-empty], 2: row length) give the row weights; ``into`` is an accumulator
    the result is ADDED to (returns None then)."""
    if not x.is_cuda or x.shape[-1] % 4 or x.dtype != torch.float32 or x.shape[0] > 2_000_000:
        # widths the float4 kernel does not take (and row counts past its 65 535-chunk grid) (the 1-wide output head): the device's generic reduction
        if weight_mode:
            deg = rowptr[1:] - rowptr[:-1]
            x = x * ((deg > 0) if weight_mode == 1 else deg).to(x.dtype)[:, None]
        r = x.sum(0) if scale == 1.0 else x.sum(0) * scale
        if into is None:
            return r
        into.add_(r)
        return None
    R, C = x.shape
    if into is not None and _DEFER["active"] and DEFER_WGRAD:
        # (the leading columns of a wider matrix are summed in place: the batched kernel takes a row stride)
        if not (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.stride(0) >= C and x.data_ptr() % 16 == 0):
            x = _f32c(x)
        _DEFER["colsum"].append((x, rowptr, weight_mode, into, scale))   # runs with all the others at defer_flush
        return None
    x = _f32c(x)
    L = hip.lib()
    out = into if into is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    ws_bytes = L.hg_colsum_workspace_bytes(R, C)
    ws = _workspace(max(ws_bytes, 16), x.device)
    hip.check(L.hg_colsum_f32(_ptr(x), _ptr(rowptr) if rowptr is not None else None, weight_mode, float(scale), R, C,
                              1 if into is not None else 0, _ptr(out), _ptr(ws), ws_bytes, _stream(x.device)),
              "hg_colsum_f32")
    return None if into is not None else out


# Weight gradients.  One [256 x K].[K x 256] product has too few output tiles to fill the chip (the library's
# best kernel runs it at 35 TFLOP/s, 17.4 us; the split-K hg_wgrad_f32 at 12 us plus slabs -- no gain for the
# step as a whole), but a backward pass has 21 of them and nothing reads them before the optimiser.  While
# reductions are deferred (graphed trainer) they are therefore only RECORDED here and run together at
# defer_flush (hg_wgrad_batch_f32): one launch per shape.  Outside deferral the library GEMM is used.
USE_WGRAD_KERNEL = False      # the single-product kernel (ops.wgrad) for immediate weight gradients
DEFER_WGRAD = True            # batched weight gradients at defer_flush
# a lone LARGE product through the batched kernel with a deeper split of K: measured 70 us against the library's 54 for the
# [2176 x 4.7 k] . [4.7 k x 256] gradient of the EGNN's first edge Linear (eight slabs of 2.2 MB) -- off
BATCH_LONE_WGRAD = os.environ.get("EQH_LONE_WGRAD", "0") == "1"


def _wgrad_shape_ok(dy2, x2):
    return (dy2.is_cuda and dy2.dtype == torch.float32 and x2.dtype == torch.float32
            and dy2.shape[1] % 64 == 0 and x2.shape[1] % 64 == 0 and 512 <= dy2.shape[0] <= (1 << 24))


def _wgrad_ok(dy2, x2):
    return USE_WGRAD_KERNEL and _wgrad_shape_ok(dy2, x2)


def _wgrad_deferred(dy2, x2, alpha, into) -> bool:
    """Record alpha * dy2.T @ x2 -> += into for the batched launch at defer_flush; False if not applicable."""
    if not (DEFER_WGRAD and _DEFER["active"] and into is not None and into.stride(1) == 1 and _wgrad_shape_ok(dy2, x2)):
        return False
    _DEFER["wgrad"].append((dy2, x2, alpha, into))
    return True


def wgrad(dy2, x2, alpha: float = 1.0, into=None):
    """alpha * dy2.T @ x2 through hg_wgrad_f32 (split-K fp32 MFMA, fixed order).  ``into``: a [O, I] view
    (possibly a column block of a wider matrix, unit inner stride) the product is ADDED to; returns None
    then, else the new [O, I] tensor."""
    dy2, x2 = _f32c(dy2), _f32c(x2)
    K, O = dy2.shape
    I = x2.shape[1]
    if into is not None:
        assert into.shape == (O, I) and into.stride(1) == 1
        out, ld, acc = into, into.stride(0), 1
    else:
        out, ld, acc = torch.empty((O, I), dtype=torch.float32, device=dy2.device), I, 0
    L = hip.lib()
    ws_bytes = L.hg_wgrad_workspace_bytes(K, O, I)
    ws = _workspace(max(ws_bytes, 16), dy2.device)
    hip.check(L.hg_wgrad_f32(_ptr(dy2), _ptr(x2), K, O, I, float(alpha), _ptr(out), ld, acc, _ptr(ws), ws_bytes,
                             _stream(dy2.device)), "hg_wgrad_f32")
    return None if into is not None else out


SKINNY_WGRAD = not os.environ.get("EQH_NO_SKINNY_WGRAD")


def _wgrad_skinny(dy2, x2, tgt) -> bool:
    """tgt [O x J] += dy2.T @ x2 for an input block of at most 16 columns (the m_i block of the EGNN node MLP) through
    hg_wgrad_skinny_f32: one small launch + the step's deferred slab reduction instead of a split-K library product."""
    if not (SKINNY_WGRAD and dy2.is_cuda and dy2.dim() == 2 and x2.dim() == 2 and x2.shape[1] <= 16 and dy2.shape[0] == x2.shape[0]
            and dy2.dtype == torch.float32 and x2.dtype == torch.float32 and dy2.stride(1) == 1 and x2.stride(1) == 1
            and tgt.stride(1) == 1 and dy2.shape[0] > 0):
        return False
    K, O = dy2.shape
    J = x2.shape[1]
    L = hip.lib()
    ws_bytes = L.hg_wgrad_skinny_workspace_bytes(K, O, J)
    ws = _workspace(max(ws_bytes, 16), dy2.device)
    if _DEFER["active"]:
        _DEFER["keep"].extend((dy2, x2))
    hip.check(L.hg_wgrad_skinny_f32(_ptr(dy2), dy2.stride(0), _ptr(x2), x2.stride(0), K, O, J, 1.0, _ptr(tgt), tgt.stride(0), 1,
                                    _ptr(ws), ws_bytes, _stream(dy2.device)), "hg_wgrad_skinny_f32")
    return True


def _linear_weight_grad(weight, c0, c1, dy2, x2, r0=None, r1=None):
    """Weight gradient dy2.T @ x2 of a Linear over the column block [c0, c1) of ``weight``: added to the
    parameter's persistent accumulator (in place, or recorded for the batched launch of defer_flush) when there
    is one -- returns None then -- else returned for autograd (full parameter shape)."""
    gbuf = getattr(weight, "_eqh_gbuf", None)
    if gbuf is not None:
        tgt = gbuf if c0 is None else gbuf[:, c0:c1]
        if r0 is not None:
            tgt = tgt[r0:r1]
        side = wgrad_stream(dy2.device) if WGRAD_ON_SIDE_STREAM else None
        if side is None and _wgrad_skinny(dy2, x2, tgt):
            pass
        elif side is None and _wgrad_deferred(dy2, x2, 1.0, tgt):
            pass
        elif side is None and _wgrad_ok(dy2, x2):
            wgrad(dy2, x2, into=tgt)
        elif (side is None and USE_X6 and dy2.is_cuda and dy2.shape[0] >= X6_WGRAD_ROWS and gemm_supported(dy2, x2, True, False)
              and gemm_out_ok(tgt)):
            # (shapes the batched kernel does not take, e.g. the 272-wide input of the EGNN node MLP: 93 against 127 us)
            dy2, x2 = _f32c(dy2), _f32c(x2)
            if _DEFER["active"]:
                _DEFER["keep"].extend((dy2, x2))
            gemm(dy2, x2, trans_a=True, trans_b=False, d=tgt, out=tgt)
        elif side is None:
            tgt.addmm_(dy2.t(), x2)
        else:
            # weight gradients are off the critical path of the backward chain: issue them
            # on a second HIP stream (a parallel branch of the captured graph); the trainer
            # joins the stream before the optimiser
            side.wait_stream(torch.cuda.current_stream(dy2.device))
            with torch.cuda.stream(side):
                tgt.addmm_(dy2.t(), x2)
            dy2.record_stream(side)
            x2.record_stream(side)
        return None
    # no accumulator (the parameter also receives gradients from plain autograd ops): the gradient goes to autograd.  A
    # deep product ([256 x 250 k].[250 k x 256] on FAFormer's edge rows) takes the split-K x6 kernel: 0.21 against the
    # library's 0.71 ms
    deep = USE_X6 and dy2.is_cuda and dy2.shape[0] >= X6_DEEP_ROWS and gemm_supported(dy2, x2, True, False)
    if c0 is None and r0 is None:
        if deep:
            return gemm(_f32c(dy2), _f32c(x2), trans_a=True, trans_b=False)
        return wgrad(dy2, x2) if _wgrad_ok(dy2, x2) else dy2.t() @ x2
    dw = torch.zeros_like(weight)
    blk = dw if c0 is None else dw[:, c0:c1]
    blk = blk if r0 is None else blk[r0:r1]
    if deep and blk.stride(1) == 1:
        gemm(_f32c(dy2), _f32c(x2), trans_a=True, trans_b=False, out=blk)
    else:
        blk.copy_(dy2.t() @ x2)
    return dw


def _merged_acc(shape, device):
    """A zeroed accumulator for a merged weight: carved from the window owner's persistent MergedScratch (graphed
    trainer, defer_begin(scratch=...)) or from one zero-filled slab per deferral window (one fill kernel for all merged
    weights of a step)."""
    n = shape[0] * shape[1]
    ms = _DEFER.get("scratch")
    if ms is not None:
        return ms.take(n, device).view(shape)
    slab = _DEFER.get("zslab")
    if slab is None or slab[1] + n > slab[0].numel() or slab[0].device != device:
        slab = [torch.zeros(max(4 * n, 1 << 18), dtype=torch.float32, device=device), 0]
        _DEFER["zslab"] = slab
    out = slab[0][slab[1]:slab[1] + n].view(shape)
    slab[1] += (n + 63) // 64 * 64
    return out
