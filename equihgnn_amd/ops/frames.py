"""FAFormer operators: frame-averaged SwiGLU MLP pieces, edge hidden layer, row dots, gates, attention sums, 3x3 eigh
(fa_former_layer.py).

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import os

import torch

from .. import hip
from ._base import (LINEAR_PARAMS, _acc_target, _f32c, _hand_out, _note_acc, _ptr, _require_gpu, _rows_ld, _stream, _workspace,
                    timed)
from .aggregate import (CSR, _segment_reduce)
from .products import (USE_X6, gemm, mm_nn)
from .grads import (_linear_weight_grad, colsum)


_SEED_POOL = {"buf": None, "next": 0}


class dropout_seeds:
    """``with dropout_seeds(device, n):`` -- the dropout sites inside take their seeds from ONE tensor of n draws (one
    generator launch per forward pass instead of one per site: ~20 launches of 4.6 us in a FAFormer step).  The draw
    happens inside the block, i.e. inside a captured step: every replay sees new seeds."""

    def __init__(self, device, n: int = 64, enabled: bool = True):
        self.device, self.n, self.enabled, self.mine = device, n, enabled, False

    def __enter__(self):
        if self.enabled and _SEED_POOL["buf"] is None:
            _SEED_POOL["buf"] = torch.randint(0, 2 ** 62, (self.n,), dtype=torch.int64, device=self.device)
            _SEED_POOL["next"] = 0
            self.mine = True
        return self

    def __exit__(self, *exc):
        if self.mine:
            _SEED_POOL["buf"] = None
        return False


def _dropout_seed(device, p):
    """A fresh int64 seed in device memory (drawn by torch's generator: graph-safe, a new value per replay)."""
    if p <= 0.0:
        return None
    buf = _SEED_POOL["buf"]
    if buf is not None and buf.device == torch.device(device) and _SEED_POOL["next"] < buf.numel():
        i = _SEED_POOL["next"]
        _SEED_POOL["next"] = i + 1
        return buf[i:i + 1]
    return torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=device)


class _SwigluDropout(torch.autograd.Function):
    """dropout_p(silu(pre[:, :H]) * pre[:, H:]) in one pass each way (faf_swiglu_dropout_*, csrc/faformer_ew.hip)."""

    @staticmethod
    def forward(ctx, pre, p, seed=None):
        _require_gpu(pre, "swiglu_dropout")
        pre2 = _f32c(pre).reshape(-1, pre.shape[-1])
        R, H = pre2.shape[0], pre2.shape[1] // 2
        seed = seed if (seed is not None and p > 0) else _dropout_seed(pre.device, p)
        out = torch.empty((R, H), dtype=torch.float32, device=pre.device)
        hip.check(hip.lib().faf_swiglu_dropout_fwd(_ptr(pre2), R, H, float(p), _ptr(seed), _ptr(out), _stream(pre.device)),
                  "faf_swiglu_dropout_fwd")
        ctx.save_for_backward(pre2)
        ctx.seed, ctx.p, ctx.shape = seed, float(p), pre.shape
        return out.view(*pre.shape[:-1], H)

    @staticmethod
    def backward(ctx, dout):
        (pre2,) = ctx.saved_tensors
        R, H = pre2.shape[0], pre2.shape[1] // 2
        dout = _f32c(dout).reshape(R, H)
        dpre = torch.empty_like(pre2)
        hip.check(hip.lib().faf_swiglu_dropout_bwd(_ptr(pre2), _ptr(dout), R, H, ctx.p, _ptr(ctx.seed), _ptr(dpre),
                                                   _stream(pre2.device)), "faf_swiglu_dropout_bwd")
        return dpre.view(ctx.shape), None, None


class _DropoutMean(torch.autograd.Function):
    """mean over dim -2 of dropout_p(x) in one pass each way (faf_dropout_mean_*).  ``bias``: the bias PARAMETER of the
    Linear that produced x (called with ``bias_grad=False``): its gradient, the column sums of dx, rides the backward
    pass instead of costing another pass over dx (faf_dropout_mean_bwd_colsum)."""

    @staticmethod
    def forward(ctx, x, p, seed=None, bias=None):
        _require_gpu(x, "dropout_mean")
        F_, C = x.shape[-2], x.shape[-1]
        x2 = _f32c(x).reshape(-1, C)
        R = x2.shape[0] // F_
        seed = seed if (seed is not None and p > 0) else _dropout_seed(x.device, p)
        out = torch.empty((R, C), dtype=torch.float32, device=x.device)
        timed("k_drop_mean_fwd", 4 * C * R * (F_ + 1),        # one read of [R * F, C], one write of [R, C]
              lambda: hip.check(hip.lib().faf_dropout_mean_fwd(_ptr(x2), R, F_, C, float(p), _ptr(seed), _ptr(out),
                                                               _stream(x.device)), "faf_dropout_mean_fwd"))
        ctx.seed, ctx.p, ctx.shape, ctx.bias = seed, float(p), x.shape, bias
        return out.view(*x.shape[:-2], C)

    @staticmethod
    def backward(ctx, dout):
        F_, C = ctx.shape[-2], ctx.shape[-1]
        dout = _f32c(dout).reshape(-1, C)
        R = dout.shape[0]
        dx = torch.empty((R * F_, C), dtype=torch.float32, device=dout.device)
        L = hip.lib()
        bias, db = ctx.bias, None
        ws_bytes = L.faf_dropout_mean_bwd_colsum_workspace_bytes(R, F_, C) if (bias is not None and ctx.needs_input_grad[3]) else 0
        if ws_bytes:
            acc = _acc_target(bias)
            tgt = acc if acc is not None else torch.empty(C, dtype=torch.float32, device=dout.device)
            ws = _workspace(ws_bytes, dout.device)
            timed("k_drop_mean_bwd", 4 * C * R * (F_ + 1),
                  lambda: hip.check(L.faf_dropout_mean_bwd_colsum(_ptr(dout), R, F_, C, ctx.p, _ptr(ctx.seed), _ptr(dx),
                                                                  _ptr(tgt), 1 if acc is not None else 0, _ptr(ws), ws_bytes,
                                                                  _stream(dout.device)), "faf_dropout_mean_bwd_colsum"))
            db = None if acc is not None else tgt
        else:
            timed("k_drop_mean_bwd", 4 * C * R * (F_ + 1),
                  lambda: hip.check(L.faf_dropout_mean_bwd(_ptr(dout), R, F_, C, ctx.p, _ptr(ctx.seed), _ptr(dx),
                                                           _stream(dout.device)), "faf_dropout_mean_bwd"))
            if bias is not None and ctx.needs_input_grad[3]:      # (a width the rider does not take)
                db = colsum(dx, into=_acc_target(bias))
        return dx.view(ctx.shape), None, None, db


class _LinearDropoutMean(torch.autograd.Function):
    """mean over the F = 8 frames of dropout_p(h W^T + b) for h [..., 8, K] -- fc2, the per-frame dropout and the frame
    average of FAFormer's frame MLP (fa_former_layer.py:61-120) -- as ONE x6 GEMM whose epilogue drops and averages: the
    [E * 8, C] product (2 GB at the Molecule3D batch) is neither written nor read back.  Backward: the gradient of the
    virtual product is formed by faf_dropout_mean_bwd_colsum (same keep decisions; the bias gradient rides along),
    then the Linear's two products as in ops.linear (input gradient through mm_nn, weight gradient deferred)."""

    @staticmethod
    def forward(ctx, h, weight, bias, p, seed):
        _require_gpu(h, "linear_dropout_mean")
        K, C = h.shape[-1], weight.shape[0]
        h2 = _f32c(h).reshape(-1, K)
        seed = seed if (seed is not None and p > 0) else _dropout_seed(h.device, p)
        out = gemm(h2, weight, trans_b=True, bias=bias, mean8=(float(p), seed if p > 0 else None))
        ctx.save_for_backward(h2, weight)
        ctx.seed, ctx.p, ctx.shape, ctx.bias = seed, float(p), h.shape, bias
        return out.view(*h.shape[:-2], C)

    @staticmethod
    def backward(ctx, dout):
        h2, weight = ctx.saved_tensors
        C = weight.shape[0]
        dout = _f32c(dout).reshape(-1, C)
        R = dout.shape[0]
        dy = torch.empty((R * 8, C), dtype=torch.float32, device=dout.device)
        L = hip.lib()
        bias = ctx.bias
        want_db = bias is not None and ctx.needs_input_grad[2]
        acc = _acc_target(bias) if want_db else None
        db = None
        ws_bytes = L.faf_dropout_mean_bwd_colsum_workspace_bytes(R, 8, C) if want_db else 0
        if ws_bytes:
            tgt = acc if acc is not None else torch.empty(C, dtype=torch.float32, device=dout.device)
            ws = _workspace(ws_bytes, dout.device)
            timed("k_drop_mean_bwd", 4 * C * R * 9,
                  lambda: hip.check(L.faf_dropout_mean_bwd_colsum(_ptr(dout), R, 8, C, ctx.p, _ptr(ctx.seed), _ptr(dy), _ptr(tgt),
                                                                  1 if acc is not None else 0, _ptr(ws), ws_bytes,
                                                                  _stream(dout.device)), "faf_dropout_mean_bwd_colsum"))
            db = None if acc is not None else tgt
        else:
            timed("k_drop_mean_bwd", 4 * C * R * 9,
                  lambda: hip.check(L.faf_dropout_mean_bwd(_ptr(dout), R, 8, C, ctx.p, _ptr(ctx.seed), _ptr(dy),
                                                           _stream(dout.device)), "faf_dropout_mean_bwd"))
            if want_db:
                db = colsum(dy, into=acc)
        dh = mm_nn(dy, weight).view(ctx.shape) if ctx.needs_input_grad[0] else None
        dw = _linear_weight_grad(weight, None, None, dy, h2) if ctx.needs_input_grad[1] else None
        return dh, dw, db, None, None


def linear_dropout_mean_supported(h, weight) -> bool:
    return (h.is_cuda and h.dtype == torch.float32 and h.dim() >= 3 and h.shape[-2] == 8 and USE_X6
            and weight.shape[0] % 4 == 0 and h.shape[-1] % 4 == 0 and h.numel() // h.shape[-1] >= 8192)


def linear_dropout_mean(h, weight, bias, p: float, seed=None):
    """dropout_p(F.linear(h, weight, bias)).mean(-2) for h [..., 8, K]; see _LinearDropoutMean."""
    if torch.is_grad_enabled() and weight.requires_grad and weight.is_leaf and not hasattr(weight, "_eqh_transient"):
        LINEAR_PARAMS[id(weight)] = weight
    _note_acc(bias)
    return _LinearDropoutMean.apply(h, weight, bias, p, seed)


class _FramePre(torch.autograd.Function):
    """pre[e, f, :] = w3 (y[e] * s_f) + base[e] over the 8 sign frames, one pass each way (faf_frame_pre_*)."""

    @staticmethod
    def forward(ctx, y, w3, base):
        _require_gpu(y, "frame_pre")
        lead = y.shape[:-1]
        y2, w3c = _f32c(y).reshape(-1, 3), _f32c(w3)
        H = w3c.shape[0]
        base2 = _f32c(base.expand(*lead, H)).reshape(-1, H)
        E = y2.shape[0]
        out = torch.empty((E, 8, H), dtype=torch.float32, device=y.device)
        hip.check(hip.lib().faf_frame_pre_fwd(_ptr(y2), _ptr(w3c), _ptr(base2), E, H, _ptr(out), _stream(y.device)),
                  "faf_frame_pre_fwd")
        ctx.save_for_backward(y2, w3c)
        ctx.lead, ctx.base_shape = lead, base.shape
        return out.view(*lead, 8, H)

    @staticmethod
    def backward(ctx, dpre):
        y2, w3c = ctx.saved_tensors
        E, H = y2.shape[0], w3c.shape[0]
        dpre = _f32c(dpre).reshape(E, 8, H)
        dev = y2.device
        dy = torch.empty_like(y2)
        dbase = torch.empty((E, H), dtype=torch.float32, device=dev)
        dw3 = torch.empty_like(w3c)
        L = hip.lib()
        ws_bytes = L.faf_frame_pre_bwd_workspace_bytes(E, H)
        ws = _workspace(max(ws_bytes, 16), dev)
        hip.check(L.faf_frame_pre_bwd(_ptr(y2), _ptr(w3c), _ptr(dpre), E, H, _ptr(dy), _ptr(dbase), _ptr(dw3), 0, _ptr(ws),
                                      ws_bytes, _stream(dev)), "faf_frame_pre_bwd")
        dbase = dbase.view(*ctx.lead, H)
        if tuple(ctx.base_shape) != tuple(dbase.shape):          # base was broadcast (a bias vector): sum it back
            dbase = dbase.sum_to_size(ctx.base_shape)
        return dy.view(*ctx.lead, 3), dw3, dbase


class _FrameHidden(torch.autograd.Function):
    """LayerNorm(dropout_p(SiLU(a) * b)) of [a | b] = w3 (y * s_f) + base over the 8 sign frames: frame_pre, swiglu_dropout
    and the row LayerNorm in one launch each way (faf_frame_hidden_*); the [.., 8, 256] pre-activations never exist.
    ``base``: rows [..., 256], or (vector form) fc1's bias [256] with the optional K = 1 Linear extra [..., 1] * wx [256]
    evaluated inside the kernel.  ``w3`` may be the whole fc1.weight [256, 4] (then ``wx`` must be None: column 3 IS wx):
    the kernels read it in place and the gradient comes back as one [256, 4] tensor -- no column copies forward, no
    zero-fill / copy / add of two column gradients backward."""

    @staticmethod
    def forward(ctx, y, w3, base, extra, wx, gamma, beta, eps, p, seed, acc_params):
        _require_gpu(y, "frame_hidden")
        lead = y.shape[:-1]
        y2, w3c, gamma, beta = _f32c(y).reshape(-1, 3), _f32c(w3), _f32c(gamma), _f32c(beta)
        if w3c.shape[0] != 256 or gamma.numel() != 128 or w3c.shape[1] not in (3, 4):
            raise ValueError("frame_hidden: fc1 with 256 outputs expected")
        w_ld = w3c.shape[1]
        E = y2.shape[0]
        vec = base.dim() == 1
        if vec:
            base2, ld = _f32c(base), 0
            if w_ld == 4:
                if wx is not None:
                    raise ValueError("frame_hidden: with the whole [256, 4] weight, column 3 is wx")
                wxc = w3c.view(-1)[3:]                      # (a view of the same storage: w3 + 3)
            else:
                wxc = _f32c(wx) if wx is not None else torch.zeros(256, dtype=torch.float32, device=y.device)
            ex = _f32c(extra).reshape(-1) if extra is not None else None
            if ex is not None and ex.shape[0] != E:
                raise ValueError("frame_hidden: one extra value per point expected")
        else:
            if w_ld == 4:
                raise ValueError("frame_hidden: the whole [256, 4] weight goes with the vector form of base")
            base2, ld, wxc, ex = _f32c(base).reshape(-1, 256), 256, None, None
            if base2.shape[0] != E:
                raise ValueError("frame_hidden: one base row per point expected")
        seed = seed if (seed is not None and p > 0) else _dropout_seed(y.device, p)
        out = torch.empty((E, 8, 128), dtype=torch.float32, device=y.device)
        # algorithmic bytes: 12 B of coordinates (+ a [256] base row when it is per point) in, [8, 128] out
        timed("k_frame_hidden_fwd", E * (12 + (0 if ld == 0 else 1024) + 4096),
              lambda: hip.check(hip.lib().faf_frame_hidden_fwd(_ptr(y2), _ptr(w3c), _ptr(base2), ld, _ptr(ex), _ptr(wxc), _ptr(gamma),
                                                               _ptr(beta), E, float(p), _ptr(seed), float(eps), _ptr(out), w_ld,
                                                               _stream(y.device)), "faf_frame_hidden_fwd"))
        ctx.save_for_backward(y2, w3c, base2, gamma, ex, wxc)
        ctx.meta = (lead, ld, float(eps), float(p), seed, tuple(base.shape), None if extra is None else tuple(extra.shape),
                    wx is not None, w_ld)
        ctx.acc = acc_params
        return out.view(*lead, 8, 128)

    @staticmethod
    def backward(ctx, dhn):
        y2, w3c, base2, gamma, ex, wxc = ctx.saved_tensors
        lead, ld, eps, p, seed, base_shape, extra_shape, has_wx, w_ld = ctx.meta
        E = y2.shape[0]
        dev = y2.device
        vec = wxc is not None
        dhn = _f32c(dhn).reshape(E, 8, 128)
        dy = torch.empty_like(y2)
        dbase = torch.empty((256,) if vec else (E, 256), dtype=torch.float32, device=dev)
        dwx = torch.empty(256, dtype=torch.float32, device=dev) if vec else None
        dex = torch.empty(E, dtype=torch.float32, device=dev) if ex is not None else None
        dw3 = torch.empty((256, 3), dtype=torch.float32, device=dev)
        L = hip.lib()
        ws_bytes = L.faf_frame_hidden_bwd_workspace_bytes(E)
        ws = _workspace(max(ws_bytes, 16), dev)
        tg = [_acc_target(q) for q in ctx.acc]            # (gamma, beta)
        small = torch.empty((2, 128), dtype=torch.float32, device=dev)
        timed("k_frame_hidden_bwd", E * (12 + (0 if ld == 0 else 2048) + 4096),     # d hidden in, (d base out)
              lambda: hip.check(L.faf_frame_hidden_bwd(_ptr(y2), _ptr(w3c), _ptr(base2), ld, _ptr(ex), _ptr(wxc), _ptr(gamma),
                                                       _ptr(dhn), E, p, _ptr(seed), eps, _ptr(dy), _ptr(dbase), _ptr(dwx),
                                                       _ptr(dex), _ptr(dw3), _ptr(small[0]), _ptr(small[1]), 0, w_ld, _ptr(ws),
                                                       ws_bytes, _stream(dev)), "faf_frame_hidden_bwd"))
        dgam, dbet = _hand_out(list(small), tg)
        if not vec:
            dbase = dbase.view(base_shape)
        if w_ld == 4:
            dw3, dwx = torch.cat((dw3, dwx[:, None]), 1), None
        return (dy.view(*lead, 3), dw3, dbase, None if dex is None else dex.view(extra_shape), dwx if has_wx else None,
                dgam, dbet, None, None, None, None)


class _EdgeHidden(torch.autograd.Function):
    """LayerNorm(dropout_p(SiLU(a) * b)) of [a | b] = A[i] + B[nbr[i, k]] + Cf[i, k] on the kNN edges, one launch each way
    (faf_edge_hidden_*): the gathered / summed [N, K, 256] pre-activations and the [N, K, 128] gated values never exist."""

    @staticmethod
    def forward(ctx, A, B, Cf, nbr, csr_t, gamma, beta, eps, p, seed, acc_params):
        _require_gpu(A, "edge_hidden")
        A, B, gamma, beta = _f32c(A), _f32c(B), _f32c(gamma), _f32c(beta)
        N, K = nbr.shape
        Cf2 = _f32c(Cf).reshape(N * K, 256)
        if A.shape != (N, 256) or B.shape != (N, 256) or gamma.numel() != 128 or nbr.dtype != torch.int32:
            raise ValueError("edge_hidden: A, B [N, 256], Cf [N * K, 256], nbr int32 [N, K], gamma [128] expected")
        seed = seed if (seed is not None and p > 0) else _dropout_seed(A.device, p)
        out = torch.empty((N, K, 128), dtype=torch.float32, device=A.device)
        hip.check(hip.lib().faf_edge_hidden_fwd(_ptr(A), _ptr(B), _ptr(Cf2), _ptr(nbr), _ptr(gamma), _ptr(beta), N, K, float(p),
                                                _ptr(seed), float(eps), _ptr(out), _stream(A.device)), "faf_edge_hidden_fwd")
        ctx.save_for_backward(A, B, Cf2, gamma)
        ctx.meta = (nbr, csr_t, float(eps), float(p), seed, Cf.shape)
        ctx.acc = acc_params
        return out

    @staticmethod
    def backward(ctx, dhn):
        A, B, Cf2, gamma = ctx.saved_tensors
        nbr, csr_t, eps, p, seed, cf_shape = ctx.meta
        N, K = nbr.shape
        dev = A.device
        dhn = _f32c(dhn).reshape(N * K, 128)
        dpre = torch.empty_like(Cf2)
        dA = torch.empty_like(A)
        L = hip.lib()
        ws_bytes = L.faf_edge_hidden_bwd_workspace_bytes(N)
        ws = _workspace(max(ws_bytes, 16), dev)
        tg = [_acc_target(q) for q in ctx.acc]
        acc = all(t is not None for t in tg)
        small = None if acc else torch.empty((2, 128), dtype=torch.float32, device=dev)
        o = tg if acc else list(small)
        hip.check(L.faf_edge_hidden_bwd(_ptr(A), _ptr(B), _ptr(Cf2), _ptr(nbr), _ptr(gamma), _ptr(dhn), N, K, p, _ptr(seed), eps,
                                        _ptr(dpre), _ptr(dA), _ptr(o[0]), _ptr(o[1]), 1 if acc else 0, _ptr(ws), ws_bytes,
                                        _stream(dev)), "faf_edge_hidden_bwd")
        dB = _segment_reduce(dpre, csr_t.perm, csr_t.rowptr, None, csr_t.n_rows, False)     # rows of d pre by sender
        dgam, dbet = (None, None) if acc else _hand_out(list(small), tg)
        return dA, dB, dpre.view(cf_shape), None, None, dgam, dbet, None, None, None, None


def edge_hidden(A, B, Cf, nbr, csr_t: CSR, gamma, beta, eps: float = 1e-5, p: float = 0.0, seed=None):
    """LayerNorm(dropout_p(SiLU(a) * b)) with [a | b] = A[i] + B[nbr[i, k]] + Cf[i, k]: A, B [N, 256], Cf [N, K, 256], nbr int32
    [N, K], csr_t the transposed neighbour CSR (rows = senders), gamma / beta [128] (the PARAMETERS) -> [N, K, 128]."""
    _note_acc(gamma, beta)
    return _EdgeHidden.apply(A, B, Cf, nbr, csr_t, gamma, beta, eps, p, seed, (gamma, beta))


def _tall_colsum(t):
    """t.sum(0) for a tall [R, J <= 4] matrix.  torch reduces it with ONE 64-thread workgroup (600 us for the
    [248 k, 2] logit gradients of a Molecule3D batch); through a [R / 64, 64 J] view the first pass has 64 J independent
    columns and runs chip-wide."""
    R, J = t.shape
    if R >= 4096 and R % 64 == 0 and t.is_contiguous():
        return t.view(R // 64, 64 * J).sum(0).view(64, J).sum(0)
    return t.sum(0)


class _RowDot(torch.autograd.Function):
    """y = x @ U.T + bias for a FEW output columns (J <= 4), one pass over x each way (faf_rowdot_*).  With
    ``passthrough`` the node also returns x itself for x's OTHER consumer, and the backward adds that consumer's gradient
    in the same pass (no autograd add over the [rows, C] tensor)."""

    @staticmethod
    def forward(ctx, x, U, bias, passthrough):
        _require_gpu(x, "rowdot")
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        Uc = _f32c(U)
        R, C = x2.shape
        J = Uc.shape[0]
        y = torch.empty((R, J), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().faf_rowdot_fwd(_ptr(x2), _ptr(Uc), _ptr(_f32c(bias) if bias is not None else None), R, C, J,
                                           _ptr(y), _stream(x.device)), "faf_rowdot_fwd")
        ctx.save_for_backward(x2, Uc)
        ctx.shape, ctx.has_bias = x.shape, bias is not None
        ctx.set_materialize_grads(False)
        y = y.view(*x.shape[:-1], J)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x2, Uc = ctx.saved_tensors
        R, C = x2.shape
        J = Uc.shape[0]
        if dy is None:
            return dpass, None, None, None
        dy2 = _f32c(dy).reshape(R, J)
        add = _f32c(dpass).reshape(R, C) if dpass is not None else None
        dx = torch.empty_like(x2)
        dU = torch.empty_like(Uc)
        L = hip.lib()
        ws_bytes = L.faf_rowdot_bwd_workspace_bytes(R, C, J)
        ws = _workspace(max(ws_bytes, 16), x2.device)
        hip.check(L.faf_rowdot_bwd(_ptr(x2), _ptr(Uc), _ptr(dy2), _ptr(add), R, C, J, _ptr(dx), _ptr(dU), 0, _ptr(ws),
                                   ws_bytes, _stream(x2.device)), "faf_rowdot_bwd")
        db = _tall_colsum(dy2) if ctx.has_bias else None
        return dx.view(ctx.shape), dU, db, None


def rowdot(x, U, bias=None, passthrough: bool = False):
    """x [..., C] @ U [J, C].T + bias [J] -> [..., J] for J <= 4 (fp32, C % 4 == 0, C <= 1024); see _RowDot."""
    return _RowDot.apply(x, U, bias, passthrough)


def rowdot_supported(x, J: int) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024 and 1 <= J <= 4


class _GateRows(torch.autograd.Function):
    """out = res + xd * sigmoid(xd . w + b), xd = dropout_p(x): EdgeModule's gate with the dropout in front of it and the
    residual behind it, one pass each way (faf_gate_*).  ``lin_bias``: the bias PARAMETER of the Linear that produced x (called
    with ``bias_grad=False``): its gradient, the column sums of dx, rides the backward pass instead of reading dx again."""

    @staticmethod
    def forward(ctx, x, w, b, res, p, seed, acc_params, lin_bias=None):
        _require_gpu(x, "gate_rows")
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        wc, bc = _f32c(w).reshape(-1), _f32c(b).reshape(-1)
        R, C = x2.shape
        r2 = _f32c(res).reshape(R, C) if res is not None else None
        seed = seed if (seed is not None and p > 0) else _dropout_seed(x.device, p)
        out = torch.empty_like(x2)
        hip.check(hip.lib().faf_gate_fwd(_ptr(x2), _ptr(wc), _ptr(bc), _ptr(r2), R, C, float(p), _ptr(seed), _ptr(out),
                                         _stream(x.device)), "faf_gate_fwd")
        ctx.save_for_backward(x2, wc, bc)
        ctx.meta = (x.shape, float(p), seed, res is not None, w.shape, b.shape)
        ctx.acc, ctx.lin_bias = acc_params, lin_bias
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, wc, bc = ctx.saved_tensors
        shape, p, seed, has_res, w_shape, b_shape = ctx.meta
        R, C = x2.shape
        dout2 = _f32c(dout).reshape(R, C)
        dx = torch.empty_like(x2)
        L = hip.lib()
        ws_bytes = L.faf_gate_bwd_workspace_bytes(R, C)
        ws = _workspace(max(ws_bytes, 16), x2.device)
        tg = [_acc_target(q) for q in ctx.acc]
        acc = all(t is not None for t in tg)
        small = None if acc else torch.empty(C + 4, dtype=torch.float32, device=x2.device)
        dw_t = tg[0].reshape(-1) if acc else small[:C]
        db_t = tg[1].reshape(-1) if acc else small[C:C + 1]
        lb, dlb, lb_acc = ctx.lin_bias, None, None
        if lb is not None:
            lb_acc = _acc_target(lb)
            dlb = lb_acc if lb_acc is not None else torch.empty(C, dtype=torch.float32, device=x2.device)
        hip.check(L.faf_gate_bwd(_ptr(x2), _ptr(wc), _ptr(bc), _ptr(dout2), R, C, p, _ptr(seed), _ptr(dx), _ptr(dw_t),
                                 _ptr(db_t), 1 if acc else 0, _ptr(dlb), 1 if lb_acc is not None else 0, _ptr(ws), ws_bytes,
                                 _stream(x2.device)), "faf_gate_bwd")
        if acc:
            dw = db = None
        else:
            dw, db = _hand_out([small[:C].view(w_shape), small[C:C + 1].view(b_shape)], tg)
        return (dx.view(shape), dw, db, (dout if has_res else None), None, None, None,
                None if (lb is None or lb_acc is not None) else dlb)


def gate_rows(x, w, b, res=None, p: float = 0.0, seed=None, lin_bias=None):
    """res + dropout_p(x) * sigmoid(dropout_p(x) . w + b) over the last dim; w [C] (or [1, C]) and b [1] are the PARAMETERS;
    ``lin_bias``: see _GateRows."""
    _note_acc(w, b)
    if lin_bias is not None:
        _note_acc(lin_bias)
    return _GateRows.apply(x, w, b, res, p, seed, (w, b), lin_bias)


class _AttnSum(torch.autograd.Function):
    """out[n, c] = sum_m attn[n, c // D, m] * x[n, m, c] (faf_attn_sum_fwd / _bwd, csrc/faformer_ew.hip)."""

    @staticmethod
    def forward(ctx, attn, x):
        _require_gpu(x, "attn_sum")
        attn, x = _f32c(attn), _f32c(x)
        N, H, K = attn.shape
        C = x.shape[-1]
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().faf_attn_sum_fwd(_ptr(attn), _ptr(x), N, K, H, C // H, _ptr(out), _stream(x.device)),
                  "faf_attn_sum_fwd")
        ctx.save_for_backward(attn, x)
        return out

    @staticmethod
    def backward(ctx, dout):
        attn, x = ctx.saved_tensors
        dout = _f32c(dout)
        N, H, K = attn.shape
        C = x.shape[-1]
        dx, dattn = torch.empty_like(x), torch.empty_like(attn)
        hip.check(hip.lib().faf_attn_sum_bwd(_ptr(attn), _ptr(x), _ptr(dout), N, K, H, C // H, _ptr(dx), _ptr(dattn),
                                             _stream(x.device)), "faf_attn_sum_bwd")
        return dattn, dx


def attn_sum_supported(attn, x) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and attn.dim() == 3 and x.dim() == 3):
        return False
    n, h, k = attn.shape
    c = x.shape[-1]
    lpr, hl = c // 4, (c // h) // 4 if h else 0
    return (x.shape[0] == n and x.shape[1] == k and k <= 16 and c % (4 * h) == 0 and 1 <= lpr <= 64
            and lpr & (lpr - 1) == 0 and hl >= 1 and hl & (hl - 1) == 0)


def attn_sum(attn, x):
    """sum_m attn[n, h, m] * x[n, m, h*D:(h+1)*D] for attn [N, H, K], x [N, K, H*D] -> [N, H*D]."""
    return _AttnSum.apply(attn, x)


class _AttnGatherSum(torch.autograd.Function):
    """out[n, c] = sum_m attn[n, c // D, m] * x[nbr[n, m], c] for node rows x [NS, H*D] (faf_attn_gather_sum_*): the
    attention-weighted sum over the neighbours' values without the gathered [N, K, H*D] tensor either way."""

    @staticmethod
    def forward(ctx, attn, x, nbr, csr_t):
        _require_gpu(x, "attn_gather_sum")
        attn = _f32c(attn)
        x2, ld = _rows_ld(x)                     # (a column block of the qkv product is read in place)
        N, H, K = attn.shape
        C = x2.shape[1]
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().faf_attn_gather_sum_fwd(_ptr(attn), _ptr(x2), ld, _ptr(nbr), N, K, H, C // H, _ptr(out),
                                                    _stream(x.device)), "faf_attn_gather_sum_fwd")
        ctx.save_for_backward(attn, x2, nbr)
        ctx.csr, ctx.ld = csr_t, ld
        return out

    @staticmethod
    def backward(ctx, dout):
        attn, x2, nbr = ctx.saved_tensors
        csr = ctx.csr
        dout = _f32c(dout)
        N, H, K = attn.shape
        NS, C = x2.shape
        dattn = torch.empty_like(attn)
        dx = torch.empty((NS, C), dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[1] else None
        hip.check(hip.lib().faf_attn_gather_sum_bwd(_ptr(attn), _ptr(x2), ctx.ld, _ptr(nbr), _ptr(dout), _ptr(csr.rowptr),
                                                    _ptr(csr.perm), N, NS, K, H, C // H, _ptr(dattn), _ptr(dx),
                                                    _stream(dout.device)), "faf_attn_gather_sum_bwd")
        return dattn, dx, None, None


def attn_gather_sum_supported(h: int, x, nbr, csr_t) -> bool:
    """``h`` heads over node rows x [NS, C], neighbour list nbr [N, K] int32, transposed neighbour CSR."""
    if not (USE_GEOM and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and nbr.dim() == 2 and nbr.dtype == torch.int32):
        return False
    k, c = nbr.shape[1], x.shape[-1]
    lpr, hl = c // 4, (c // h) // 4 if h else 0
    return (nbr.is_contiguous() and csr_t.n_rows == x.shape[0] and k <= 16 and h >= 1 and c % (4 * h) == 0
            and 1 <= lpr <= 64 and lpr & (lpr - 1) == 0 and hl >= 1 and hl & (hl - 1) == 0)


def attn_gather_sum(attn, x, nbr, csr_t: CSR):
    """sum_m attn[n, h, m] * x[nbr[n, m], h*D:(h+1)*D] for attn [N, H, K], node rows x [NS, H*D], nbr [N, K] int32 and
    the CSR of the transposed neighbour graph (for dx); see _AttnGatherSum."""
    return _AttnGatherSum.apply(attn, x, nbr, csr_t)


def frame_pre(y, w3, base):
    """First Linear of FAFormer's frame-averaged MLP: [..., 3] x [H, 3] (+ base [..., H] or [H]) -> [..., 8, H] over
    the 8 sign frames in the order of fa_former_layer.py:70-84; H = 256."""
    return _FramePre.apply(y, w3, base)


def frame_hidden(y, w3, base, gamma, beta, eps: float = 1e-5, p: float = 0.0, seed=None, extra=None, wx=None):
    """LayerNorm(dropout_p(SiLU(a) * b)) with [a | b] = w3 (y * s_f) + base over the 8 sign frames of y [..., 3] ->
    [..., 8, 128]; w3 [256, 3], gamma / beta [128] (the PARAMETERS).  ``base``: rows [..., 256]; or fc1's bias [256], then
    the row of a point is bias + extra * wx (extra [..., 1], wx [256] = fc1.weight[:, 3]; both optional)."""
    _note_acc(gamma, beta)
    return _FrameHidden.apply(y, w3, base, extra, wx, gamma, beta, eps, p, seed, (gamma, beta))


def swiglu_dropout(pre, p: float = 0.0, seed=None):
    """dropout_p(silu(a) * b) for pre = [a | b] along the last dim (fp32, last dim % 8 == 0).  ``seed``: an int64 device
    tensor [1] to take the dropout decisions from (default: a fresh draw)."""
    return _SwigluDropout.apply(pre, p, seed)


def dropout_mean(x, p: float = 0.0, seed=None, bias=None):
    """dropout_p(x).mean(-2) for fp32 x [..., F, C] (C % 4 == 0).  ``bias``: see _DropoutMean (the producing Linear must
    have been called with ``bias_grad=False``)."""
    if bias is not None:
        _note_acc(bias)
    return _DropoutMean.apply(x, p, seed, bias)


def eigh3(cov):
    """Eigenvectors (columns, ascending eigenvalues) of a batch of symmetric 3x3 matrices [B,3,3];
    no gradient (the reference detaches the covariance, fa_former_layer.py:98-99)."""
    _require_gpu(cov, "eigh3")
    cov = _f32c(cov.detach())
    vec = torch.empty_like(cov)
    hip.check(hip.lib().geo_eigh3(_ptr(cov), cov.shape[0], None, _ptr(vec), _stream(cov.device)), "geo_eigh3")
    return vec


# ---------------------------------------------------------------------------------------------------------------
# The small geometric steps (csrc/faformer_geom.hip): one launch each way instead of 15-25 elementwise / reduction ones
# ---------------------------------------------------------------------------------------------------------------
USE_GEOM = not os.environ.get("EQH_NO_GEOM")   # tests / tools switch the fused geometry off to compare it with the torch expression of the same step


def geom_supported(x) -> bool:
    """centre_mix / cloud_frame take fp32 device tensors."""
    return bool(USE_GEOM) and x.is_cuda and x.dtype == torch.float32


def _moments_ws(n, device):
    nbytes = hip.lib().faf_moments_workspace_bytes(n)
    return _workspace(max(nbytes, 16), device), nbytes


class _CentreMix(torch.autograd.Function):
    """c * g + geo * (1 - g), g = sigmoid(logit), c = the (masked) centroid of geo accumulated in float64
    (faf_centre_mix_*).  ``zero_params`` take part in the reference's expression with a factor that cancels
    (faformer.MLPAttnEdgeAggregation): they receive an exactly-zero gradient instead of None."""

    @staticmethod
    def forward(ctx, geo, logit, row_mask, *zero_params):
        _require_gpu(geo, "centre_mix")
        g2, lg = _f32c(geo), _f32c(logit).reshape(-1)
        m = _f32c(row_mask).reshape(-1) if row_mask is not None else None
        n = g2.shape[0]
        out = torch.empty_like(g2)
        aux = torch.empty(4, dtype=torch.float32, device=geo.device)
        ws, nbytes = _moments_ws(n, geo.device)
        hip.check(hip.lib().faf_centre_mix_fwd(_ptr(g2), _ptr(lg), _ptr(m), n, _ptr(out), _ptr(aux), _ptr(ws), nbytes,
                                               _stream(geo.device)), "faf_centre_mix_fwd")
        ctx.save_for_backward(g2, lg, m, aux)
        ctx.zero, ctx.lshape = zero_params, logit.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        g2, lg, m, aux = ctx.saved_tensors
        n = g2.shape[0]
        dout = _f32c(dout)
        dgeo, dlogit = torch.empty_like(g2), torch.empty_like(lg)
        ws, nbytes = _moments_ws(n, g2.device)
        hip.check(hip.lib().faf_centre_mix_bwd(_ptr(dout), _ptr(g2), _ptr(lg), _ptr(m), _ptr(aux), n, _ptr(dgeo), _ptr(dlogit),
                                               _ptr(ws), nbytes, _stream(g2.device)), "faf_centre_mix_bwd")
        zeros = [None if _acc_target(q) is not None else torch.zeros_like(q) for q in ctx.zero]
        return (dgeo, dlogit.view(ctx.lshape), None, *zeros)


def centre_mix(geo, logit, row_mask=None, zero_params=()):
    """geo [N, 3], logit [N, 1] (the gate BEFORE its sigmoid), row_mask [N, 1] float or None -> [N, 3]; see _CentreMix."""
    _note_acc(*zero_params)
    return _CentreMix.apply(geo, logit, row_mask, *zero_params)


class _CloudFrame(torch.autograd.Function):
    """y = (x - c m) V for one point set x [N, 3]: c the masked centroid, V the eigenvectors of the masked covariance
    (float64 sums, no gradient through V) -- faf_cloud_frame_*."""

    @staticmethod
    def forward(ctx, x, row_mask):
        _require_gpu(x, "cloud_frame")
        x2 = _f32c(x)
        m = _f32c(row_mask).reshape(-1) if row_mask is not None else None
        n = x2.shape[0]
        y = torch.empty_like(x2)
        aux = torch.empty(13, dtype=torch.float32, device=x.device)
        ws, nbytes = _moments_ws(n, x.device)
        hip.check(hip.lib().faf_cloud_frame_fwd(_ptr(x2), _ptr(m), n, _ptr(y), _ptr(aux), _ptr(ws), nbytes, _stream(x.device)),
                  "faf_cloud_frame_fwd")
        ctx.save_for_backward(m, aux)
        return y

    @staticmethod
    def backward(ctx, dy):
        m, aux = ctx.saved_tensors
        dy = _f32c(dy)
        n = dy.shape[0]
        dx = torch.empty_like(dy)
        ws, nbytes = _moments_ws(n, dy.device)
        hip.check(hip.lib().faf_cloud_frame_bwd(_ptr(dy), _ptr(m), _ptr(aux), n, _ptr(dx), _ptr(ws), nbytes, _stream(dy.device)),
                  "faf_cloud_frame_bwd")
        return dx, None


def cloud_frame(x, row_mask=None):
    """Frame coordinates of the cloud x [N, 3] (fa_former_layer.py:86-113 on one point set); row_mask [N, 1] float or None."""
    return _CloudFrame.apply(x, row_mask)


class _EdgeFrame(torch.autograd.Function):
    """Per-atom frames over the neighbour offsets: (y [N, K, 3], d2 [N, K, 1]) from geo [N, 3], the neighbours' padded
    coordinates gj [N, K, 4] and the radius mask [N, K] (bool) -- faf_edge_frame_*."""

    @staticmethod
    def forward(ctx, geo, gj, mask):
        _require_gpu(geo, "edge_frame")
        g2, j2 = _f32c(geo), _f32c(gj)
        mk = mask.contiguous()
        n, k = j2.shape[0], j2.shape[1]
        y = torch.empty((n, k, 3), dtype=torch.float32, device=geo.device)
        d2 = torch.empty((n, k, 1), dtype=torch.float32, device=geo.device)
        V = torch.empty((n, 9), dtype=torch.float32, device=geo.device)
        hip.check(hip.lib().faf_edge_frame_fwd(_ptr(g2), _ptr(j2), _ptr(mk), n, k, _ptr(y), _ptr(d2), _ptr(V),
                                               _stream(geo.device)), "faf_edge_frame_fwd")
        ctx.save_for_backward(g2, j2, mk, V)
        ctx.set_materialize_grads(False)
        return y, d2

    @staticmethod
    def backward(ctx, dy, dd2):
        g2, j2, mk, V = ctx.saved_tensors
        if dy is None and dd2 is None:
            return None, None, None
        n, k = j2.shape[0], j2.shape[1]
        dy = _f32c(dy) if dy is not None else None
        dd2 = _f32c(dd2) if dd2 is not None else None
        dgeo, dgj = torch.empty_like(g2), torch.empty_like(j2)
        hip.check(hip.lib().faf_edge_frame_bwd(_ptr(g2), _ptr(j2), _ptr(mk), _ptr(V), _ptr(dy), _ptr(dd2), n, k, _ptr(dgeo),
                                               _ptr(dgj), _stream(g2.device)), "faf_edge_frame_bwd")
        return dgeo, dgj, None


def edge_frame_supported(geo, gj, mask) -> bool:
    return (USE_GEOM and geo.is_cuda and geo.dtype == torch.float32 and gj.dim() == 3 and gj.shape[-1] == 4
            and 1 <= gj.shape[1] <= 16 and mask.dtype == torch.bool)


def edge_frame(geo, gj, mask):
    """See _EdgeFrame (fa_former_layer.py:357-372 with create_frame :86-113 per atom)."""
    return _EdgeFrame.apply(geo, gj, mask)


class _AttnLogits(torch.autograd.Function):
    """dropout_p(softmax_K(masked(a_q[i] + a_k[j] + l_e))) -> [N, H, K] (faf_attn_logits_*); qa [N, 4] holds a_q then a_k
    of the H <= 2 heads, qan [N, K, 4] is qa gathered by neighbour."""

    @staticmethod
    def forward(ctx, qa, qan, le, mask, p, seed):
        _require_gpu(qa, "attn_logits")
        qa2, qn2, le2 = _f32c(qa), _f32c(qan), _f32c(le)
        mk = mask.contiguous()
        n, k, h = le2.shape
        seed = seed if (seed is not None and p > 0) else _dropout_seed(qa.device, p)
        attn = torch.empty((n, h, k), dtype=torch.float32, device=qa.device)
        prob = torch.empty_like(attn) if p > 0 else attn
        hip.check(hip.lib().faf_attn_logits_fwd(_ptr(qa2), _ptr(qn2), _ptr(le2), _ptr(mk), n, k, h, float(p), _ptr(seed),
                                                _ptr(prob), _ptr(attn), _stream(qa.device)), "faf_attn_logits_fwd")
        # (with p = 0 the saved probabilities ARE the output: keep a detached alias, not the output itself, so that the
        # node does not hold its own result -- see conv_stack._MergedConvStack)
        ctx.save_for_backward(prob.detach() if p > 0 else attn.detach().view_as(attn), mk)
        ctx.meta = (float(p), seed, qa.shape, qan.shape, le.shape)
        return attn

    @staticmethod
    def backward(ctx, dattn):
        prob, mk = ctx.saved_tensors
        p, seed, s_qa, s_qan, s_le = ctx.meta
        n, h, k = prob.shape
        dattn = _f32c(dattn)
        dqa = torch.empty((n, 4), dtype=torch.float32, device=prob.device)
        dqan = torch.empty((n, k, 4), dtype=torch.float32, device=prob.device)
        dle = torch.empty((n, k, h), dtype=torch.float32, device=prob.device)
        hip.check(hip.lib().faf_attn_logits_bwd(_ptr(prob), _ptr(dattn), _ptr(mk), n, k, h, p, _ptr(seed), _ptr(dqa), _ptr(dqan),
                                                _ptr(dle), _stream(prob.device)), "faf_attn_logits_bwd")
        return dqa.view(s_qa), dqan.view(s_qan), dle.view(s_le), None, None, None


def attn_logits_supported(qa, le, mask) -> bool:
    return (USE_GEOM and qa.is_cuda and qa.dtype == torch.float32 and qa.shape[-1] == 4 and le.dim() == 3
            and 1 <= le.shape[1] <= 16 and 1 <= le.shape[2] <= 2 and mask.dtype == torch.bool)


def attn_logits(qa, qan, le, mask, p: float = 0.0, seed=None):
    """See _AttnLogits (fa_former_layer.py:483-496)."""
    return _AttnLogits.apply(qa, qan, le, mask, p, seed)


class _EdgeLogitWeights(torch.autograd.Function):
    """(u [h, de], c [h]) = the Linear(deh, 1) ``we`` folded into the QUERY rows (the first de) of the edge Linear
    W [2 de, de], b [2 de] (faf_edge_logit_weights_*): one launch each way; the gradients go straight into the
    parameters' accumulators when they have them (W's value rows are written by ops.linear(..., rows=...))."""

    @staticmethod
    def forward(ctx, W, b, we, h):
        _require_gpu(W, "edge_logit_weights")
        Wc, bc, wc = _f32c(W), _f32c(b), _f32c(we).reshape(-1)
        de, deh = Wc.shape[1], wc.numel()
        u = torch.empty((h, de), dtype=torch.float32, device=W.device)
        c = torch.empty(h, dtype=torch.float32, device=W.device)
        hip.check(hip.lib().faf_edge_logit_weights_fwd(_ptr(Wc), de, _ptr(bc), _ptr(wc), h, deh, de, _ptr(u), _ptr(c),
                                                       _stream(W.device)), "faf_edge_logit_weights_fwd")
        ctx.save_for_backward(Wc, bc, wc)
        ctx.params, ctx.h, ctx.we_shape = (W, b, we), h, we.shape
        ctx.set_materialize_grads(False)
        return u, c

    @staticmethod
    def backward(ctx, du, dc):
        Wc, bc, wc = ctx.saved_tensors
        if du is None and dc is None:
            return None, None, None, None
        h, de, deh = ctx.h, Wc.shape[1], wc.numel()
        du = _f32c(du) if du is not None else None
        dc = _f32c(dc) if dc is not None else None
        tg = [_acc_target(q) for q in ctx.params]
        dW = tg[0] if tg[0] is not None else torch.zeros_like(Wc)
        db = tg[1] if tg[1] is not None else torch.zeros_like(bc)
        dwe = tg[2].reshape(-1) if tg[2] is not None else torch.empty_like(wc)
        hip.check(hip.lib().faf_edge_logit_weights_bwd(_ptr(Wc), de, _ptr(bc), _ptr(wc), _ptr(du), _ptr(dc), h, deh, de,
                                                       _ptr(dW), de, _ptr(db), _ptr(dwe), int(tg[0] is not None),
                                                       int(tg[1] is not None), int(tg[2] is not None), _stream(Wc.device)),
                  "faf_edge_logit_weights_bwd")
        return (None if tg[0] is not None else dW, None if tg[1] is not None else db,
                None if tg[2] is not None else dwe.view(ctx.we_shape), None)


def edge_logit_weights(W, b, we, h: int):
    """See _EdgeLogitWeights; W, b, we are the PARAMETERS (layernorm_qkv_edge's Linear, edge_attn's weight)."""
    if torch.is_grad_enabled() and W.requires_grad and W.is_leaf and not hasattr(W, "_eqh_transient"):
        LINEAR_PARAMS[id(W)] = W
    _note_acc(b, we)
    return _EdgeLogitWeights.apply(W, b, we, h)


class _LnRowDot(torch.autograd.Function):
    """(xe, le, x) = (LayerNorm(x), LayerNorm(x) @ U.T + cb, x itself) in one pass each way (faf_ln_rowdot_*): the
    LayerNorm in front of FAFormer's edge Linear with the per-head edge logits riding along.  The third output is x for
    its OTHER consumer (the residual of the edge update), whose gradient the backward adds in the same pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, U, cb, eps, acc_params):
        _require_gpu(x, "ln_rowdot")
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        g, b, Uc = _f32c(gamma), _f32c(beta), _f32c(U)
        R, C = x2.shape
        J = Uc.shape[0]
        out = torch.empty_like(x2)
        le = torch.empty((R, J), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().faf_ln_rowdot_fwd(_ptr(x2), _ptr(g), _ptr(b), _ptr(Uc), _ptr(_f32c(cb) if cb is not None else None),
                                              R, C, J, float(eps), _ptr(out), _ptr(le), _stream(x.device)), "faf_ln_rowdot_fwd")
        ctx.save_for_backward(x2, g, b, Uc)
        ctx.meta = (x.shape, float(eps), cb is not None)
        ctx.acc = acc_params
        ctx.set_materialize_grads(False)
        return out.view(x.shape), le.view(*x.shape[:-1], J), x.view_as(x)

    @staticmethod
    def backward(ctx, dxe, dle, dpass):
        x2, g, b, Uc = ctx.saved_tensors
        shape, eps, has_cb = ctx.meta
        R, C = x2.shape
        J = Uc.shape[0]
        if dxe is None and dle is None:
            return dpass, None, None, None, None, None, None
        dy, ld = _rows_ld(dxe.reshape(R, C)) if dxe is not None else (None, C)
        dl = _f32c(dle).reshape(R, J) if dle is not None else None
        add = _f32c(dpass).reshape(R, C) if dpass is not None else None
        dx = torch.empty_like(x2)
        dU = torch.empty_like(Uc)
        L = hip.lib()
        ws_bytes = L.faf_ln_rowdot_bwd_workspace_bytes(R, C, J)
        ws = _workspace(max(ws_bytes, 16), x2.device)
        tg = [_acc_target(q) for q in ctx.acc]
        acc = all(t is not None for t in tg)
        small = None if acc else torch.empty((2, C), dtype=torch.float32, device=x2.device)
        hip.check(L.faf_ln_rowdot_bwd(_ptr(x2), _ptr(g), _ptr(b), _ptr(Uc), _ptr(dy), ld, _ptr(dl), _ptr(add), R, C, J, eps,
                                      _ptr(dx), _ptr(tg[0] if acc else small[0]), _ptr(tg[1] if acc else small[1]),
                                      1 if acc else 0, _ptr(dU), _ptr(ws), ws_bytes, _stream(x2.device)), "faf_ln_rowdot_bwd")
        dgam, dbet = (None, None) if acc else _hand_out(list(small), tg)
        dcb = _tall_colsum(dl) if (has_cb and dl is not None) else None
        return dx.view(shape), dgam, dbet, dU, dcb, None, None


def ln_rowdot_supported(x, J: int) -> bool:
    return USE_GEOM and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024 and 1 <= J <= 2


def ln_rowdot(x, gamma, beta, U, cb=None, eps: float = 1e-5):
    """(LayerNorm(x), LayerNorm(x) @ U.T + cb, x) for x [..., C], U [J <= 2, C]; gamma / beta are the PARAMETERS; see
    _LnRowDot."""
    _note_acc(gamma, beta)
    return _LnRowDot.apply(x, gamma, beta, U, cb, eps, (gamma, beta))


class _DropoutAdd(torch.autograd.Function):
    """res + dropout_p(x) in one pass (faf_dropout_add; no mask tensor: the backward recomputes the keep decisions)."""

    @staticmethod
    def forward(ctx, x, res, p, seed):
        _require_gpu(x, "dropout_add")
        x2 = _f32c(x)
        r2 = _f32c(res) if res is not None else None
        seed = seed if (seed is not None and p > 0) else _dropout_seed(x.device, p)
        out = torch.empty_like(x2)
        hip.check(hip.lib().faf_dropout_add(_ptr(x2), _ptr(r2), x2.numel(), float(p), _ptr(seed), _ptr(out), _stream(x.device)),
                  "faf_dropout_add")
        ctx.p, ctx.seed, ctx.has_res = float(p), seed, res is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        dx = torch.empty_like(dout)
        hip.check(hip.lib().faf_dropout_add(_ptr(dout), None, dout.numel(), ctx.p, _ptr(ctx.seed), _ptr(dx), _stream(dout.device)),
                  "faf_dropout_add")
        return dx, (dout if ctx.has_res else None), None, None


def dropout_add_supported(x, res) -> bool:
    return (USE_GEOM and x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0
            and (res is None or (res.shape == x.shape and res.dtype == torch.float32)))


def dropout_add(x, res=None, p: float = 0.0, seed=None):
    """res + dropout_p(x) (res optional); fp32, numel % 4 == 0."""
    return _DropoutAdd.apply(x, res, p, seed)
