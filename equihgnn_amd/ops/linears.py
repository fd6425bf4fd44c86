"""nn.Linear-shaped autograd nodes (one, two of one input, with addend), merged consecutive Linears, matmul fans.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ._base import (LINEAR_PARAMS, _DEFER, _acc_target, _f32c, _note_acc)
from .products import (USE_X6, X6_WGRAD_OUTPUTS, X6_WGRAD_ROWS, gemm, gemm_out_ok, gemm_supported, mm_nn, mm_nt, small_mm_batch)
from .grads import (_linear_weight_grad, _merged_acc, _wgrad_deferred, _wgrad_ok, colsum, wgrad)


class _MergedWeight(torch.autograd.Function):
    """Wc = A[:, a0:a1] @ B and bc = A[:, a0:a1] @ bb + bo: the weight and bias of TWO consecutive Linears with only a
    linear map between them (y = A_blk (B x + bb) + bo), formed at weight level ([C x C] x [C x C]: 33 MFLOP instead of
    a [rows x C] x [C x C] product per application and per direction).  A, B, bb, bo are the PARAMETERS.  Backward:
    dA_blk = dWc B^T + dbc bb^T, dB = A_blk^T dWc, dbb = A_blk^T dbc, dbo = dbc -- added to the parameters' persistent
    accumulators when they have them."""

    @staticmethod
    def forward(ctx, A, B, bb, bo, a0, a1):
        blk = A if a0 is None else A[:, a0:a1]
        ctx.save_for_backward(A, B, bb, bo)
        ctx.set_materialize_grads(False)
        ctx.cols = (a0, a1)
        wc = blk @ B
        if bb is None:
            return wc, None
        bc = torch.mv(blk, bb)
        if bo is not None:
            bc = bc + bo
        return wc, bc

    @staticmethod
    def backward(ctx, dwc, dbc):
        A, B, bb, bo = ctx.saved_tensors
        a0, a1 = ctx.cols
        blk = A if a0 is None else A[:, a0:a1]
        ga, gb = _acc_target(A), _acc_target(B)
        if dwc is None and dbc is None:
            return None, None, None, None, None, None
        if dwc is None:
            dwc = torch.zeros((blk.shape[0], B.shape[1]), dtype=B.dtype, device=B.device)
        # A
        if ga is not None:
            tgt = ga if a0 is None else ga[:, a0:a1]
            tgt.addmm_(dwc, B.t())
            if dbc is not None and bb is not None:
                tgt.addr_(dbc, bb)
            dA = None
        else:
            dblk = dwc @ B.t()
            if dbc is not None and bb is not None:
                dblk = dblk.addr_(dbc, bb)
            if a0 is None:
                dA = dblk
            else:
                dA = torch.zeros_like(A)
                dA[:, a0:a1] = dblk
        # B
        if gb is not None:
            gb.addmm_(blk.t(), dwc)
            dB = None
        else:
            dB = blk.t() @ dwc
        dbb = dbo = None
        if bb is not None and dbc is not None:
            t = _acc_target(bb)
            if t is not None:
                t.addmv_(blk.t(), dbc)
            else:
                dbb = torch.mv(blk.t(), dbc)
            if bo is not None:
                t = _acc_target(bo)
                if t is not None:
                    t.add_(dbc)
                else:
                    dbo = dbc
        return dA, dB, dbb, dbo, None, None


class _MergedWeights(torch.autograd.Function):
    """Several merged weights (see _MergedWeight) in ONE launch each way (hg_small_mm_batch): forward
    Wc_i = A_i[:, cols_i] @ B_i, bc_i = A_i[:, cols_i] @ bb_i + bo_i; backward dA_i, dB_i, dbb_i, dbo_i straight into the
    parameters' accumulators where they have them.  apply(cols, A_0, B_0, bb_0, bo_0, A_1, ...) -> (Wc_0, bc_0, Wc_1, ...)
    (bc_i is None without bb_i)."""

    @staticmethod
    def forward(ctx, cols, *ts):
        n = len(ts) // 4
        ctx.cols, ctx.n = cols, n
        ctx.save_for_backward(*[t for t in ts if t is not None])
        ctx.present = [t is not None for t in ts]
        ctx.set_materialize_grads(False)
        outs, probs = [], []
        for i in range(n):
            A, B, bb, bo = ts[4 * i:4 * i + 4]
            a0, a1 = cols[i] if cols[i] is not None else (None, None)
            blk = A if a0 is None else A[:, a0:a1]
            wc = torch.empty((blk.shape[0], B.shape[1]), dtype=torch.float32, device=A.device)
            bc = torch.empty(blk.shape[0], dtype=torch.float32, device=A.device) if bb is not None else None
            pr = dict(a=blk, b=B, c=wc)
            if bb is not None:
                pr.update(x=bb, z=bo, y=bc)
            probs.append(pr)
            outs.extend((wc, bc))
        small_mm_batch(probs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        it = iter(ctx.saved_tensors)
        ts = [next(it) if p else None for p in ctx.present]
        grads, probs = [None], []
        for i in range(ctx.n):
            A, B, bb, bo = ts[4 * i:4 * i + 4]
            dwc, dbc = douts[2 * i], douts[2 * i + 1]
            a0, a1 = ctx.cols[i] if ctx.cols[i] is not None else (None, None)
            blk = A if a0 is None else A[:, a0:a1]
            if dwc is None and dbc is None:
                grads.extend((None, None, None, None))
                continue
            if dwc is None:
                dwc = torch.zeros((blk.shape[0], B.shape[1]), dtype=B.dtype, device=B.device)
            dwc = dwc if dwc.stride(1) == 1 else dwc.contiguous()
            use_b = bb is not None and dbc is not None
            if use_b:
                dbc = dbc.contiguous()
            ga, gb = _acc_target(A), _acc_target(B)
            # dA[:, cols] (+)= dWc B^T (+ dbc (x) bb);  dbo += dbc rides along
            if ga is not None:
                tgt, dA = (ga if a0 is None else ga[:, a0:a1]), None
            else:
                dA = torch.zeros_like(A)
                tgt = dA if a0 is None else dA[:, a0:a1]
            p1 = dict(a=dwc, b=B, tb=True, c=tgt, accumulate=ga is not None)
            dbo = None
            if use_b:
                p1.update(u=dbc, v=bb)
                if bo is not None:
                    t = _acc_target(bo)
                    if t is not None:
                        p1.update(w=t)
                    else:
                        dbo = dbc
            # dB (+)= A_blk^T dWc;  dbb (+)= A_blk^T dbc rides along
            dB = None if gb is not None else torch.empty_like(B)
            p2 = dict(a=blk, ta=True, b=dwc, c=gb if gb is not None else dB, accumulate=gb is not None)
            dbb = None
            if use_b:
                t = _acc_target(bb)
                if t is None:
                    dbb = torch.empty_like(bb)
                p2.update(x=dbc, y=t if t is not None else dbb, acc_y=t is not None)
            probs.extend((p1, p2))
            grads.extend((dA, dB, dbb, dbo))
        for j in range(0, len(probs), 8):
            small_mm_batch(probs[j:j + 8])
        return tuple(grads)


def merged_weights(items):
    """[(Wc, bc)] for items = [(A, B, bb, bo, cols)]: Wc = A[:, cols] @ B, bc = A[:, cols] @ bb + bo (None without bb), all in
    one launch (and one for the whole backward).  While gradient reductions are deferred (graphed trainer) the results
    carry accumulators of their own, so the weight gradients of the Linears that use them join the batched launch of
    defer_flush like any parameter's, and defer_flush then back-propagates them to the parameters."""
    flat, cols = [], []
    for A, B, bb, bo, c in items:
        if torch.is_grad_enabled():
            for w in (A, B):
                if w.requires_grad and w.is_leaf:
                    LINEAR_PARAMS[id(w)] = w
            _note_acc(bb, bo)
        flat.extend((A, B, bb, bo))
        cols.append(tuple(c) if c is not None else None)
    res = _MergedWeights.apply(tuple(cols), *flat)
    pairs = [(res[2 * i], res[2 * i + 1]) for i in range(len(items))]
    if not (_DEFER["active"] and torch.is_grad_enabled() and any(w.requires_grad for w, _ in pairs)):
        return pairs
    outs, accs, leaves = [], [], []
    for wc, bc in pairs:
        # the Linears see detached leaves with accumulators; the weight-level products stay out of the main backward
        O, I = wc.shape
        ld = (I + 3) // 4 * 4
        acc = _merged_acc((O + 1, max(ld, O)), wc.device)   # rows 0..O-1: dWc; row O: dbc
        outs.append(wc)
        accs.append(acc[:O, :I])
        wl = wc.detach().requires_grad_()
        wl._eqh_transient = True
        wl._eqh_gbuf = acc[:O, :I]
        bl = None
        if bc is not None:
            bl = bc.detach().requires_grad_()
            bl._eqh_transient = True
            bl._eqh_gbuf = acc[O, :O]
            outs.append(bc)
            accs.append(acc[O, :O])
        leaves.append((wl, bl))
    _DEFER["merged"].append((outs, accs))
    return leaves


def merged_weight(A, B, bb=None, bo=None, cols=None):
    """(Wc, bc) with Wc = A[:, cols] @ B and bc = A[:, cols] @ bb + bo (bc None without bb) for parameters A [O, *],
    B [K, I], bb [K], bo [O]: see _MergedWeight.  While gradient reductions are deferred (graphed trainer) the two
    results carry accumulators of their own, so the weight gradients of the Linears that use them join the batched
    launch of defer_flush like any parameter's, and defer_flush then back-propagates them to A, B, bb, bo."""
    if torch.is_grad_enabled():
        for w in (A, B):
            if w.requires_grad and w.is_leaf:
                LINEAR_PARAMS[id(w)] = w
        _note_acc(bb, bo)
    a0, a1 = cols if cols is not None else (None, None)
    wc, bc = _MergedWeight.apply(A, B, bb, bo, a0, a1)
    if _DEFER["active"] and torch.is_grad_enabled() and wc.requires_grad:
        # the Linears see detached leaves with accumulators; the weight-level product stays out of the main backward
        O, I = wc.shape
        ld = (I + 3) // 4 * 4
        acc = _merged_acc((O + 1, max(ld, O)), wc.device)   # rows 0..O-1: dWc; row O: dbc
        outs, accs = [wc], [acc[:O, :I]]
        wl = wc.detach().requires_grad_()
        wl._eqh_transient = True
        wl._eqh_gbuf = acc[:O, :I]
        bl = None
        if bc is not None:
            bl = bc.detach().requires_grad_()
            bl._eqh_transient = True
            bl._eqh_gbuf = acc[O, :O]
            outs.append(bc)
            accs.append(acc[O, :O])
        _DEFER["merged"].append((outs, accs))
        return wl, bl
    return wc, bc


class _Linear(torch.autograd.Function):
    """y = x @ W[r0:r1, c0:c1].T (+ bias[r0:r1]): a library GEMM whose WEIGHT gradient, when the parameter
    carries a persistent accumulator (``param._eqh_gbuf``, same shape as the parameter), is
    accumulated by the GEMM itself (addmm_ with beta = 1 into the accumulator's block)
    instead of being materialised and then added by autograd — shared weights (the conv layer is
    applied L times) and column-split weights (W·cat(a,b) = Wa·a + Wb·b) cost no extra kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, c0, c1, r0=None, r1=None, relu=False, bias_grad=True):
        w = weight if c0 is None else weight[:, c0:c1]
        b = bias
        if r0 is not None:
            w = w[r0:r1]
            b = bias[r0:r1] if bias is not None else None
        ctx.cols, ctx.rows = (c0, c1), (r0, r1)
        ctx.has_bias = bias is not None and bool(bias_grad)      # (bias_grad False: another node produces the bias gradient)
        ctx.bias_param = bias
        ctx.relu = bool(relu)
        if relu:    # relu(x W^T + b) with the activation in the GEMM's epilogue (2-D x, bias given: checked by linear())
            y = mm_nt(x, w, bias=b, relu=True)
            ctx.save_for_backward(x, weight, y)
            return y
        ctx.save_for_backward(x, weight)
        if x.dim() == 2:
            return mm_nt(x, w, bias=b)
        if x.is_contiguous() and x.is_cuda:
            return mm_nt(x.reshape(-1, x.shape[-1]), w, bias=b).view(*x.shape[:-1], w.shape[0])
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        if ctx.relu:
            x, weight, y = ctx.saved_tensors
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)
        else:
            x, weight = ctx.saved_tensors
        c0, c1 = ctx.cols
        r0, r1 = ctx.rows
        w = weight if c0 is None else weight[:, c0:c1]
        if r0 is not None:
            w = w[r0:r1]
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = mm_nn(dy2, w).view(*dy.shape[:-1], w.shape[1]) if dy2.is_cuda else dy @ w
        dw = _linear_weight_grad(weight, c0, c1, dy2, x2, r0, r1) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            acc = _acc_target(ctx.bias_param)
            if r0 is None:
                db = colsum(dy2, into=acc)   # None when accumulated in place
            elif acc is not None:
                colsum(dy2, into=acc[r0:r1])
            else:
                db = torch.zeros_like(ctx.bias_param)
                db[r0:r1] = colsum(dy2)
        return dx, dw, db, None, None, None, None, None, None


class _Linear2(torch.autograd.Function):
    """(x @ Wa[:, a0:a1].T, x @ Wb[:, b0:b1].T) for two bias-free Linears of the SAME input: the backward pass
    receives both output gradients together, so the input gradient is one GEMM plus one accumulating GEMM
    (addmm_, beta = 1) instead of two GEMMs and an add kernel.  Weight gradients as in _Linear."""

    @staticmethod
    def forward(ctx, x, wa, a0, a1, wb, b0, b1):
        ctx.save_for_backward(x, wa, wb)
        ctx.cols = (a0, a1, b0, b1)
        ctx.set_materialize_grads(False)
        wa_, wb_ = (wa if a0 is None else wa[:, a0:a1]), (wb if b0 is None else wb[:, b0:b1])
        if x.dim() == 2 and x.is_cuda:
            return mm_nt(x, wa_), mm_nt(x, wb_)
        return F.linear(x, wa_), F.linear(x, wb_)

    @staticmethod
    def backward(ctx, dya, dyb):
        x, wa, wb = ctx.saved_tensors
        a0, a1, b0, b1 = ctx.cols
        x2 = x.reshape(-1, x.shape[-1])
        dx, dwa, dwb = None, None, None
        for dy, w, c0, c1, slot in ((dya, wa, a0, a1, 1), (dyb, wb, b0, b1, 4)):
            if dy is None:
                continue
            ws = w if c0 is None else w[:, c0:c1]
            dy2 = dy.reshape(-1, dy.shape[-1])
            if ctx.needs_input_grad[0]:
                dx = mm_nn(dy2, ws) if dx is None else mm_nn(dy2, ws, d=dx, out=dx)
            if ctx.needs_input_grad[slot]:
                g = _linear_weight_grad(w, c0, c1, dy2, x2)
                if slot == 1:
                    dwa = g
                else:
                    dwb = g
        if dx is not None:
            dx = dx.view_as(x)
        return dx, dwa, None, None, dwb, None, None


class _LinearAddC(torch.autograd.Function):
    """y = scale * (x @ W.T) + c in ONE GEMM launch (beta = 1 epilogue); ``c`` carries whatever is
    added after the Linear (residual mix, row-masked bias).  Weight gradient as in _Linear; the two
    backward GEMMs take ``scale`` as their alpha, so no scaling kernel runs either way."""

    @staticmethod
    def forward(ctx, x, weight, c, scale, fan=None):
        ctx.save_for_backward(x, weight)
        ctx.scale = float(scale)
        ctx.fan = fan
        return mm_nt(x, weight, d=c, alpha=ctx.scale, beta=1.0)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        a = ctx.scale
        dx = mm_nn(dy, weight, alpha=a) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            gbuf = getattr(weight, "_eqh_gbuf", None)
            if gbuf is not None and _wgrad_deferred(dy, x, a, gbuf):
                pass
            elif gbuf is not None and _wgrad_ok(dy, x):
                wgrad(dy, x, a, into=gbuf)
            elif gbuf is not None:
                gbuf.addmm_(dy.t(), x, alpha=a)
            elif _wgrad_ok(dy, x):
                dw = wgrad(dy, x, a)
            else:
                dw = torch.addmm(weight, dy.t(), x, beta=0.0, alpha=a)
        # (with a GradFan the gradient of c -- dy itself -- is summed by the LayerNorm backward that produced dy)
        return dx, dw, (dy if ctx.needs_input_grad[2] and ctx.fan is None else None), None, None


def linear(x, weight, bias=None, cols=None, rows=None, relu=False, bias_grad=True):
    """F.linear(x, weight[rows[0]:rows[1], cols[0]:cols[1]], bias[rows[0]:rows[1]]) through _Linear (``weight`` and
    ``bias`` are the PARAMETERS, not slices of them, so that their gradient accumulators can be found).  ``relu``:
    relu(...) with the activation in the GEMM epilogue (2-D fp32 x on the GPU with a bias; else a separate kernel).
    ``bias_grad=False``: the bias gradient is produced by the consumer of the output (ops.dropout_mean(..., bias=...))."""
    if relu and not (x.is_cuda and x.dim() == 2 and bias is not None and x.dtype == torch.float32):
        return torch.relu(linear(x, weight, bias, cols, rows))
    if torch.is_grad_enabled() and weight.requires_grad and weight.is_leaf and not hasattr(weight, "_eqh_transient"):
        LINEAR_PARAMS[id(weight)] = weight
    _note_acc(bias)
    c0, c1 = cols if cols is not None else (None, None)
    r0, r1 = rows if rows is not None else (None, None)
    return _Linear.apply(x, weight, bias, c0, c1, r0, r1, relu, bias_grad)


def linear2(x, wa, cols_a, wb, cols_b):
    """(F.linear(x, wa[:, cols_a]), F.linear(x, wb[:, cols_b])) for two bias-free Linears of one input (the
    PARAMETERS are passed, not slices); see _Linear2."""
    if torch.is_grad_enabled():
        for w in (wa, wb):
            if w.requires_grad and w.is_leaf:
                LINEAR_PARAMS[id(w)] = w
    a0, a1 = cols_a if cols_a is not None else (None, None)
    b0, b1 = cols_b if cols_b is not None else (None, None)
    return _Linear2.apply(x, wa, a0, a1, wb, b0, b1)


def linear_add(x, weight, c, scale: float = 1.0, fan=None):
    """scale * F.linear(x, weight) + c with the addition done by the GEMM epilogue (2-D x, c).  ``fan``: the GradFan of
    ``c`` when the result goes straight into bias_relu_ln(..., fan=fan), whose backward then collects c's gradient."""
    if torch.is_grad_enabled() and weight.requires_grad and weight.is_leaf and not hasattr(weight, "_eqh_transient"):
        LINEAR_PARAMS[id(weight)] = weight
    return _LinearAddC.apply(x, weight, c, scale, fan)


class _MatmulFan(torch.autograd.Function):
    """ys[i] = x @ Ws[i] for weights stored [in, out_i] (the FiberLinear layout of equiformer_layer.py:168-191): ONE autograd
    node for all products of the same input, so that its gradient is one GEMM plus accumulating GEMMs (no add kernels), and
    the weight gradients x^T dy_i of leaf weights with a persistent accumulator join the batched launch of defer_flush."""

    @staticmethod
    def forward(ctx, x, *Ws):
        ctx.save_for_backward(x, *Ws)
        ctx.set_materialize_grads(False)
        if x.is_cuda and x.is_contiguous():
            x2 = x.reshape(-1, x.shape[-1])
            return tuple(mm_nn(x2, W).view(*x.shape[:-1], W.shape[1]) for W in Ws)
        return tuple(x @ W for W in Ws)

    @staticmethod
    def backward(ctx, *dys):
        x, *Ws = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1])
        dx = None
        dWs = []
        for k, (W, dy) in enumerate(zip(Ws, dys)):
            if dy is None:
                dWs.append(None)
                continue
            dy2 = dy.reshape(-1, dy.shape[-1])
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = mm_nt(dy2, W) if dy2.is_cuda else dy2 @ W.t()
                elif dy2.is_cuda:
                    dx = mm_nt(dy2, W, d=dx)
                else:
                    dx.addmm_(dy2, W.t())
            if not ctx.needs_input_grad[1 + k]:
                dWs.append(None)
                continue
            tgt = _acc_target(W)
            if tgt is not None:
                if _wgrad_deferred(x2, dy2, 1.0, tgt):          # into [in, out] += x2^T dy2
                    pass
                elif (USE_X6 and x2.is_cuda and x2.shape[0] >= X6_WGRAD_ROWS and gemm_supported(x2, dy2, True, False)
                      and gemm_out_ok(tgt)):
                    # (widths the batched kernel does not take: the [104 x 104] attention mix over 39 k edge rows, split-K)
                    xa, da = _f32c(x2), _f32c(dy2)
                    if _DEFER["active"]:
                        _DEFER["keep"].extend((xa, da))
                    gemm(xa, da, trans_a=True, trans_b=False, d=tgt, out=tgt)
                else:
                    tgt.addmm_(x2.t(), dy2)
                dWs.append(None)
            elif (USE_X6 and x2.is_cuda and x2.shape[0] >= 1024 and x2.shape[1] * dy2.shape[1] >= X6_WGRAD_OUTPUTS
                  and gemm_supported(x2, dy2, True, False)):
                # a wide transient weight (Equiformer's [256 x 16384] radial node weights): 128 x 128 tiles on the x6 kernel,
                # 114 against the library's 153 us
                dWs.append(gemm(_f32c(x2), _f32c(dy2), trans_a=True, trans_b=False))
            else:
                dWs.append(x2.t() @ dy2)
        return (dx.view(x.shape) if dx is not None else None, *dWs)


def matmul_fan(x, *Ws):
    """(x @ W for W in Ws), weights [in, out_i]; see _MatmulFan.  Leaf weights are registered for persistent accumulators."""
    if torch.is_grad_enabled():
        for W in Ws:
            if W.requires_grad and W.is_leaf and W.dim() == 2 and not hasattr(W, "_eqh_transient"):
                LINEAR_PARAMS[id(W)] = W
    return _MatmulFan.apply(x, *Ws)


def matmul(x, W):
    """x @ W for a weight stored [in, out] through _MatmulFan (batched / in-place weight gradient)."""
    return matmul_fan(x, W)[0]
