"""The L applications of the merged MHNNSConv (conv.py:169-182; layers.MHNNSConv._forward_merged) as ONE autograd node on the
row-panel kernels (csrc/panel.hip): per application three launches forward (F2, the incidence aggregation, F3 chained with the
next application's F1) and three backward (the incidence backward, B2, B1 chained with the previous application's B3) instead
of nine and twelve, no library GEMM, no stand-alone LayerNorm launches.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import os

import torch

from .. import hip
from ._base import (LINEAR_PARAMS, _acc_target, _f32c, _note_acc, _ptr, _require_gpu, _stream, _workspace, timed)
from .aggregate import entry_weights
from .grads import (_linear_weight_grad, _wgrad_deferred, colsum)
from .rows import inc_fwd_col_bytes
from .panel import (conv_panel, conv_panel_slab, panel_gemm, panel_pack, panel_supported)

FOLD_B2 = not os.environ.get("EQH_NO_B2_FOLD")     # dhbar = dqb w12 inside B1 (after its gather) instead of a launch of its own
FOLD_INC = not os.environ.get("EQH_NO_INC_FOLD")   # the incidence aggregation (k_inc_fwd_col) as the prologue of F3 instead of a launch
USE_CONV_STACK = not os.environ.get("EQH_NO_CONV_STACK")     # tests switch it off to compare with the unfused path


def conv_stack_supported(X, C: int) -> bool:
    return (USE_CONV_STACK and X.is_cuda and X.dim() == 2 and X.dtype == torch.float32 and X.shape[1] == C
            and panel_supported(C) and X.shape[0] > 0)


def _wgrad_scaled(weight, dy, x, alpha):
    """alpha * dy.T @ x into the weight's accumulator (deferred to the batched launch when possible); returns None then, else
    the gradient tensor."""
    gbuf = getattr(weight, "_eqh_gbuf", None)
    if gbuf is not None:
        if not _wgrad_deferred(dy, x, alpha, gbuf):
            gbuf.addmm_(dy.t(), x, alpha=alpha)
        return None
    return torch.addmm(weight, dy.t(), x, beta=0.0, alpha=alpha)


def _sum_opt(a, b):
    return b if a is None else (a if b is None else a + b)


class _MergedConvStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, cw, W1a, b1a, g1, be1, W2a, g2, be2, w12, b12, w23, b3a, g3, be3, W3b, b3b, ix, L, scale, eps, relu_out):
        _require_gpu(X, "merged_conv_stack")
        X, cw = _f32c(X), _f32c(cw)
        N, C = X.shape
        dev = X.device
        M = ix.by_e.n_rows
        need_grad = any(ctx.needs_input_grad)
        W2v = W2a[:, :C]
        items = [(W1a, True), (W2v, True), (w12, True), (w23, True), (W3b, True)]
        if need_grad:
            items += [(W3b, False), (w23, False), (w12, False), [(W1a, False), (W2v, False)]]
        imgs = panel_pack(items)
        iW1a, iW2v, iw12, iw23, iW3b = imgs[:5]
        new = lambda r: torch.empty((r, C), dtype=torch.float32, device=dev)
        flops = lambda rows, n: 2 * rows * C * C * n
        L_ = hip.lib()
        saved = []
        h1, h1n, pa = new(N), new(N), new(N)
        timed("k_conv_f1", flops(N, 2), lambda: conv_panel(hip.HG_CONV_F1, N, C, dev, eps=eps[0], in0=X, ld0=X.stride(0), w0=iW1a,
                                                           w1=iW2v, b0=b1a, g0=g1, be0=be1, out0=h1, out1=h1n, out2=pa))
        x_in = X.detach()          # (plain aliases on ctx: a tensor that carries a grad_fn there would tie the graph into a cycle)
        for l in range(L):
            from ._base import signal_take
            # (before application l: where a trainer may release the next batch's index build -- posted by the F2 launch itself)
            sig = signal_take(f"conv_{l}")
            if sig is None and l == L - 1:
                sig = signal_take("conv_last")
            hbar, qb, s = new(M), new(M), new(N)
            timed("k_conv_f2", flops(M, 1), lambda: conv_panel(hip.HG_CONV_F2, M, C, dev, in0=h1n, rowptr=ix.by_e.rowptr,
                                                               col=ix.by_e.col, w0=iw12, bias_out=b12, out0=hbar, out1=qb, signal=sig))
            by_v = ix.by_v
            tail = l + 1 < L
            u, x3, xn = new(N), new(N), new(N)
            nh1, nh1n, npa = (new(N), new(N), new(N)) if tail else (None, None, None)
            f3 = dict(eps=eps[2], scale=scale, relu=relu_out, tail=tail, in1=cw, w0=iw23, b0=b3a, g0=g3, be0=be3, w1=iW3b,
                      bias_out=b3b, out0=u, out1=x3, out2=xn, w2=iW1a, w3=iW2v, b1=b1a, g1=g1, be1=be1, out3=nh1, out4=nh1n,
                      out5=npa)
            if FOLD_INC:
                # the per-incidence hidden layer + hyperedge -> node mean (conv.py:175-177) is F3's prologue: s is written for
                # the backward pass, never read back
                timed("k_conv_f3", flops(N, 4 if tail else 2), lambda: conv_panel(
                    hip.HG_CONV_F3, N, C, dev, in0=pa, in2=qb, rowptr=by_v.rowptr, col=by_v.col, g_inc=g2, be_inc=be2,
                    eps_inc=eps[1], out6=s, **f3))
            else:
                timed("k_inc_fwd_col", inc_fwd_col_bytes(by_v.nnz, N, C), lambda: hip.check(L_.hg_incidence_ln_reduce_fwd_col(
                    _ptr(pa), _ptr(qb), _ptr(by_v.rowptr), _ptr(by_v.col), 1, _ptr(g2), _ptr(be2), N, C, 1, float(eps[1]), _ptr(s),
                    _stream(dev)), "hg_incidence_ln_reduce_fwd_col"))
                timed("k_conv_f3", flops(N, 4 if tail else 2), lambda: conv_panel(hip.HG_CONV_F3, N, C, dev, in0=s, **f3))
            saved.append((x_in, h1, hbar, pa, qb, s, u, x3, xn))
            x_in = xn
            if tail:
                h1, h1n, pa = nh1, nh1n, npa
        if need_grad:
            ctx.save_for_backward(W1a, b1a, g1, W2a, g2, w12, w23, b3a, g3, W3b)
            ctx.saved_rows = saved
            ctx.imgs = imgs[5:]
            ctx.meta = (ix, L, float(scale), eps, bool(relu_out))
            ctx.params = (b1a, g1, be1, g2, be2, b12, b3a, g3, be3, b3b)
        # the OUTPUT object gets this node as its grad_fn; ctx.saved_rows holds xn itself, so hand out an alias -- a reference
        # cycle (node -> ctx -> xn -> node) would keep every step's graph, and with it AccumulateGrad nodes bound to the stream
        # of an earlier step, alive until the garbage collector runs (a hipGraph capture then fails: legacy-stream dependency)
        return xn.view_as(xn)

    @staticmethod
    def backward(ctx, dout):
        W1a, b1a, g1, W2a, g2, w12, w23, b3a, g3, W3b = ctx.saved_tensors
        ix, L, scale, eps, relu_out = ctx.meta
        iW3b_n, iw23_n, iw12_n, istack = ctx.imgs
        p_b1a, p_g1, p_be1, p_g2, p_be2, p_b12, p_b3a, p_g3, p_be3, p_b3b = ctx.params
        rows = ctx.saved_rows
        N, C = rows[0][0].shape
        dev = dout.device
        M = ix.by_e.n_rows
        by_v, by_e = ix.by_v, ix.by_e
        new = lambda r: torch.empty((r, C), dtype=torch.float32, device=dev)
        flops = lambda r, n: 2 * r * C * C * n
        L_ = hip.lib()
        ew = entry_weights(by_v, by_e)
        # vector gradients of LayerNorm 1 / 3 (+ the bias before them): straight into the accumulators when all three have one
        t1 = [_acc_target(p) for p in (p_b1a, p_g1, p_be1)]
        t3 = [_acc_target(p) for p in (p_b3a, p_g3, p_be3)]
        # (a chained launch reduces both groups with ONE accumulate flag: in place only when all six own an accumulator)
        acc1 = acc3 = all(t is not None for t in t1 + t3)
        sm1, sm3 = [], []

        def vec_out(acc_ok, targets, store):
            if acc_ok:
                return targets
            small = torch.empty((3, C), dtype=torch.float32, device=dev)
            store.append(small)
            return list(small)

        dcw = new(N)
        g2_acc = _acc_target(p_g2)
        dgamma2 = g2_acc if g2_acc is not None else None
        dg2_parts, dbe2, db12, db3b = [], None, None, None
        dW = {"W1a": None, "W2a": None, "w12": None, "w23": None, "W3b": None}
        dout = _f32c(dout)
        # B3 of the last application
        x_in, h1, hbar, pa, qb, s, u, x3, xn = rows[L - 1]
        g_l = new(N) if relu_out else dout
        dpre, ds = new(N), new(N)
        o3 = vec_out(acc3, t3, sm3)
        timed("k_conv_b3", flops(N, 2), lambda: conv_panel(
            hip.HG_CONV_B3, N, C, dev, eps=eps[2], scale=scale, acc_first=True, accumulate=acc3, in0=dout, ld0=dout.stride(0),
            in1=xn if relu_out else None, w0=iW3b_n, w1=iw23_n, in2=u, b0=b3a, g0=g3, out0=g_l if relu_out else None, out1=dpre,
            out2=ds, acc_out=dcw, slab=conv_panel_slab(N, C, dev), dbias=o3[0], dgamma=o3[1], dbeta=o3[2]))
        dX = None
        for l in range(L - 1, -1, -1):
            x_in, h1, hbar, pa, qb, s, u, x3, xn = rows[l]
            # weight / bias gradients of this application's tail (W3b, b3b, w23) from g, x3, dpre, s
            dW["W3b"] = _sum_opt(dW["W3b"], _linear_weight_grad(W3b, None, None, g_l, x3))
            db3b = _sum_opt(db3b, colsum(g_l, into=_acc_target(p_b3b)))
            dW["w23"] = _sum_opt(dW["w23"], _wgrad_scaled(w23, dpre, s, scale))
            # incidence backward: ds -> dpa, dqb (+ d gamma2; d beta2 = row-weighted column sum of ds)
            dpa, dqb = new(N), new(M)
            dg2 = dgamma2 if dgamma2 is not None else torch.empty(C, dtype=torch.float32, device=dev)
            ws_bytes = L_.hg_incidence_ln_reduce_bwd_workspace_bytes(N, C)
            ws = _workspace(ws_bytes, dev)
            nnz_ = by_v.nnz
            timed("k_inc_bwd_both", 4 * C * (4 * nnz_ + 2 * (N + M)) + 2 * 20 * nnz_ + 4 * (N + M + 2),
                  lambda: hip.check(L_.hg_incidence_ln_reduce_bwd(
                      _ptr(pa), _ptr(qb), _ptr(ix.v32), _ptr(ix.e32), _ptr(by_v.rowptr), _ptr(by_v.perm), N, _ptr(by_e.rowptr),
                      _ptr(by_e.perm), M, _ptr(ix.v32), _ptr(by_v.rowptr), _ptr(ds), _ptr(g2), C, 1, float(eps[1]), _ptr(dpa),
                      _ptr(dqb), _ptr(dg2), 1 if dgamma2 is not None else 0, _ptr(ws), ws_bytes, _stream(dev)),
                      "hg_incidence_ln_reduce_bwd"))
            if dgamma2 is None:
                dg2_parts.append(dg2)
            dbe2 = _sum_opt(dbe2, colsum(ds, by_v.rowptr, 1, into=_acc_target(p_be2)))
            # B2 (dhbar = dqb w12) rides inside B1: the gathered mean is linear, so B1 gathers dqb and multiplies the sums by w12
            b1_in, b1_w3 = dqb, iw12_n
            if not FOLD_B2:
                b1_in, b1_w3 = new(M), None
                timed("k_conv_b2", flops(M, 1), lambda: panel_gemm(dqb, iw12_n, C, out=b1_in))
            dW["w12"] = _sum_opt(dW["w12"], _linear_weight_grad(w12, None, None, dqb, hbar))
            db12 = _sum_opt(db12, colsum(dqb, into=_acc_target(p_b12)))
            # B1 (+ B3 of the application before)
            tail = l > 0
            dh1 = new(N)
            o1 = vec_out(acc1, t1, sm1)
            if tail:
                pu, pxn = rows[l - 1][6], rows[l - 1][8]
                ng = new(N) if relu_out else None
                ndx = None if relu_out else new(N)
                ndpre, nds = new(N), new(N)
                o3 = vec_out(acc3, t3, sm3)
                timed("k_conv_b1", flops(N, 5), lambda: conv_panel(
                    hip.HG_CONV_B1, N, C, dev, eps=eps[0], scale=scale, tail=True, acc_first=False, accumulate=acc1,
                    in0=b1_in, w3=b1_w3, rowptr=by_v.rowptr, col=by_v.col, wq=ew, in1=h1, b0=b1a, g0=g1, in2=dpa, w0=istack, out0=dh1,
                    out1=ndx, slab=conv_panel_slab(N, C, dev), dbias=o1[0], dgamma=o1[1], dbeta=o1[2],
                    in3=pxn if relu_out else None, w1=iW3b_n, w2=iw23_n, out5=pu, b1=b3a, g1=g3, out2=ng, out3=ndpre, out4=nds,
                    acc_out=dcw, slab2=conv_panel_slab(N, C, dev), dbias2=o3[0], dgamma2=o3[1], dbeta2=o3[2]))
            else:
                dX = new(N)
                timed("k_conv_b1", flops(N, 3), lambda: conv_panel(
                    hip.HG_CONV_B1, N, C, dev, eps=eps[0], tail=False, accumulate=acc1, in0=b1_in, w3=b1_w3, rowptr=by_v.rowptr, col=by_v.col,
                    wq=ew, in1=h1, b0=b1a, g0=g1, in2=dpa, w0=istack, out0=dh1, out1=dX, slab=conv_panel_slab(N, C, dev),
                    dbias=o1[0], dgamma=o1[1], dbeta=o1[2]))
            dW["W1a"] = _sum_opt(dW["W1a"], _linear_weight_grad(W1a, None, None, dh1, x_in))
            dW["W2a"] = _sum_opt(dW["W2a"], _linear_weight_grad(W2a, 0, C, dpa, x_in))
            if tail:
                g_l = ng if relu_out else ndx
                dpre, ds = ndpre, nds
        def vec_grads(acc_ok, targets, parts):
            if acc_ok:
                return [None, None, None]
            tot = parts[0]
            for p in parts[1:]:
                tot = tot + p
            outs = []
            for g_, t in zip(tot, targets):
                if t is not None:
                    t.add_(g_)
                    outs.append(None)
                else:
                    outs.append(g_)
            return outs
        db1a, dg1, dbe1 = vec_grads(acc1, t1, sm1)
        db3a, dg3, dbe3 = vec_grads(acc3, t3, sm3)
        dg2_out = None
        if dgamma2 is None:
            dg2_out = dg2_parts[0]
            for p in dg2_parts[1:]:
                dg2_out = dg2_out + p
        need = ctx.needs_input_grad
        return (dX if need[0] else None, dcw if need[1] else None, dW["W1a"], db1a, dg1, dbe1, dW["W2a"], dg2_out, dbe2, dW["w12"],
                db12, dW["w23"], db3a, dg3, dbe3, dW["W3b"], db3b, None, None, None, None, None)


def merged_conv_stack(X, cw, W1, W2, W3, w12, b12, w23, ix, L: int, scale: float, relu_out: bool):
    """L applications of the merged MHNNSConv on the panel kernels.  W1 / W2 / W3: the layer's MLPs (two Linears each,
    LayerNorm), w12 / b12 / w23: the merged weights of layers.MHNNSConv._prepare_merged, cw: its layer-independent term."""
    l10, l20, l30, l31 = W1.lins[0], W2.lins[0], W3.lins[0], W3.lins[1]
    n1, n2, n3 = W1.normalizations[1], W2.normalizations[1], W3.normalizations[1]
    if torch.is_grad_enabled():
        for w in (l10.weight, l20.weight, l31.weight):
            if w.requires_grad and w.is_leaf:
                LINEAR_PARAMS[id(w)] = w
        _note_acc(l10.bias, n1.weight, n1.bias, n2.weight, n2.bias, l30.bias, n3.weight, n3.bias, l31.bias)
    return _MergedConvStack.apply(X, cw, l10.weight, l10.bias, n1.weight, n1.bias, l20.weight, n2.weight, n2.bias, w12, b12, w23,
                                  l30.bias, n3.weight, n3.bias, l31.weight, l31.bias, ix, int(L), float(scale),
                                  (float(n1.eps), float(n2.eps), float(n3.eps)), bool(relu_out))
