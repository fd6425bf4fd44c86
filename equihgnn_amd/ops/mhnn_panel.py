"""One application of MHNNConv (conv.py:87-101: node AND hyperedge features, four two-layer MLPs on concatenated pairs) as ONE
autograd node on the row-panel kernels (csrc/panel.hip) -- the `mhnn` / `mhnnm` / `egnn_equihnn` / `egnn_equihnnm` methods.

Restructured like MHNNSConv's merged path (layers.py): every first Linear is split by input block
(W cat(a, b) = W_a a + W_b b), the last Linear of a per-incidence MLP commutes with the mean behind it, and where only a
linear map separates two Linears -- W1's last Linear -> mean over the hyperedge -> the message half of W2's first Linear;
W3's last Linear -> mean over the node -> the message half of W4's first Linear -- the pair is one Linear with the product
weight (ops.merged_weights).  Per application, forward:

    pa1, pa3, cwX = X [W1a_x | W3a_x | W4a_X]^T            (one launch: hg_panel_multi, three weight streams over one A image)
    qb1, cwE      = E [W1a_e | W2a_E]^T  (+ b1a; + [deg e > 0] W2a_m b1b)                               (one launch)
    E'  = W2b LN2(relu(mean_{v in e} xhat1(relu(pa1[v] + qb1[e])) w12^T + cwE + b2a)) + b2b   (HG_CONV_F3, incidence prologue)
    qb3 = E' W3a_e^T + b3a                                                                     (hg_panel_multi, n = 1)
    X'  = W4b LN4(relu(mean_{e of v} xhat3(relu(pa3[v] + qb3[e])) w34^T + cwX + b4a)) + b4b   (HG_CONV_F3, incidence prologue)

five panel launches instead of ~30 library-GEMM / row-kernel / ATen launches, and backward two HG_CONV_B3 launches, the two
incidence backward launches and five accumulating products; weight / bias / LayerNorm-vector gradients join the step's
batched launches exactly as the merged MHNNSConv stack's do (ops/conv_stack.py).

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import ctypes
import os

import threading

import torch

from .. import hip
from ._base import (LINEAR_PARAMS, _acc_target, _f32c, _note_acc, _ptr, _require_gpu, _stream, _workspace, timed)
from .grads import (_linear_weight_grad, colsum)
from .conv_stack import (_sum_opt, _wgrad_scaled)
from .panel import (conv_panel, conv_panel_slab, panel_pack, panel_supported)
from .rows import inc_fwd_col_bytes

USE_MHNN_PANEL = not os.environ.get("EQH_NO_MHNN_PANEL")     # tests / A-B runs switch the panel path of MHNNConv off


def panel_multi(a, C, items, rows=None):
    """out[g] = a @ B_g + rw_g[:, None] * bias_g + d_g for up to three packed [C x C] images sharing the A rows (hg_panel_multi).
    ``items``: [(image, bias or None, rw or None, d or None, out)]."""
    a = _f32c(a)
    q = hip.HgPanelMulti()
    q.a, q.lda, q.rows, q.C, q.n = a.data_ptr(), a.stride(0), a.shape[0] if rows is None else rows, C, len(items)
    for g, (img, bias, rw, d, out) in enumerate(items):
        q.w[g], q.out[g], q.ldo[g] = img.data_ptr(), out.data_ptr(), out.stride(0)
        q.bias[g] = bias.data_ptr() if bias is not None else None
        q.rw[g] = rw.data_ptr() if rw is not None else None
        q.d[g] = d.data_ptr() if d is not None else None
        q.ldd[g] = d.stride(0) if d is not None else 0
    hip.check(hip.lib().hg_panel_multi(ctypes.byref(q), _stream(a.device)), "hg_panel_multi")


PANEL_SUM = not os.environ.get("EQH_NO_PANEL_SUM")      # (off: the chained single-product launches of rounds 4-5, for A/B runs)


def panel_sum(items, C, out, d=None, rows=None):
    """out = sum_g a_g @ B_g + d for up to three (a_g [rows, C], packed [C x C] image) pairs over the same rows, one launch
    (hg_panel_sum)."""
    q = hip.HgPanelSum()
    keep = []
    for g, (a, img) in enumerate(items):
        a = _f32c(a)
        keep.append(a)
        q.a[g], q.lda[g], q.w[g] = a.data_ptr(), a.stride(0), img.data_ptr()
    q.rows, q.C, q.n = (keep[0].shape[0] if rows is None else rows), C, len(items)
    q.d, q.ldd = (d.data_ptr(), d.stride(0)) if d is not None else (None, 0)
    q.out, q.ldo = out.data_ptr(), out.stride(0)
    hip.check(hip.lib().hg_panel_sum(ctypes.byref(q), _stream(out.device)), "hg_panel_sum")
    return out


def mhnn_panel_supported(X, E, conv) -> bool:
    ws = (conv.W1, conv.W2, conv.W3, conv.W4)
    if not USE_MHNN_PANEL or any(w is None for w in ws):
        return False
    C = X.shape[-1]
    ok = (X.is_cuda and X.dim() == 2 and E.dim() == 2 and X.dtype == torch.float32 and E.dtype == torch.float32 and panel_supported(C)
          and E.shape[1] == C and X.shape[0] > 0 and E.shape[0] > 0 and conv.aggr == "mean"
          and not (conv.training and conv.dropout > 0))
    for w in ws:
        ok = ok and len(w.lins) == 2 and not w.InputNorm and isinstance(w.normalizations[1], torch.nn.LayerNorm) \
            and w.lins[0].weight.shape == (C, 2 * C) and w.lins[1].weight.shape == (C, C)
    return bool(ok)


def _pack_items(C, W1a, W2a, W2b, W3a, W4a, W4b, w12, w34, need_grad):
    """The images of one MHNNConv application, forward ones first (hg_panel_pack items)."""
    blk = lambda w, half: w[:, :C] if half == 0 else w[:, C:]
    fwd = [(blk(W1a, 0), True), (blk(W3a, 0), True), (blk(W4a, 0), True), (blk(W1a, 1), True), (blk(W2a, 0), True), (w12, True),
           (W2b, True), (blk(W3a, 1), True), (w34, True), (W4b, True)]
    bwd = [(W4b, False), (w34, False), (blk(W3a, 1), False), (W2b, False), (w12, False), (blk(W2a, 0), False),
           (blk(W1a, 1), False), (blk(W4a, 0), False), (blk(W3a, 0), False), (blk(W1a, 0), False)] if need_grad else []
    return fwd + bwd


class _MHNNConvPanel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, E, W1a, b1a, g1, be1, W2a, b2a, g2, be2, W2b, b2b, W3a, b3a, g3, be3, W4a, b4a, g4, be4, W4b, b4b,
                w12, v12, w34, v34, ix, eps, packed=None):
        _require_gpu(X, "mhnn_conv_panel")
        X, E = _f32c(X), _f32c(E)
        N, C = X.shape
        M = E.shape[0]
        dev = X.device
        need_grad = any(ctx.needs_input_grad)
        # (`packed`: the images of this application, packed with those of the model's other layers in ONE launch by the
        # merged_scope around them -- with or without the backward ones, as the scope saw fit)
        if packed is not None and len(packed) >= (20 if need_grad else 10):
            imgs = packed
        else:
            imgs = panel_pack(_pack_items(C, W1a, W2a, W2b, W3a, W4a, W4b, w12, w34, need_grad))
        iW1x, iW3x, iW4x, iW1e, iW2e, iw12, iW2b, iW3e, iw34, iW4b = imgs[:10]
        new = lambda r: torch.empty((r, C), dtype=torch.float32, device=dev)
        flops = lambda rows, n: 2 * rows * C * C * n
        rw_v, rw_e = ix.has_v.reshape(-1), ix.has_e.reshape(-1)          # [deg > 0]: what a mean leaves on the bias behind it
        pa1, pa3, cwX = new(N), new(N), new(N)
        timed("k_panel_multi", flops(N, 3), lambda: panel_multi(X, C, [(iW1x, None, None, None, pa1), (iW3x, None, None, None, pa3),
                                                                      (iW4x, v34, rw_v, None, cwX)]))
        qb1, cwE = new(M), new(M)
        timed("k_panel_multi", flops(M, 2), lambda: panel_multi(E, C, [(iW1e, b1a, None, None, qb1), (iW2e, v12, rw_e, None, cwE)]))
        s_e, u_e, x3_e, En = new(M), new(M), new(M), new(M)
        by_e, by_v = ix.by_e, ix.by_v
        timed("k_conv_f3", flops(M, 2), lambda: conv_panel(
            hip.HG_CONV_F3, M, C, dev, eps=eps[1], scale=1.0, relu=False, tail=False, in0=qb1, in2=pa1, rowptr=by_e.rowptr, col=by_e.col,
            g_inc=g1, be_inc=be1, eps_inc=eps[0], out6=s_e, in1=cwE, w0=iw12, b0=b2a, g0=g2, be0=be2, w1=iW2b, bias_out=b2b, out0=u_e,
            out1=x3_e, out2=En))
        qb3 = new(M)
        timed("k_panel_multi", flops(M, 1), lambda: panel_multi(En, C, [(iW3e, b3a, None, None, qb3)]))
        s_v, u_v, x3_v, Xn = new(N), new(N), new(N), new(N)
        timed("k_conv_f3", flops(N, 2), lambda: conv_panel(
            hip.HG_CONV_F3, N, C, dev, eps=eps[3], scale=1.0, relu=False, tail=False, in0=pa3, in2=qb3, rowptr=by_v.rowptr, col=by_v.col,
            g_inc=g3, be_inc=be3, eps_inc=eps[2], out6=s_v, in1=cwX, w0=iw34, b0=b4a, g0=g4, be0=be4, w1=iW4b, bias_out=b4b, out0=u_v,
            out1=x3_v, out2=Xn))
        if need_grad:
            # En is an OUTPUT that the backward pass reads (dW3a): saved through save_for_backward, so that an in-place
            # operation on the returned E' (an inplace activation, dropout_) trips autograd's version check instead of
            # silently corrupting the weight gradient
            ctx.save_for_backward(X, E, W1a, W2a, W2b, W3a, W4a, W4b, w12, w34, g1, g3, b2a, g2, b4a, g4, En)
            ctx.rows = (pa1, pa3, qb1, qb3, s_e, u_e, x3_e, s_v, u_v, x3_v)
            ctx.imgs = imgs[10:]
            ctx.meta = (ix, eps)
            ctx.params = (b1a, g1, be1, b2a, g2, be2, b2b, b3a, g3, be3, b4a, g4, be4, b4b, v12, v34)
        return Xn, En

    @staticmethod
    def backward(ctx, dXn, dEn):
        X, E, W1a, W2a, W2b, W3a, W4a, W4b, w12, w34, g1, g3, b2a, g2, b4a, g4, En = ctx.saved_tensors
        pa1, pa3, qb1, qb3, s_e, u_e, x3_e, s_v, u_v, x3_v = ctx.rows
        iW4b_n, iw34_n, iW3e_n, iW2b_n, iw12_n, iW2e_n, iW1e_n, iW4x_n, iW3x_n, iW1x_n = ctx.imgs
        ix, eps = ctx.meta
        p_b1a, p_g1, p_be1, p_b2a, p_g2, p_be2, p_b2b, p_b3a, p_g3, p_be3, p_b4a, p_g4, p_be4, p_b4b, p_v12, p_v34 = ctx.params
        N, C = X.shape
        M = E.shape[0]
        dev = X.device
        by_e, by_v = ix.by_e, ix.by_v
        new = lambda r: torch.empty((r, C), dtype=torch.float32, device=dev)
        flops = lambda rows, n: 2 * rows * C * C * n
        L_ = hip.lib()
        blkc = lambda half: (0, C) if half == 0 else (C, 2 * C)

        def vec3(params):
            """(dbias, dgamma, dbeta) destinations of a B3 stage: the parameters' accumulators when all three own one."""
            t = [_acc_target(p) for p in params]
            if all(x is not None for x in t):
                return t, True, None
            small = torch.empty((3, C), dtype=torch.float32, device=dev)
            return list(small), False, small

        def b3(rows, dy, iWb, iwm, u, b_a, gam, params):
            dpre, ds = new(rows), new(rows)
            o, acc, small = vec3(params)
            dy = _f32c(dy)
            timed("k_conv_b3", flops(rows, 2), lambda: conv_panel(
                hip.HG_CONV_B3, rows, C, dev, eps=eps_mlp[0], scale=1.0, acc_first=True, accumulate=acc, in0=dy, ld0=dy.stride(0), w0=iWb,
                w1=iwm, in2=u, b0=b_a, g0=gam, out1=dpre, out2=ds, slab=conv_panel_slab(rows, C, dev), dbias=o[0], dgamma=o[1],
                dbeta=o[2]))
            return dpre, ds, (None, None, None) if acc else tuple(small)

        def inc_bwd(pa, qb, ds, gam, p_gam, p_bet, out_csr, okey, eps_):
            dpa, dqb = new(N), new(M)
            g_acc = _acc_target(p_gam)
            dg = g_acc if g_acc is not None else torch.empty(C, dtype=torch.float32, device=dev)
            ws_bytes = L_.hg_incidence_ln_reduce_bwd_workspace_bytes(N, C)
            ws = _workspace(ws_bytes, dev)
            nnz = by_v.nnz
            timed("k_inc_bwd_both", 4 * C * (4 * nnz + 2 * (N + M)) + 2 * 20 * nnz + 4 * (N + M + 2),
                  lambda: hip.check(L_.hg_incidence_ln_reduce_bwd(
                      _ptr(pa), _ptr(qb), _ptr(ix.v32), _ptr(ix.e32), _ptr(by_v.rowptr), _ptr(by_v.perm), N, _ptr(by_e.rowptr),
                      _ptr(by_e.perm), M, _ptr(okey), _ptr(out_csr.rowptr), _ptr(ds), _ptr(gam), C, 1, float(eps_), _ptr(dpa), _ptr(dqb),
                      _ptr(dg), 1 if g_acc is not None else 0, _ptr(ws), ws_bytes, _stream(dev)), "hg_incidence_ln_reduce_bwd"))
            dbeta = colsum(ds, out_csr.rowptr, 1, into=_acc_target(p_bet))
            return dpa, dqb, (None if g_acc is not None else dg), dbeta

        dW = {}
        acc_w = lambda key, g: dW.__setitem__(key, _sum_opt(dW.get(key), g))
        # ---- W4 (node rows): X' = W4b LN4(relu(s_v w34^T + cwX + b4a)) + b4b ------------------------------------------------
        eps_mlp = (eps[3],)
        dXn = _f32c(dXn)
        dpre_v, ds_v, (db4a, dg4, dbe4) = b3(N, dXn, iW4b_n, iw34_n, u_v, b4a, g4, (p_b4a, p_g4, p_be4))
        acc_w("W4b", _linear_weight_grad(W4b, None, None, dXn, x3_v))
        db4b = colsum(dXn, into=_acc_target(p_b4b))
        acc_w("w34", _wgrad_scaled(w34, dpre_v, s_v, 1.0))
        acc_w("W4a", _linear_weight_grad(W4a, 0, C, dpre_v, X))
        dv34 = colsum(dpre_v, by_v.rowptr, 1, into=_acc_target(p_v34))
        # ---- message 3: s_v <- (pa3, qb3) -----------------------------------------------------------------------------------------
        dpa3, dqb3, dg3, dbe3 = inc_bwd(pa3, qb3, ds_v, g3, p_g3, p_be3, by_v, ix.v32, eps[2])
        acc_w("W3a", _linear_weight_grad(W3a, C, 2 * C, dqb3, En))
        acc_w("W3a", _linear_weight_grad(W3a, 0, C, dpa3, X))
        db3a = colsum(dqb3, into=_acc_target(p_b3a))
        # ---- dE' = dEn (the layer's hyperedge output) + dqb3 W3a_e -------------------------------------------------------------------
        dEt = new(M)
        dEn_c = _f32c(dEn) if dEn is not None else None
        timed("k_panel_multi", flops(M, 1), lambda: panel_multi(dqb3, C, [(iW3e_n, None, None, dEn_c, dEt)]))
        # ---- W2 (hyperedge rows) ------------------------------------------------------------------------------------------------------
        eps_mlp = (eps[1],)
        dpre_e, ds_e, (db2a, dg2, dbe2) = b3(M, dEt, iW2b_n, iw12_n, u_e, b2a, g2, (p_b2a, p_g2, p_be2))
        acc_w("W2b", _linear_weight_grad(W2b, None, None, dEt, x3_e))
        db2b = colsum(dEt, into=_acc_target(p_b2b))
        acc_w("w12", _wgrad_scaled(w12, dpre_e, s_e, 1.0))
        acc_w("W2a", _linear_weight_grad(W2a, 0, C, dpre_e, E))
        dv12 = colsum(dpre_e, by_e.rowptr, 1, into=_acc_target(p_v12))
        # ---- message 1: s_e <- (pa1, qb1) -----------------------------------------------------------------------------------------
        dpa1, dqb1, dg1, dbe1 = inc_bwd(pa1, qb1, ds_e, g1, p_g1, p_be1, by_e, ix.e32, eps[0])
        acc_w("W1a", _linear_weight_grad(W1a, C, 2 * C, dqb1, E))
        acc_w("W1a", _linear_weight_grad(W1a, 0, C, dpa1, X))
        db1a = colsum(dqb1, into=_acc_target(p_b1a))
        # ---- input gradients: dE = dpre_e W2a_E + dqb1 W1a_e;  dX = dpre_v W4a_X + dpa3 W3a_x + dpa1 W1a_x ---------------------------
        need = ctx.needs_input_grad
        dE = dX = None
        if need[1] and PANEL_SUM:        # one launch per sum (round 6): at these row counts a launch is its slot, not its work
            dE = new(M)
            timed("k_panel_sum", flops(M, 2), lambda: panel_sum([(dpre_e, iW2e_n), (dqb1, iW1e_n)], C, dE))
        elif need[1]:
            t = new(M)
            dE = new(M)
            timed("k_panel_multi", flops(M, 1), lambda: panel_multi(dpre_e, C, [(iW2e_n, None, None, None, t)]))
            timed("k_panel_multi", flops(M, 1), lambda: panel_multi(dqb1, C, [(iW1e_n, None, None, t, dE)]))
        if need[0] and PANEL_SUM:
            dX = new(N)
            timed("k_panel_sum", flops(N, 3), lambda: panel_sum([(dpre_v, iW4x_n), (dpa3, iW3x_n), (dpa1, iW1x_n)], C, dX))
        elif need[0]:
            t1, t2 = new(N), new(N)
            dX = new(N)
            timed("k_panel_multi", flops(N, 1), lambda: panel_multi(dpre_v, C, [(iW4x_n, None, None, None, t1)]))
            timed("k_panel_multi", flops(N, 1), lambda: panel_multi(dpa3, C, [(iW3x_n, None, None, t1, t2)]))
            timed("k_panel_multi", flops(N, 1), lambda: panel_multi(dpa1, C, [(iW1x_n, None, None, t2, dX)]))
        return (dX, dE, dW.get("W1a"), db1a, dg1, dbe1, dW.get("W2a"), db2a, dg2, dbe2, dW.get("W2b"), db2b, dW.get("W3a"), db3a, dg3,
                dbe3, dW.get("W4a"), db4a, dg4, dbe4, dW.get("W4b"), db4b, dW.get("w12"), dv12, dW.get("w34"), dv34, None, None, None)


def _merged_items(conv, C):
    W1, W2, W3, W4 = conv.W1, conv.W2, conv.W3, conv.W4
    return [(W2.lins[0].weight, W1.lins[1].weight, W1.lins[1].bias, None, (C, 2 * C)),
            (W4.lins[0].weight, W3.lins[1].weight, W3.lins[1].bias, None, (C, 2 * C))]


class merged_scope:
    """``with merged_scope(convs, X, E):`` around a model's conv applications: the weight-level products of ALL the MHNNConv
    layers inside (two per layer) are formed by one ops.merged_weights call -- one launch forward, one in the backward pass --
    instead of one per application; a conv applied several times (mhnn.py's shared layer) gets them once.  The results are
    functions of the parameters of THIS step: they live in the scope object (found by ``mhnn_conv_panel`` through a per-thread
    stack of open scopes), never on the modules, so nested or concurrent scopes over the same conv cannot remove each other's."""

    def __init__(self, convs, X, E):
        seen, self.convs, self.uses = set(), [], {}
        for c in convs:
            self.uses[id(c)] = self.uses.get(id(c), 0) + 1
            if id(c) not in seen and mhnn_panel_supported(X, E, c):
                seen.add(id(c))
                self.convs.append(c)
        self.C = X.shape[-1]
        self.merged = {}
        self.packed = {}

    def __enter__(self):
        from .linears import merged_weights
        if self.convs:
            items = [it for c in self.convs for it in _merged_items(c, self.C)]
            res = merged_weights(items)
            self.merged = {id(c): (res[2 * k], res[2 * k + 1]) for k, c in enumerate(self.convs)}
            # ... and the operand images of a conv that is applied SEVERAL times (mhnn.py's shared layer) once per step instead of
            # once per application: mhnn 1.137 -> 1.105 ms.  Unshared layers keep their own pack launch right in front of their
            # kernels: packed up front (60 images, 23 MB) the later layers' images have left the L2 by the time they are streamed --
            # mhnnm 1.150 against 1.122 ms.  (EQH_NO_SCOPE_PACK=1: per application for every conv.)
            need_grad = torch.is_grad_enabled()
            lists = []
            if os.environ.get("EQH_NO_SCOPE_PACK"):
                self.convs_packed = []
            else:
                shared = len(self.convs) == 1          # one conv in the scope: every application inside it uses these images
                self.convs_packed = [c for c in self.convs if shared or self.uses.get(id(c), 0) > 1]
            for c in self.convs_packed:
                (w12, _), (w34, _) = self.merged[id(c)]
                lists.append(_pack_items(self.C, c.W1.lins[0].weight, c.W2.lins[0].weight, c.W2.lins[1].weight, c.W3.lins[0].weight,
                                         c.W4.lins[0].weight, c.W4.lins[1].weight, w12, w34, need_grad))
            imgs = panel_pack([it for lst in lists for it in lst]) if lists else []
            off = 0
            for c, lst in zip(self.convs_packed, lists):
                self.packed[id(c)] = imgs[off:off + len(lst)]
                off += len(lst)
        _open_scopes().append(self)
        return self

    def __exit__(self, *exc):
        stack = _open_scopes()
        if self in stack:
            stack.remove(self)
        self.merged = {}
        self.packed = {}
        return False


_SCOPES = threading.local()


def _open_scopes() -> list:
    if not hasattr(_SCOPES, "stack"):
        _SCOPES.stack = []
    return _SCOPES.stack


def _merged_in_scope(conv):
    for sc in reversed(_open_scopes()):
        hit = sc.merged.get(id(conv))
        if hit is not None:
            return hit
    return None


def _packed_in_scope(conv):
    for sc in reversed(_open_scopes()):
        if id(conv) in sc.merged:
            return sc.packed.get(id(conv))
    return None


def mhnn_conv_panel(conv, X, E, ix):
    """MHNNConv.forward(X, E) on the panel kernels (see the module docstring); returns (X', E')."""
    from .linears import merged_weights
    W1, W2, W3, W4 = conv.W1, conv.W2, conv.W3, conv.W4
    C = X.shape[-1]
    pre = _merged_in_scope(conv)
    (w12, v12), (w34, v34) = pre if pre is not None else merged_weights(_merged_items(conv, C))
    if torch.is_grad_enabled():
        for w in (W1.lins[0].weight, W2.lins[0].weight, W2.lins[1].weight, W3.lins[0].weight, W4.lins[0].weight, W4.lins[1].weight):
            if w.requires_grad and w.is_leaf:
                LINEAR_PARAMS[id(w)] = w
        for w in (W1, W2, W3, W4):
            _note_acc(w.lins[0].bias, w.normalizations[1].weight, w.normalizations[1].bias)
        _note_acc(W2.lins[1].bias, W4.lins[1].bias)
    n1, n2, n3, n4 = (w.normalizations[1] for w in (W1, W2, W3, W4))
    return _MHNNConvPanel.apply(X, E, W1.lins[0].weight, W1.lins[0].bias, n1.weight, n1.bias, W2.lins[0].weight, W2.lins[0].bias,
                                n2.weight, n2.bias, W2.lins[1].weight, W2.lins[1].bias, W3.lins[0].weight, W3.lins[0].bias, n3.weight,
                                n3.bias, W4.lins[0].weight, W4.lins[0].bias, n4.weight, n4.bias, W4.lins[1].weight, W4.lins[1].bias,
                                w12, v12, w34, v34, ix, (float(n1.eps), float(n2.eps), float(n3.eps), float(n4.eps)),
                                _packed_in_scope(conv) if pre is not None else None)
