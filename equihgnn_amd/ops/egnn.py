"""EGNN front end: fused edge MLP kernels, feature GEMM + LayerNorm node, weight packing (egnn_layer.py).

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import os

import torch

from .. import hip
from ._base import (
    ACC_PARAMS, LINEAR_PARAMS, _DEFER, _acc_target, _f32c, _hand_out, _note_acc, _ptr, _require_gpu, _rows_ld, _stream,
    _workspace, timed)
from .aggregate import (CSR)
from .products import (USE_X6, X6_WGRAD_ROWS, gemm, gemm_supported, mm_nn, mm_nt)
from .grads import (_merged_acc, _wgrad_deferred, colsum)


class _EgnnEdge(torch.autograd.Function):
    """m_i = sum_j silu(W2 silu(A_i + B_j + wd d2_ij) + b2) — the fused EGNN edge update
    (egnn_layer.py:298-310,357-358).  Saves only ``ab`` and the 16x16 second-layer
    pre-activations; the per-edge hidden activations are recomputed in the backward."""

    @staticmethod
    def forward(ctx, ab, wd, w2, b2, nbr, d2, csr_t: CSR, b2_param=None, w_refs=None):
        _require_gpu(ab, "egnn_edge")
        ab, wd, w2, b2 = _f32c(ab), _f32c(wd), _f32c(w2), _f32c(b2)
        N, Hp = ab.shape[0], ab.shape[1] // 2
        if nbr.shape != (N, 16) or w2.shape != (16, Hp) or wd.shape != (Hp,) or b2.shape != (16,):
            raise ValueError("egnn_edge: shapes must be ab[N,2Hp] wd[Hp] w2[16,Hp] b2[16] nbr[N,16]")
        m = torch.empty((N, 16), dtype=torch.float32, device=ab.device)
        pre2 = torch.empty((N, 16, 16), dtype=torch.float32, device=ab.device)
        # MFMA flops only: 2 * 16 outputs per (edge, hidden unit)
        timed("egnn_edge_fwd", N * 16 * Hp * 32,
              lambda: hip.check(hip.lib().egnn_edge_fwd(_ptr(ab), _ptr(wd), _ptr(w2), _ptr(b2), _ptr(nbr), _ptr(d2), N, Hp,
                                                        _ptr(m), _ptr(pre2), _stream(ab.device)), "egnn_edge_fwd"))
        ctx.save_for_backward(ab, wd, w2, pre2)
        ctx.nbr, ctx.d2, ctx.csr_t, ctx.b2_param, ctx.w_refs = nbr, d2, csr_t, b2_param, w_refs
        return m

    @staticmethod
    def backward(ctx, dm):
        ab, wd, w2, pre2 = ctx.saved_tensors
        dm, dm_ld = _rows_ld(dm)          # usually the last 16 columns of d node_in: read in place
        N, Hp = ab.shape[0], ab.shape[1] // 2
        dev = ab.device
        dab = torch.empty_like(ab)
        # (wd / w2 with accumulators of their own -- the packed weights of a deferral window, egnn_pack_weights: the slab
        # reduction adds into them and joins the step's batched one)
        tw = [_acc_target(q) for q in ctx.w_refs] if ctx.w_refs is not None else [None, None]
        w_acc = all(t is not None for t in tw)
        dwd = tw[0] if w_acc else torch.empty_like(wd)
        dw2 = tw[1] if w_acc else torch.empty_like(w2)
        dpre2 = torch.empty_like(pre2)
        L = hip.lib()
        ws_bytes = L.egnn_edge_bwd_workspace_bytes(N, Hp)
        ws = _workspace(ws_bytes, dev)   # holds the d b2 slabs: parked while reductions are deferred
        tg = _acc_target(ctx.b2_param)   # d b2 = sum of dpre2 over nodes and slots, from the same pass
        db2 = tg if tg is not None else torch.empty(16, dtype=torch.float32, device=dev)
        timed("egnn_edge_bwd", N * 16 * Hp * 96,     # MFMA flops only: three 16-wide products per (edge, hidden unit)
              lambda: hip.check(L.egnn_edge_bwd(_ptr(ab), _ptr(wd), _ptr(w2), _ptr(ctx.nbr), _ptr(ctx.d2), _ptr(pre2),
                                                _ptr(dm), dm_ld, _ptr(ctx.csr_t.rowptr), _ptr(ctx.csr_t.perm), N, Hp, _ptr(dab),
                                                _ptr(dwd), _ptr(dw2), _ptr(dpre2), _ptr(db2), 1 if tg is not None else 0,
                                                1 if w_acc else 0, _ptr(ws), ws_bytes, _stream(dev)), "egnn_edge_bwd"))
        return (dab, None if w_acc else dwd, None if w_acc else dw2, (None if tg is not None else db2), None, None, None, None,
                None)


class _EgnnFeats(torch.autograd.Function):
    """The three uses of the node features in an EGNN layer as ONE autograd node: ab = feats @ w_cat.T + b_cat
    (both halves of the first edge Linear, egnn_layer.py:298-305 split by columns), LayerNorm(feats)
    (node_norm, :192/:360) and feats itself for the residual (:362).  Their three input gradients arrive
    together, so they meet inside the LayerNorm backward kernel (its ``add`` operand) and an accumulating
    GEMM instead of two add kernels."""

    @staticmethod
    def forward(ctx, feats, w_cat, b_cat, gamma, beta, eps, acc_params, w_refs=None):
        _require_gpu(feats, "egnn_feats")
        feats = _f32c(feats)
        R, C = feats.shape
        ctx.fold = gamma is None          # the LayerNorm runs inside the node-update panel launches (egnn_node_mlp_ln)
        if ctx.fold:
            ab = mm_nt(feats, w_cat, bias=b_cat)
            ctx.save_for_backward(feats, w_cat)
            ctx.eps, ctx.acc, ctx.w_refs = 0.0, acc_params, w_refs
            ctx.set_materialize_grads(False)
            return ab, feats.view_as(feats), feats.view_as(feats)
        g, b = _f32c(gamma), _f32c(beta)
        normed = torch.empty_like(feats)
        hip.check(hip.lib().hg_layer_norm_fwd(_ptr(feats), _ptr(g), _ptr(b), R, C, float(eps), _ptr(normed),
                                              _stream(feats.device)), "hg_layer_norm_fwd")
        ab = mm_nt(feats, w_cat, bias=b_cat)
        ctx.save_for_backward(feats, w_cat, g)
        ctx.eps, ctx.acc, ctx.w_refs = float(eps), acc_params, w_refs
        ctx.set_materialize_grads(False)
        return ab, normed, feats.view_as(feats)

    @staticmethod
    def backward(ctx, d_ab, d_normed, d_res):
        if ctx.fold:
            feats, w_cat = ctx.saved_tensors
            gamma = None
            # (d_normed IS the complete gradient of feats from the node update -- LayerNorm backward + residual, formed in
            # k_node_b; the accumulating GEMM below adds the edge MLP's share onto it in place)
            if d_normed is not None and d_res is not None:
                d_res = d_normed + d_res
            elif d_normed is not None:
                d_res = d_normed
            d_normed = None
        else:
            feats, w_cat, gamma = ctx.saved_tensors
        R, C = feats.shape
        dev = feats.device
        L = hip.lib()
        dgamma = dbeta = None
        if d_normed is not None:
            d_normed, dy_ld = _rows_ld(d_normed)     # usually the first C columns of d node_in: read in place
            add = _f32c(d_res) if d_res is not None else None
            dx = torch.empty_like(feats)
            ws_bytes = L.hg_layer_norm_bwd_workspace_bytes(R, C)
            ws = _workspace(ws_bytes, dev)
            tg = [_acc_target(p) for p in ctx.acc]
            in_place = all(t is not None for t in tg)
            small = tg if in_place else list(torch.empty((2, C), dtype=torch.float32, device=dev))
            hip.check(L.hg_layer_norm_bwd(_ptr(feats), _ptr(gamma), _ptr(d_normed), dy_ld, _ptr(add), R, C, ctx.eps, _ptr(dx),
                                          _ptr(small[0]), _ptr(small[1]), 1 if in_place else 0, _ptr(ws), ws_bytes,
                                          _stream(dev)), "hg_layer_norm_bwd")
            if not in_place:
                dgamma, dbeta = _hand_out(list(small), tg)
        elif ctx.fold:
            dx = _f32c(d_res) if d_res is not None else None      # (the node update's gradient tensor: read as the GEMM's addend, not written)
        else:
            dx = _f32c(d_res).clone() if d_res is not None else None
        dw = db = None
        if d_ab is not None:
            d_ab = _f32c(d_ab)
            if dx is None:
                dx = mm_nn(d_ab, w_cat)
            elif ctx.fold:
                dx = mm_nn(d_ab, w_cat, d=dx)         # out of place: a gradient autograd handed in is not modified
            else:
                dx = mm_nn(d_ab, w_cat, d=dx, out=dx)
            tw = [_acc_target(q) for q in ctx.w_refs] if ctx.w_refs is not None else [None, None]
            if ctx.needs_input_grad[1]:
                if tw[0] is not None:       # w_cat carries an accumulator (egnn_pack_weights inside a deferral window)
                    if USE_X6 and R >= X6_WGRAD_ROWS and gemm_supported(d_ab, feats, True, False):
                        gemm(d_ab, feats, trans_a=True, trans_b=False, d=tw[0], out=tw[0])
                    elif not _wgrad_deferred(d_ab, feats, 1.0, tw[0]):
                        tw[0].addmm_(d_ab.t(), feats)
                elif USE_X6 and R >= X6_WGRAD_ROWS and gemm_supported(d_ab, feats, True, False):
                    dw = gemm(d_ab, feats, trans_a=True, trans_b=False)      # split-K x6: 257 against 321 us at 31 k atoms
                else:
                    dw = d_ab.t() @ feats
            if ctx.needs_input_grad[2]:
                if tw[1] is not None:
                    # b_cat = [b1 ; 0] (egnn_pack_weights): only its first half is a parameter's image -- the column sums of
                    # the sender half of d_ab (20 MB at the BASELINE batch) would be thrown away by egnn_pack_weights_bwd
                    half = d_ab.shape[1] // 2
                    colsum(d_ab[:, :half], into=tw[1][:half])
                    db = None
                else:
                    db = colsum(d_ab)
        return dx, dw, db, dgamma, dbeta, None, None, None


def egnn_feats(feats, w_cat, b_cat, norm):
    """(feats @ w_cat.T + b_cat, LayerNorm(feats), feats) for 2-D fp32 ``feats`` [N, C] (C % 4 == 0, C <= 1024);
    ``norm`` the nn.LayerNorm module -- or None: the LayerNorm is left to the node update (egnn_node_mlp_ln) and the second
    result is feats itself.  See _EgnnFeats."""
    if norm is None:
        return _EgnnFeats.apply(feats, w_cat, b_cat, None, None, 0.0, (), (w_cat, b_cat))
    _note_acc(norm.weight, norm.bias)
    return _EgnnFeats.apply(feats, w_cat, b_cat, norm.weight, norm.bias, norm.eps, (norm.weight, norm.bias), (w_cat, b_cat))


class _EgnnPackWeights(torch.autograd.Function):
    """(lin1.weight [H,2C+1], lin1.bias [H], lin2.weight [16,H]) -> (w_cat, b_cat, wd, w2p): the
    operand layout of the fused EGNN edge kernel, one launch each way."""

    @staticmethod
    def forward(ctx, w1, b1, w2, Hp, acc_params):
        _require_gpu(w1, "egnn_pack_weights")
        w1, b1, w2 = _f32c(w1), _f32c(b1), _f32c(w2)
        H, in_ld = w1.shape
        C = (in_ld - 1) // 2
        dev = w1.device
        w_cat = torch.empty((2 * Hp, C), dtype=torch.float32, device=dev)
        b_cat = torch.empty(2 * Hp, dtype=torch.float32, device=dev)
        wd = torch.empty(Hp, dtype=torch.float32, device=dev)
        w2p = torch.empty((16, Hp), dtype=torch.float32, device=dev)
        hip.check(hip.lib().egnn_pack_weights_fwd(_ptr(w1), _ptr(b1), _ptr(w2), H, Hp, C, _ptr(w_cat), _ptr(b_cat),
                                                  _ptr(wd), _ptr(w2p), _stream(dev)), "egnn_pack_weights_fwd")
        ctx.dims = (H, Hp, C)
        ctx.acc = acc_params
        return w_cat, b_cat, wd, w2p

    @staticmethod
    def backward(ctx, dw_cat, db_cat, dwd, dw2p):
        H, Hp, C = ctx.dims
        dev = dw_cat.device
        tg = [_acc_target(p) for p in ctx.acc]
        in_place = all(t is not None for t in tg)      # the parameters' accumulators: nothing left for autograd
        if in_place:
            dw1, db1, dw2 = tg
        else:
            dw1 = torch.empty((H, 2 * C + 1), dtype=torch.float32, device=dev)
            db1 = torch.empty(H, dtype=torch.float32, device=dev)
            dw2 = torch.empty((16, H), dtype=torch.float32, device=dev)
        hip.check(hip.lib().egnn_pack_weights_bwd(_ptr(_f32c(dw_cat)), _ptr(_f32c(db_cat)), _ptr(_f32c(dwd)),
                                                  _ptr(_f32c(dw2p)), H, Hp, C, _ptr(dw1), _ptr(db1), _ptr(dw2),
                                                  1 if in_place else 0, _stream(dev)), "egnn_pack_weights_bwd")
        if in_place:
            return None, None, None, None, None
        return (*_hand_out([dw1, db1, dw2], tg), None, None)


def egnn_pack_weights(w1, b1, w2, Hp):
    if torch.is_grad_enabled():
        for w in (w1, b1, w2):
            if w.requires_grad and w.is_leaf:
                (LINEAR_PARAMS if w.dim() == 2 else ACC_PARAMS)[id(w)] = w
    res = _EgnnPackWeights.apply(w1, b1, w2, Hp, (w1, b1, w2))
    if not (DEFER_PACK_BWD and _DEFER["active"] and torch.is_grad_enabled() and any(t.requires_grad for t in res)):
        return res
    # Inside a deferral window (graphed trainer) the packed weights are detached leaves with zeroed accumulators of their
    # own, like the merged weights of ops.merged_weights: the weight / bias gradients of their consumers (egnn_feats' batched
    # product and column sum, the edge kernels' slab reduction) join the step's batched launches, and defer_flush then runs
    # this node's backward ONCE on the finished sums -- instead of two slab reductions and a column-sum pass launched
    # mid-backward only because the un-packing kernel sat there (17 us per c2 step).
    shapes = [(t.shape[0], t.shape[1]) if t.dim() == 2 else (1, t.shape[0]) for t in res]
    accs = [_merged_acc(sh, res[0].device).view(t.shape) for sh, t in zip(shapes, res)]
    leaves = []
    for t, a in zip(res, accs):
        leaf = t.detach().requires_grad_()
        leaf._eqh_transient = True
        leaf._eqh_gbuf = a
        leaves.append(leaf)
    _DEFER["merged"].append((list(res), accs))
    return tuple(leaves)


def egnn_edge(ab, wd, w2, b2, nbr, d2, csr_t: CSR):
    _note_acc(b2)
    return _EgnnEdge.apply(ab, wd, w2, b2, nbr, d2, csr_t, b2, (wd, w2))


NODE_WGRAD_BLOCKS = not os.environ.get("EQH_NO_NODE_WGRAD_BLOCKS")
DEFER_PACK_BWD = not os.environ.get("EQH_NO_DEFER_PACK")     # the packed EGNN edge weights get accumulators of their own inside a deferral window


class _EgnnNodeMlp(torch.autograd.Function):
    """out = Linear3(silu(Linear0([normed | m_i]))) + feats (egnn_layer.py:180-187,360-362) as ONE panel launch each way
    (csrc/panel.hip: HG_EGNN_NODE_F / _B) instead of cat + GEMM + SiLU + GEMM + add and their backward launches.  The weight
    and bias gradients are formed from the stored rows (node_in, dpre; hid, dout) by the batched / deferred products."""

    @staticmethod
    def forward(ctx, normed, m_i, res, w0, b0, w3, b3, gamma=None, beta=None, eps=0.0):
        """``gamma`` given (round 6): ``normed`` is the RAW feature rows, LayerNorm(gamma, beta, eps) is formed inside the
        launch (and its backward inside the backward launch), ``res`` is ignored (the residual is the same rows)."""
        from .panel import conv_panel, panel_pack
        _require_gpu(normed, "egnn_node_mlp")
        ctx.fold = gamma is not None
        normed, m_i = _f32c(normed), _f32c(m_i)
        res = normed if ctx.fold else _f32c(res)
        N, C = normed.shape
        dev = normed.device
        need_grad = any(ctx.needs_input_grad)
        items = [(w0[:C], True), (w0[C:], True), (w3, True)]
        if need_grad:
            items += [(w3[:, :C], False), (w3[:, C:], False), (w0, False, C + 32)]
        imgs = panel_pack(items)
        node_in = torch.empty((N, C + 16), dtype=torch.float32, device=dev)
        hpre = torch.empty((N, 2 * C), dtype=torch.float32, device=dev)
        hid = torch.empty_like(hpre)
        out = torch.empty((N, C), dtype=torch.float32, device=dev)
        ln = dict(g0=_f32c(gamma), be0=_f32c(beta), eps=float(eps)) if ctx.fold else {}
        timed("k_node_f", 2 * N * ((C + 16) * 2 * C + 2 * C * C), lambda: conv_panel(
            hip.HG_EGNN_NODE_F, N, C, dev, in0=normed, in1=m_i, in2=res, w0=imgs[0], w1=imgs[1], w2=imgs[2], b0=b0, bias_out=b3,
            out0=node_in, out1=hpre, out2=hid, out3=out, **ln))
        if need_grad:
            if ctx.fold:
                ctx.save_for_backward(w0, b0, w3, b3, node_in, hpre, hid, m_i, normed, ln["g0"])
                ctx.ln = (gamma, beta, float(eps))
            else:
                ctx.save_for_backward(w0, b0, w3, b3, node_in, hpre, hid, m_i)
            ctx.imgs = imgs[3:]
        return out

    @staticmethod
    def backward(ctx, dout):
        from .panel import conv_panel, conv_panel_slab
        from .grads import _linear_weight_grad
        if ctx.fold:
            w0, b0, w3, b3, node_in, hpre, hid, m_i, feats, g0 = ctx.saved_tensors
        else:
            w0, b0, w3, b3, node_in, hpre, hid, m_i = ctx.saved_tensors
        N, C = hid.shape[0], hid.shape[1] // 2
        dev = dout.device
        dout, ld = _rows_ld(dout)
        dpre = torch.empty_like(hpre)
        dgam = dbet = None
        if ctx.fold:
            # d feats = LayerNorm backward of d normed + dout (the residual), d m_i, and the LayerNorm's vector gradients (slab sums)
            dfeats = torch.empty((N, C), dtype=torch.float32, device=dev)
            dm = torch.empty((N, 16), dtype=torch.float32, device=dev)
            p_gamma, p_beta, eps = ctx.ln
            tg = [_acc_target(p_gamma), _acc_target(p_beta)]
            acc = all(t is not None for t in tg)
            small = tg if acc else list(torch.empty((2, C), dtype=torch.float32, device=dev))
            junk = torch.empty(C, dtype=torch.float32, device=dev)          # (the slab's bias third: a plain LayerNorm has none; never read)
            timed("k_node_b", 2 * N * (2 * C * C + 2 * C * (C + 16)), lambda: conv_panel(
                hip.HG_EGNN_NODE_B, N, C, dev, eps=eps, accumulate=acc, in0=dout, ld0=ld, in1=hpre, w0=ctx.imgs[0], w1=ctx.imgs[1],
                w2=ctx.imgs[2], out0=dpre, out1=dfeats, out2=dm, in3=feats, g0=g0, slab=conv_panel_slab(N, C, dev), dbias=junk,
                dgamma=small[0], dbeta=small[1]))
            if not acc:
                dgam, dbet = small[0], small[1]
        else:
            dnode_in = torch.empty_like(node_in)
            timed("k_node_b", 2 * N * (2 * C * C + 2 * C * (C + 16)), lambda: conv_panel(
                hip.HG_EGNN_NODE_B, N, C, dev, in0=dout, ld0=ld, in1=hpre, w0=ctx.imgs[0], w1=ctx.imgs[1], w2=ctx.imgs[2], out0=dpre,
                out1=dnode_in))
        dout_c = dout if ld == C else dout.contiguous()
        dw3 = dw0 = None
        blocks = NODE_WGRAD_BLOCKS and _acc_target(w3) is not None and _acc_target(w0) is not None and C % 64 == 0
        if blocks:
            # [C x 2 C] and [2 C x (C + 16)] as [C x C] blocks over column blocks of hid / dpre / node_in, read in place: they
            # join the step's batch of [C x C] weight gradients (hg_wgrad_batch_f32) instead of two lone library products
            # (26 + 31 us at the BASELINE batch); the 16-wide remainder of W0 is one small product
            if ctx.needs_input_grad[5]:
                for c0 in (0, C):
                    _linear_weight_grad(w3, c0, c0 + C, dout_c, hid[:, c0:c0 + C])
            if ctx.needs_input_grad[3]:
                for r0 in (0, C):
                    _linear_weight_grad(w0, 0, C, dpre[:, r0:r0 + C], node_in[:, :C], r0, r0 + C)
                _linear_weight_grad(w0, C, C + 16, dpre, m_i)      # (= node_in[:, C:], already contiguous: no copy launch)
        else:
            dw3 = _linear_weight_grad(w3, None, None, dout_c, hid) if ctx.needs_input_grad[5] else None
            dw0 = _linear_weight_grad(w0, None, None, dpre, node_in) if ctx.needs_input_grad[3] else None
        db3 = colsum(dout_c, into=_acc_target(b3)) if ctx.needs_input_grad[6] else None
        db0 = colsum(dpre, into=_acc_target(b0)) if ctx.needs_input_grad[4] else None
        if ctx.fold:
            return dfeats, dm, None, dw0, db0, dw3, db3, dgam, dbet, None
        return dnode_in[:, :C], dnode_in[:, C:], dout, dw0, db0, dw3, db3, None, None, None


def egnn_node_mlp_supported(normed, m_i, lin0, lin3) -> bool:
    from .panel import panel_supported
    C = normed.shape[-1]
    return (normed.is_cuda and normed.dim() == 2 and normed.dtype == torch.float32 and panel_supported(C) and m_i.shape[-1] == 16
            and tuple(lin0.weight.shape) == (2 * C, C + 16) and tuple(lin3.weight.shape) == (C, 2 * C) and normed.shape[0] > 0)


def egnn_node_mlp(normed, m_i, res, lin0, lin3):
    """lin3(silu(lin0(cat(normed, m_i)))) + res; see _EgnnNodeMlp."""
    if torch.is_grad_enabled():
        for w in (lin0.weight, lin3.weight):
            if w.requires_grad and w.is_leaf:
                LINEAR_PARAMS[id(w)] = w
    _note_acc(lin0.bias, lin3.bias)
    return _EgnnNodeMlp.apply(normed, m_i, res, lin0.weight, lin0.bias, lin3.weight, lin3.bias)


NODE_LN_FOLD = not os.environ.get("EQH_NO_NODE_LN_FOLD")     # node_norm inside the node-update panel launches (A/B runs)


def egnn_node_mlp_ln(feats, m_i, lin0, lin3, norm):
    """lin3(silu(lin0(cat(LayerNorm(feats), m_i)))) + feats (egnn_layer.py:192,360-362) with the LayerNorm -- and its backward,
    summed with the residual's gradient -- inside the node update's one panel launch each way."""
    if torch.is_grad_enabled():
        for w in (lin0.weight, lin3.weight):
            if w.requires_grad and w.is_leaf:
                LINEAR_PARAMS[id(w)] = w
    _note_acc(lin0.bias, lin3.bias, norm.weight, norm.bias)
    return _EgnnNodeMlp.apply(feats, m_i, None, lin0.weight, lin0.bias, lin3.weight, lin3.bias, norm.weight, norm.bias, norm.eps)
