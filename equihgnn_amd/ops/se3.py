"""Equiformer operators: row GEMMs of the radial tensor product, radial trunk, attention pooling, RMS norm, edge
geometry (equiformer_layer.py).

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import ctypes

import os

import torch

from .. import hip
from ._base import (
    ACC_PARAMS, LINEAR_PARAMS, _DEFER, _acc_target, _f32c, _hand_out, _ptr, _require_gpu, _stream, _workspace, timed)
from .products import gemm, gemm_out_ok, gemm_supported, mm_nn, mm_nt


class _RowGemm(torch.autograd.Function):
    """out[e] = z[e] @ w[row(e)] — hg_rowgemm_fwd/bwd (the radial tensor product of
    equiformer_layer.py:376-383 re-associated; see csrc/rowgemm.hip)."""

    @staticmethod
    def forward(ctx, z, w, rowptr, perm):
        _require_gpu(z, "rowgemm")
        z, w = _f32c(z), _f32c(w)
        E, Kd = z.shape
        R, Kd2, L = w.shape
        if Kd2 != Kd or rowptr.numel() != R + 1:
            raise ValueError("rowgemm: z[E,Kd], w[R,Kd,L], rowptr[R+1] expected")
        out = torch.zeros((E, L), dtype=torch.float32, device=z.device)  # entries outside every row stay 0
        timed("k_rowgemm_fwd", 2 * E * Kd * L,
              lambda: hip.check(hip.lib().hg_rowgemm_fwd(_ptr(z), _ptr(w), _ptr(rowptr), _ptr(perm), R, Kd, L, _ptr(out), 0,
                                                         _stream(z.device)), "hg_rowgemm_fwd"))
        ctx.save_for_backward(z, w)
        ctx.rowptr, ctx.perm = rowptr, perm
        return out

    @staticmethod
    def backward(ctx, dout):
        z, w = ctx.saved_tensors
        dout = _f32c(dout)
        R, Kd, L = w.shape
        dz = torch.zeros_like(z) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        hip.check(hip.lib().hg_rowgemm_bwd(_ptr(z), _ptr(w), _ptr(dout), _ptr(ctx.rowptr), _ptr(ctx.perm), R, Kd,
                                           L, _ptr(dz), 0, _ptr(dw), _stream(z.device)), "hg_rowgemm_bwd")
        return dz, dw, None, None


class _RowGemm2(torch.autograd.Function):
    """out[e] = z[e] @ wa[row_a(e)] + z[e] @ wb[row_b(e)] for two groupings of the same entries that each cover
    EVERY entry (sender rows and receiver rows of the neighbour graph): the second pass accumulates into the
    first one's output, and so do the two halves of dz, so neither a zero fill nor an add kernel runs.
    With bias blocks (ba [Ra, MB, L], bb [Rb, MB, L], coef [E, MB] or None = 1):
    out[e] += sum_m coef[e, m] (ba[row_a(e), m] + bb[row_b(e), m]) inside the same launches, their gradients beside the
    node matrices' (hg_rowgemm_fwd_bias / _bwd_bias); coef carries no gradient."""

    @staticmethod
    def forward(ctx, z, wa, rowptr_a, perm_a, wb, rowptr_b, perm_b, ba=None, bb=None, coef=None, z_factored=False):
        _require_gpu(z, "rowgemm2")
        z, wa, wb = _f32c(z), _f32c(wa), _f32c(wb)
        E = z.shape[0]
        Ra, Kd, L = wa.shape
        Rb = wb.shape[0]
        zf = 1 if z_factored else 0
        if zf and (coef is None or z.shape[1] * coef.shape[1] != Kd or z.shape[1] != 64):
            raise ValueError("rowgemm2: z_factored wants z[E,64], coef[E,MB], node matrices [R, 64 * MB, L]")
        if wa.shape[1:] != wb.shape[1:] or (not zf and z.shape[1] != Kd) or rowptr_a.numel() != Ra + 1 or rowptr_b.numel() != Rb + 1:
            raise ValueError("rowgemm2: z[E,Kd], wa[Ra,Kd,L], wb[Rb,Kd,L], rowptr_a[Ra+1], rowptr_b[Rb+1] expected")
        MB = 0
        if ba is not None:
            ba, bb = _f32c(ba), _f32c(bb)
            MB = ba.shape[1]
            coef = _f32c(coef) if coef is not None else None
            if ba.shape != (Ra, MB, L) or bb.shape != (Rb, MB, L) or (coef is not None and coef.shape != (E, MB)):
                raise ValueError("rowgemm2: bias blocks [R, MB, L] and coef [E, MB] expected")
        elif zf:
            coef = _f32c(coef)
            MB = coef.shape[1]
        out = torch.empty((E, L), dtype=torch.float32, device=z.device)
        L_ = hip.lib()
        st = _stream(z.device)
        timed("k_rowgemm_fwd", 2 * E * Kd * L,
              lambda: hip.check(L_.hg_rowgemm_fwd_bias(_ptr(z), _ptr(wa), _ptr(rowptr_a), _ptr(perm_a), Ra, Kd, L, _ptr(out), 0,
                                                       _ptr(ba), _ptr(coef), MB, zf, st), "hg_rowgemm_fwd_bias"))
        timed("k_rowgemm_fwd", 2 * E * Kd * L,
              lambda: hip.check(L_.hg_rowgemm_fwd_bias(_ptr(z), _ptr(wb), _ptr(rowptr_b), _ptr(perm_b), Rb, Kd, L, _ptr(out), 1,
                                                       _ptr(bb), _ptr(coef), MB, zf, st), "hg_rowgemm_fwd_bias"))
        ctx.save_for_backward(z, wa, wb, coef)
        ctx.idx = (rowptr_a, perm_a, rowptr_b, perm_b)
        ctx.MB, ctx.zf, ctx.has_bias = MB, zf, ba is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        z, wa, wb, coef = ctx.saved_tensors
        rowptr_a, perm_a, rowptr_b, perm_b = ctx.idx
        dout = _f32c(dout)
        Ra, Kd, L = wa.shape
        Rb = wb.shape[0]
        MB, zf, hb = ctx.MB, ctx.zf, ctx.has_bias
        need_z = ctx.needs_input_grad[0]
        dz = torch.empty_like(z) if need_z else None
        dwa = torch.empty_like(wa) if ctx.needs_input_grad[1] or hb else None
        dwb = torch.empty_like(wb) if ctx.needs_input_grad[4] or hb else None
        dba = torch.empty((Ra, MB, L), dtype=torch.float32, device=z.device) if hb else None
        dbb = torch.empty((Rb, MB, L), dtype=torch.float32, device=z.device) if hb else None
        L_ = hip.lib()
        st = _stream(z.device)
        E = z.shape[0]
        nf = (2 if need_z else 0) * E * Kd * L
        timed("k_rowgemm_bwd", nf + (2 * E * Kd * L if dwa is not None else 0),
              lambda: hip.check(L_.hg_rowgemm_bwd_bias(_ptr(z), _ptr(wa), _ptr(dout), _ptr(rowptr_a), _ptr(perm_a), Ra, Kd, L, _ptr(dz),
                                                       0, _ptr(dwa), _ptr(coef), MB, _ptr(dba), zf, st), "hg_rowgemm_bwd_bias"))
        timed("k_rowgemm_bwd", nf + (2 * E * Kd * L if dwb is not None else 0),
              lambda: hip.check(L_.hg_rowgemm_bwd_bias(_ptr(z), _ptr(wb), _ptr(dout), _ptr(rowptr_b), _ptr(perm_b), Rb, Kd, L, _ptr(dz),
                                                       1, _ptr(dwb), _ptr(coef), MB, _ptr(dbb), zf, st), "hg_rowgemm_bwd_bias"))
        return dz, dwa, None, None, dwb, None, None, dba, dbb, None, None


class _RowOuter(torch.autograd.Function):
    """out[r] = sum_{e in row r} a[e]^T (x) b[e]  ([R, Ka, Lb]; the dw half of hg_rowgemm_bwd), with
    da[e] = out_grad[row(e)] . b[e] (its dz half) and db[e] = a[e] . out_grad[row(e)] (hg_rowgemm_fwd)."""

    @staticmethod
    def forward(ctx, a, b, rowptr, perm):
        _require_gpu(a, "row_outer")
        a, b = _f32c(a), _f32c(b)
        E, Ka = a.shape
        Lb = b.shape[1]
        R = rowptr.numel() - 1
        if b.shape[0] != E:
            raise ValueError("row_outer: a[E,Ka], b[E,Lb], rowptr[R+1] expected")
        out = torch.empty((R, Ka, Lb), dtype=torch.float32, device=a.device)
        timed("k_rowgemm_bwd", 2 * E * Ka * Lb,
              lambda: hip.check(hip.lib().hg_rowgemm_bwd(_ptr(a), None, _ptr(b), _ptr(rowptr), _ptr(perm), R, Ka, Lb, None, 0,
                                                         _ptr(out), _stream(a.device)), "hg_rowgemm_bwd"))
        ctx.save_for_backward(a, b)
        ctx.idx = (rowptr, perm, R)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        rowptr, perm, R = ctx.idx
        dout = _f32c(dout)
        E, Ka = a.shape
        Lb = b.shape[1]
        L_, st = hip.lib(), _stream(a.device)
        da = db = None
        if ctx.needs_input_grad[0]:     # entries outside every row get no gradient
            da = torch.zeros_like(a) if perm is not None else torch.empty_like(a)
            hip.check(L_.hg_rowgemm_bwd(_ptr(a), _ptr(dout), _ptr(b), _ptr(rowptr), _ptr(perm), R, Ka, Lb, _ptr(da), 0, None, st),
                      "hg_rowgemm_bwd")
        if ctx.needs_input_grad[1]:
            db = torch.zeros_like(b) if perm is not None else torch.empty_like(b)
            hip.check(L_.hg_rowgemm_fwd(_ptr(a), _ptr(dout), _ptr(rowptr), _ptr(perm), R, Ka, Lb, _ptr(db), 0, st), "hg_rowgemm_fwd")
        return da, db, None, None


def row_outer(a, b, rowptr, perm=None):
    """out[r] = sum over the entries e of row r of a[e]^T (x) b[e]  -> [R, Ka, Lb].  With ``perm`` None the rows must cover
    every entry (rowptr[0] = 0, rowptr[R] = E: the receiver lists of the neighbour graph)."""
    return _RowOuter.apply(a, b, rowptr, perm)


class _PooledRadial(torch.autograd.Function):
    """p[n, lo] = sum_(li,k) y[n, li, k] W3[lo, li, k] + sum_li xbar[n, li] b3[lo, li] for the radial network's last Linear
    (weight [(lo, li), mid], bias [(lo, li)]: equiformer_layer.py:451-479) in the PARAMETER's own layout -- the weight viewed
    [lo, li * mid] is the [out, in] matrix of a plain Linear over y, so neither a re-laid copy of the 16 MB weight nor a
    transposition of its gradient exists, and the gradient is accumulated straight into the parameter's accumulator."""

    @staticmethod
    def forward(ctx, y, xbar, weight, bias, lo):
        n = y.shape[0]
        wv, b3 = weight.view(lo, -1), bias.view(lo, -1)
        y2 = y.reshape(n, -1)
        out = mm_nt(y2, wv)
        out = mm_nt(xbar, b3, d=out)
        ctx.save_for_backward(y2, xbar, weight, bias)
        ctx.lo, ctx.yshape = lo, y.shape
        return out

    @staticmethod
    def backward(ctx, dp):
        y2, xbar, weight, bias = ctx.saved_tensors
        lo = ctx.lo
        dp = _f32c(dp)
        wv, b3 = weight.view(lo, -1), bias.view(lo, -1)
        dy = mm_nn(dp, wv).view(ctx.yshape) if ctx.needs_input_grad[0] else None
        dxbar = mm_nn(dp, b3) if ctx.needs_input_grad[1] else None
        dw = db = None
        if ctx.needs_input_grad[2]:
            tgt = _acc_target(weight)
            if tgt is not None:
                tv = tgt.view(lo, -1)
                if gemm_supported(dp, y2, True, False) and gemm_out_ok(tv):
                    if _DEFER["active"]:
                        _DEFER["keep"].extend((dp, y2))
                    gemm(dp, y2, trans_a=True, trans_b=False, d=tv, out=tv)
                else:
                    tv.addmm_(dp.t(), y2)
            elif gemm_supported(dp, y2, True, False):
                dw = gemm(dp, y2, trans_a=True, trans_b=False).view_as(weight)
            else:
                dw = (dp.t() @ y2).view_as(weight)
        if ctx.needs_input_grad[3]:
            tgt = _acc_target(bias)
            if tgt is not None:
                tgt.view(lo, -1).addmm_(dp.t(), xbar)
            else:
                db = (dp.t() @ xbar).view_as(bias)
        return dy, dxbar, dw, db, None


def pooled_radial(y, xbar, weight, bias, lo: int):
    """See _PooledRadial; ``weight`` / ``bias`` are the PARAMETERS of the radial network's last Linear."""
    if torch.is_grad_enabled():
        if weight.requires_grad and weight.is_leaf:
            LINEAR_PARAMS[id(weight)] = weight
        if bias.requires_grad and bias.is_leaf:
            ACC_PARAMS[id(bias)] = bias
    return _PooledRadial.apply(y, xbar, weight, bias, lo)


class _RadialWeightLayout(torch.autograd.Function):
    """nn.Linear(mid, lo * li).weight [(lo, li), mid] -> [li, mid * lo_p] with columns ordered (k, lo) and lo zero-padded to lo_p:
    w.view(lo, li, mid).permute(1, 2, 0) (+ pad) as one tiled transposition each way (eqh_permute_tiles_f32)."""

    @staticmethod
    def forward(ctx, w, lo, li, mid, lo_p):
        _require_gpu(w, "radial_weight_layout")
        w = _f32c(w)
        out = torch.empty((li, mid * lo_p), dtype=torch.float32, device=w.device)
        # b = li, x = k (contiguous in the source), y = lo (contiguous in the destination)
        hip.check(hip.lib().eqh_permute_tiles_f32(_ptr(w), _ptr(out), mid, lo, lo_p, li, li * mid, mid, lo_p, mid * lo_p,
                                                  _stream(w.device)), "eqh_permute_tiles_f32")
        ctx.dims = (lo, li, mid, lo_p, tuple(w.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        lo, li, mid, lo_p, shape = ctx.dims
        dout = _f32c(dout)
        dw = torch.empty(shape, dtype=torch.float32, device=dout.device)
        # b = li, x = lo (contiguous in the source), y = k (contiguous in the destination)
        hip.check(hip.lib().eqh_permute_tiles_f32(_ptr(dout), _ptr(dw), lo, mid, mid, li, lo_p, mid * lo_p, li * mid, mid,
                                                  _stream(dout.device)), "eqh_permute_tiles_f32")
        return dw, None, None, None, None


def radial_weight_layout(w, lo: int, li: int, mid: int, lo_p: int):
    """See _RadialWeightLayout; ``w`` is the [lo * li, mid] weight PARAMETER of the radial network's last Linear."""
    return _RadialWeightLayout.apply(w, lo, li, mid, lo_p)


class _AttnPool(torch.autograd.Function):
    """Softmax over (self + 16 neighbour) slots of LeakyReLU-Linear logits, SiLU values, value Linear and the
    weighted sum, per node, one launch each way (eqf_attn_pool_fwd / _bwd, csrc/attn_pool.hip)."""

    @staticmethod
    def forward(ctx, me, edge, maskf, w_logit, wv, v_off, scale, slope, acc_params):
        _require_gpu(me, "attn_pool")
        me, edge, maskf = _f32c(me), _f32c(edge), _f32c(maskf)
        wl, wvv = _f32c(w_logit.detach()).reshape(-1), _f32c(wv.detach())
        N, D = me.shape
        K, V = maskf.shape[1], wvv.shape[0]
        out = torch.empty((N, V), dtype=torch.float32, device=me.device)
        attn = torch.empty((N, K + 1), dtype=torch.float32, device=me.device)
        hip.check(hip.lib().eqf_attn_pool_fwd(_ptr(me), _ptr(edge), _ptr(maskf), _ptr(wl), _ptr(wvv), N, K, D, v_off, V,
                                              float(scale), float(slope), _ptr(out), _ptr(attn), _stream(me.device)),
                  "eqf_attn_pool_fwd")
        ctx.save_for_backward(me, edge, maskf, wl, wvv, attn)
        ctx.meta = (v_off, float(scale), float(slope), w_logit.shape)
        ctx.acc = acc_params
        return out

    @staticmethod
    def backward(ctx, dout):
        me, edge, maskf, wl, wvv, attn = ctx.saved_tensors
        v_off, scale, slope, wl_shape = ctx.meta
        dout = _f32c(dout)
        N, D = me.shape
        K, V = maskf.shape[1], wvv.shape[0]
        dev = me.device
        dme, dedge = torch.empty_like(me), torch.empty_like(edge)
        L = hip.lib()
        ws_bytes = L.eqf_attn_pool_bwd_workspace_bytes(N)
        ws = _workspace(max(ws_bytes, 16), dev)
        tg = [_acc_target(p) for p in ctx.acc]            # (w_logit, wv)
        in_place = all(t is not None for t in tg)
        dwl, dwv = tg if in_place else (torch.empty(wl_shape, dtype=torch.float32, device=dev), torch.empty_like(wvv))
        hip.check(L.eqf_attn_pool_bwd(_ptr(me), _ptr(edge), _ptr(maskf), _ptr(wl), _ptr(wvv), _ptr(attn), _ptr(dout), N, K,
                                      D, v_off, V, scale, slope, _ptr(dme), _ptr(dedge), _ptr(dwl), _ptr(dwv),
                                      1 if in_place else 0, _ptr(ws), ws_bytes, _stream(dev)), "eqf_attn_pool_bwd")
        if in_place:
            return dme, dedge, None, None, None, None, None, None, None
        dwl, dwv = _hand_out([dwl, dwv], tg)
        return dme, dedge, None, dwl, dwv, None, None, None, None


def attn_pool_supported(me, edge, maskf, w_logit, wv, v_off) -> bool:
    return (me.is_cuda and me.dim() == 2 and edge.dim() == 2 and me.dtype == torch.float32 and maskf.dim() == 2
            and maskf.shape[1] == 16 and edge.shape[0] == me.shape[0] * 16 and edge.shape[1] == me.shape[1]
            and tuple(wv.shape) == (48, 48) and w_logit.numel() == 4 and me.shape[1] % 4 == 0 and v_off % 4 == 0
            and v_off >= 4 and v_off + 48 <= me.shape[1])


def attn_pool(me, edge, maskf, w_logit, wv, v_off: int, scale: float, slope: float):
    """out[n] = sum_s softmax_s(scale * w_logit . leaky_relu(x_s[:4])) * (silu(x_s[v_off:v_off+48]) @ wv) over the
    slots x_0 = me[n], x_1.. = edge[n*16 + s - 1] (valid where maskf[n, s-1] != 0; slot 0 always)."""
    if torch.is_grad_enabled():
        for w_ in (w_logit, wv):
            if w_.requires_grad and w_.is_leaf:
                LINEAR_PARAMS[id(w_)] = w_
    return _AttnPool.apply(me, edge, maskf, w_logit, wv, v_off, scale, slope, (w_logit, wv))


class _RmsNormRows(torch.autograd.Function):
    """t / max(||t|| * scale, eps) * g over dense rows (eqf_rms_norm_fwd / _bwd, csrc/rmsnorm.hip).  ``rep`` > 1: the rows
    are [d, rep] blocks flattened (a degree-l feature with its 2l + 1 components, equiformer_layer.py:194-225), the norm runs
    over the whole row with scale = d^-1/2 and channel c's gain multiplies its rep components."""

    @staticmethod
    def forward(ctx, x, g, eps, acc_param, rep=1, tiled=False):
        _require_gpu(x, "rms_norm_rows")
        x = _f32c(x)
        gv = _f32c(g.detach()).reshape(-1)
        if rep > 1:     # (tiled: the row is [rep, d] -- component-major -- instead of [d, rep])
            gv = gv.repeat(rep) if tiled else gv[:, None].expand(-1, rep).reshape(-1)
        R, C = x.shape
        out = torch.empty_like(x)
        scale = float(torch.tensor((C // rep) ** -0.5, dtype=torch.float32))
        hip.check(hip.lib().eqf_rms_norm_fwd(_ptr(x), _ptr(gv), R, C, scale, float(eps), _ptr(out), _stream(x.device)),
                  "eqf_rms_norm_fwd")
        ctx.save_for_backward(x, gv)
        ctx.eps, ctx.scale, ctx.acc, ctx.gshape, ctx.rep, ctx.tiled = float(eps), scale, acc_param, g.shape, rep, bool(tiled)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gv = ctx.saved_tensors
        dy = _f32c(dy)
        R, C = x.shape
        dx = torch.empty_like(x)
        L = hip.lib()
        ws_bytes = L.eqf_rms_norm_bwd_workspace_bytes(R, C)
        ws = _workspace(max(ws_bytes, 16), x.device)
        tg = _acc_target(ctx.acc)
        if ctx.rep > 1:       # the gain's gradient per component, then summed over the components of a channel
            dgf = torch.empty(C, dtype=torch.float32, device=x.device)
            hip.check(L.eqf_rms_norm_bwd(_ptr(x), _ptr(gv), _ptr(dy), R, C, ctx.scale, ctx.eps, _ptr(dx), _ptr(dgf), 0, _ptr(ws),
                                         ws_bytes, _stream(x.device)), "eqf_rms_norm_bwd")
            dg = (dgf.view(ctx.rep, -1).sum(0) if ctx.tiled else dgf.view(-1, ctx.rep).sum(1)).view(ctx.gshape)
            if tg is not None:
                tg.add_(dg)
                dg = None
            return dx, dg, None, None, None, None
        dg = tg if tg is not None else torch.empty(ctx.gshape, dtype=torch.float32, device=x.device)
        hip.check(L.eqf_rms_norm_bwd(_ptr(x), _ptr(gv), _ptr(dy), R, C, ctx.scale, ctx.eps, _ptr(dx), _ptr(dg),
                                     1 if tg is not None else 0, _ptr(ws), ws_bytes, _stream(x.device)), "eqf_rms_norm_bwd")
        return dx, (None if tg is not None else dg), None, None, None, None


def rms_norm_rows(x, g, eps: float, rep: int = 1, tiled: bool = False):
    """The Norm of the Equiformer (equiformer_layer.py:194-225) for 2-D fp32 rows; ``g`` is the ``transforms.l`` parameter
    [d, 1].  rep = 1: degree 0, rows [*, d].  rep = 2l + 1: degree l, rows [*, d * rep] (the [d, rep] block flattened; ``tiled``: the
    [rep, d] block -- component-major features)."""
    if torch.is_grad_enabled() and g.requires_grad and g.is_leaf:
        (LINEAR_PARAMS if g.dim() == 2 else ACC_PARAMS)[id(g)] = g
    return _RmsNormRows.apply(x, g, eps, g, rep, tiled)


class _RadialTrunk(torch.autograd.Function):
    """Linear(1,64) -> SiLU -> LN -> Linear(64,64) -> SiLU -> LN per edge, one launch each way
    (eqf_radial_trunk_fwd / _bwd, csrc/radial.hip); the backward recomputes the forward from ``dist``."""

    @staticmethod
    def forward(ctx, dist, eps, acc_params, w0, b0, g1, be1, w1, b1, g2, be2):
        _require_gpu(dist, "radial_trunk")
        dist = _f32c(dist).reshape(-1)
        ps = [_f32c(t.detach()) for t in (w0, b0, g1, be1, w1, b1, g2, be2)]
        E, M = dist.shape[0], ps[4].shape[0]
        out = torch.empty((E, M), dtype=torch.float32, device=dist.device)
        vp = ctypes.c_void_p * 8
        hip.check(hip.lib().eqf_radial_trunk_fwd(_ptr(dist), vp(*[t.data_ptr() for t in ps]), E, M, float(eps), _ptr(out),
                                                 _stream(dist.device)), "eqf_radial_trunk_fwd")
        ctx.dist, ctx.ps, ctx.eps, ctx.acc = dist, ps, float(eps), acc_params
        return out

    @staticmethod
    def backward(ctx, dh):
        dist, ps = ctx.dist, ctx.ps
        dh = _f32c(dh)
        E, M = dist.shape[0], ps[4].shape[0]
        dev = dist.device
        L = hip.lib()
        ws_bytes = L.eqf_radial_trunk_bwd_workspace_bytes(E)
        ws = _workspace(max(ws_bytes, 16), dev)
        tg = [_acc_target(p) for p in ctx.acc]          # order: w0, b0, g1, w1, b1, g2
        in_place = all(t is not None for t in tg)
        grads = tg if in_place else [torch.empty_like(ps[i]) for i in (0, 1, 2, 4, 5, 6)]
        vp8, vp6 = ctypes.c_void_p * 8, ctypes.c_void_p * 6
        hip.check(L.eqf_radial_trunk_bwd(_ptr(dist), vp8(*[t.data_ptr() for t in ps]), _ptr(dh), E, M, ctx.eps,
                                         vp6(*[g.data_ptr() for g in grads]), 1 if in_place else 0, _ptr(ws), ws_bytes,
                                         _stream(dev)), "eqf_radial_trunk_bwd")
        if in_place:
            return (None,) * 11
        dw0, db0, dg1, dw1, db1, dg2 = _hand_out(grads, tg)
        return None, None, None, dw0, db0, dg1, None, dw1, db1, dg2, None


def radial_trunk_supported(dist, lin0, ln1, lin1, ln2) -> bool:
    return (dist.is_cuda and not dist.requires_grad and dist.dtype == torch.float32 and lin0.in_features == 1
            and lin0.out_features == 64 and lin1.in_features == 64 and lin1.out_features == 64
            and lin0.bias is not None and lin1.bias is not None)


def radial_trunk(dist, lin0, ln1, lin1, ln2, eps: float = 1e-5):
    """[E] or [E,1] distances -> [E,64]: ``lin0`` Linear(1,64), ``ln1``/``ln2`` modules with ``gamma`` (parameter)
    and ``beta`` (buffer), ``lin1`` Linear(64,64); SiLU between Linear and LayerNorm (equiformer_layer.py:451-479)."""
    params = (lin0.weight, lin0.bias, ln1.gamma, lin1.weight, lin1.bias, ln2.gamma)
    if torch.is_grad_enabled():
        for w in params:
            if w.requires_grad and w.is_leaf:
                (LINEAR_PARAMS if w.dim() == 2 else ACC_PARAMS)[id(w)] = w
    return _RadialTrunk.apply(dist, eps, params, lin0.weight, lin0.bias, ln1.gamma, ln1.beta, lin1.weight, lin1.bias,
                              ln2.gamma, ln2.beta)


def rowgemm(z, w, rowptr, perm=None):
    """out[e, :] = z[e, :] @ w[row(e)]; rows given by rowptr (+ perm: entry ids per row)."""
    return _RowGemm.apply(z, w, rowptr, perm)


def rowgemm2(z, wa, rowptr_a, perm_a, wb, rowptr_b, perm_b, bias_a=None, bias_b=None, coef=None, z_factored=False):
    """rowgemm(z, wa, rowptr_a, perm_a) + rowgemm(z, wb, rowptr_b, perm_b) when BOTH groupings cover every entry
    of z (no entry outside all rows): one output buffer, the second pass accumulates.  ``bias_a`` / ``bias_b`` [R, MB, L]
    (+ ``coef`` [E, MB]): the rows' bias blocks added in the same launches (rowgemm_bias_supported(Kd, L)).  ``z_factored``: z is
    [E, 64] and the row operand's column (m, k) is coef[e, m] z[e, k] (node matrices [R, 64 MB, L], L <= 64)."""
    if bias_a is None and not z_factored:
        return _RowGemm2.apply(z, wa, rowptr_a, perm_a, wb, rowptr_b, perm_b)
    Kd = int(wa.shape[-2])
    if not rowgemm_bias_supported(Kd, int(wa.shape[-1])):
        # only the streaming kernels carry the bias blocks / the factored row operand; with EQH_ROWGEMM_LDS=1 (round 4's
        # workgroup-per-row kernels, kept for A/B runs) or a shape outside theirs the C ABI would answer EQH_ERR_ARG
        raise ValueError(f"rowgemm2: bias blocks / z_factored need the streaming row-product kernels, which do not take "
                         f"Kd={Kd}, L={int(wa.shape[-1])}" + (" while EQH_ROWGEMM_LDS=1 selects the LDS kernels"
                                                               if os.environ.get("EQH_ROWGEMM_LDS") == "1" else ""))
    return _RowGemm2.apply(z, wa, rowptr_a, perm_a, wb, rowptr_b, perm_b, bias_a, bias_b, coef, bool(z_factored))


def rowgemm_bias_supported(Kd: int, L: int) -> bool:
    return bool(hip.lib().hg_rowgemm_bias_supported(int(Kd), int(L)))


class _Pool3(torch.autograd.Function):
    """out[n, m, c] = sum_k w3[n, k, m] t[n, k, c] (eqf_pool3, component-major); w3 carries no gradient (geometry)."""

    @staticmethod
    def forward(ctx, t, w3):
        _require_gpu(t, "pool3")
        t, w3 = _f32c(t), _f32c(w3)
        N, K, C = t.shape
        if w3.shape != (N, K, 3) or C % 4:
            raise ValueError("pool3: t[N,K,C] (C a multiple of 4), w3[N,K,3] expected")
        out = torch.empty((N, 3, C), dtype=torch.float32, device=t.device)
        hip.check(hip.lib().eqf_pool3(_ptr(t), _ptr(w3), N, K, C, 0, _ptr(out), _stream(t.device)), "eqf_pool3")
        ctx.save_for_backward(w3)
        ctx.dims = (N, K, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (w3,) = ctx.saved_tensors
        N, K, C = ctx.dims
        dout = _f32c(dout)
        dt = torch.empty((N, K, C), dtype=torch.float32, device=dout.device)
        hip.check(hip.lib().eqf_pool3(_ptr(dout), _ptr(w3), N, K, C, 1, _ptr(dt), _stream(dout.device)), "eqf_pool3")
        return dt, None


def pool3(t, w3):
    """masked mean over the K neighbour slots of t[n, k, c] r_hat[n, k, m] -> [N, 3, C], component-major (w3 = mean_w * r_hat)."""
    return _Pool3.apply(t, w3)


def edge_geometry(pos, nbr, dist, radius: float, full_d: bool = False):
    """eqf_edge_geometry: (rhat [E,3] = D[:, m=0], maskf [N,K], mean_w [N,K], mean_w_rhat [N,K,3]) for the
    self-excluded neighbour lists of geo_knn(mode 1), plus the whole D[1] [E,3,3] with ``full_d``; see
    csrc/edge_geom.hip.  No gradient."""
    _require_gpu(pos, "edge_geometry")
    pos, dist = _f32c(pos.detach()), _f32c(dist)
    N, K = nbr.shape
    if nbr.dtype != torch.int32 or dist.shape != (N, K) or pos.shape != (N, 3) or K > 16:
        raise ValueError("edge_geometry: pos[N,3] fp32, nbr[N,K] int32, dist[N,K] fp32, K <= 16 expected")
    dev = pos.device
    rhat = torch.empty((N * K, 3), dtype=torch.float32, device=dev)
    maskf = torch.empty((N, K), dtype=torch.float32, device=dev)
    mean_w = torch.empty((N, K), dtype=torch.float32, device=dev)
    mean_w_rhat = torch.empty((N, K, 3), dtype=torch.float32, device=dev)
    dmat = torch.empty((N * K, 3, 3), dtype=torch.float32, device=dev) if full_d else None
    hip.check(hip.lib().eqf_edge_geometry(_ptr(pos), _ptr(nbr.contiguous()), _ptr(dist), N, K, float(radius), _ptr(rhat),
                                          _ptr(maskf), _ptr(mean_w), _ptr(mean_w_rhat), _ptr(dmat), _stream(dev)),
              "eqf_edge_geometry")
    return (rhat, maskf, mean_w, mean_w_rhat, dmat) if full_d else (rhat, maskf, mean_w, mean_w_rhat)
