"""Row kernels: per-incidence hidden layer + reduction, bias/ReLU/LayerNorm rows, gathered LayerNorm reduce, BatchNorm
/ LayerNorm rows, residual mix.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import torch

from .. import hip
from ._base import (
    LINEAR_PARAMS, _acc_target, _f32c, _hand_out, _note_acc, _ptr, _require_gpu, _rows_ld, _stream, _workspace, timed)
from .aggregate import (CSR, entry_weights, segment_reduce_bytes)
from .products import (mm_nn, mm_nt)
from .grads import (_wgrad_deferred, _wgrad_ok, colsum, wgrad)


def inc_fwd_bytes(nnz: int, rows: int, C: int) -> int:
    """Algorithmic bytes of k_inc_fwd (generic form): two gathered rows per incidence + one output row, three index words
    per incidence, the rowptr, gamma and beta."""
    return 4 * C * (2 * nnz + rows) + 12 * nnz + 4 * (rows + 1) + 8 * C


def inc_fwd_col_bytes(nnz: int, rows: int, C: int) -> int:
    """Algorithmic bytes of k_inc_fwd_col: the row operand is fetched ONCE per output row
    (incidence.hip: own row read before the entry loop), one gathered row per incidence, one output row; the col word of
    every incidence, the rowptr, gamma and beta: 4C (nnz + 2R) + 4 nnz + 4 (R + 1) + 8C."""
    return 4 * C * (nnz + 2 * rows) + 4 * nnz + 4 * (rows + 1) + 8 * C


class _IncidenceLnReduce(torch.autograd.Function):
    """S[r] = reduce_{p in row r} LayerNorm(relu(pa[ia[p]] + qb[ib[p]])) — one launch instead of
    gather, gather, add, ReLU, LayerNorm, segmented reduce (csrc/incidence.hip)."""

    @staticmethod
    def forward(ctx, pa, qb, gamma, beta, ia32, ib32, csr_a: CSR, csr_b: CSR, out_csr: CSR, okey32, mean, eps,
                acc_params):
        _require_gpu(pa, "incidence_ln_reduce")
        pa, qb, gamma, beta = _f32c(pa), _f32c(qb), _f32c(gamma), _f32c(beta)
        C = pa.shape[1]
        out = torch.empty((out_csr.n_rows, C), dtype=torch.float32, device=pa.device)
        if okey32 is ia32 or okey32 is ib32:
            # the output row is one operand's own index: (rowptr, col) of the output CSR says it all
            timed("k_inc_fwd_col", inc_fwd_col_bytes(out_csr.nnz, out_csr.n_rows, C), lambda: hip.check(hip.lib().hg_incidence_ln_reduce_fwd_col(
                _ptr(pa), _ptr(qb), _ptr(out_csr.rowptr), _ptr(out_csr.col), 1 if okey32 is ia32 else 0, _ptr(gamma),
                _ptr(beta), out_csr.n_rows, C, 1 if mean else 0, float(eps), _ptr(out), _stream(pa.device)),
                "hg_incidence_ln_reduce_fwd_col"))
        else:
            timed("k_inc_fwd", inc_fwd_bytes(out_csr.nnz, out_csr.n_rows, C), lambda: hip.check(hip.lib().hg_incidence_ln_reduce_fwd(
                _ptr(pa), _ptr(qb), _ptr(ia32), _ptr(ib32), _ptr(out_csr.rowptr), _ptr(out_csr.perm), _ptr(gamma),
                _ptr(beta), out_csr.n_rows, C, 1 if mean else 0, float(eps), _ptr(out), _stream(pa.device)),
                "hg_incidence_ln_reduce_fwd"))
        ctx.save_for_backward(pa, qb, gamma)
        ctx.meta = (ia32, ib32, csr_a, csr_b, out_csr, okey32, mean, eps)
        ctx.acc = acc_params
        return out

    @staticmethod
    def backward(ctx, ds):
        pa, qb, gamma = ctx.saved_tensors
        ia32, ib32, csr_a, csr_b, out_csr, okey32, mean, eps = ctx.meta
        ds = _f32c(ds)
        C = pa.shape[1]
        dev = pa.device
        dpa, dqb = torch.empty_like(pa), torch.empty_like(qb)
        g_acc, b_acc = (_acc_target(p) for p in ctx.acc)
        dgamma = g_acc if g_acc is not None else torch.empty_like(gamma)
        L = hip.lib()
        ws_bytes = L.hg_incidence_ln_reduce_bwd_workspace_bytes(csr_a.n_rows, C)
        ws = _workspace(ws_bytes, dev)
        # algorithmic bytes: each operand side walks every incidence once and gathers, per incidence, the OTHER
        # operand's row and the output-gradient row, reads its own row once and writes its own gradient row:
        # 4C (4 nnz + 2 (Ra + Rb)) + five index words per incidence and side + both rowptrs
        nnz_ = csr_a.nnz
        timed("k_inc_bwd_both", 4 * C * (4 * nnz_ + 2 * (csr_a.n_rows + csr_b.n_rows)) + 2 * 20 * nnz_
              + 4 * (csr_a.n_rows + csr_b.n_rows + 2),
              lambda: hip.check(L.hg_incidence_ln_reduce_bwd(
                  _ptr(pa), _ptr(qb), _ptr(ia32), _ptr(ib32), _ptr(csr_a.rowptr), _ptr(csr_a.perm), csr_a.n_rows,
                  _ptr(csr_b.rowptr), _ptr(csr_b.perm), csr_b.n_rows, _ptr(okey32), _ptr(out_csr.rowptr), _ptr(ds),
                  _ptr(gamma), C, 1 if mean else 0, float(eps), _ptr(dpa), _ptr(dqb), _ptr(dgamma),
                  1 if g_acc is not None else 0, _ptr(ws), ws_bytes, _stream(dev)), "hg_incidence_ln_reduce_bwd"))
        # d beta = sum_r w_r ds[r], w_r = [row non-empty] (mean) or the row length (sum)
        dbeta = colsum(ds, out_csr.rowptr, 1 if mean else 2, into=b_acc)
        return (dpa, dqb, None if g_acc is not None else dgamma, dbeta) + (None,) * 9


class _BiasReluLn(torch.autograd.Function):
    """LayerNorm(relu(h + bias)) over dense rows in one launch; the backward returns dh and, from the
    same pass, the bias / gamma / beta gradients (csrc/incidence.hip)."""

    @staticmethod
    def forward(ctx, h, bias, gamma, beta, eps, acc_params, fan=None):
        _require_gpu(h, "bias_relu_ln")
        h, bias, gamma, beta = _f32c(h), _f32c(bias), _f32c(gamma), _f32c(beta)
        R, C = h.shape
        out = torch.empty_like(h)
        hip.check(hip.lib().hg_bias_relu_ln_fwd(_ptr(h), _ptr(bias), _ptr(gamma), _ptr(beta), R, C, float(eps),
                                                _ptr(out), _stream(h.device)), "hg_bias_relu_ln_fwd")
        ctx.save_for_backward(h, bias, gamma)
        ctx.eps = eps
        ctx.acc = acc_params  # the Parameter objects (their accumulators are looked up at backward time)
        ctx.fan = fan
        return out

    @staticmethod
    def backward(ctx, dy):
        h, bias, gamma = ctx.saved_tensors
        dy = _f32c(dy)
        R, C = h.shape
        dh = torch.empty_like(h)
        L = hip.lib()
        ws_bytes = L.hg_bias_relu_ln_bwd_workspace_bytes(R, C)
        ws = _workspace(ws_bytes, h.device)
        tg = [_acc_target(p) for p in ctx.acc]
        acc = all(t is not None for t in tg)   # all three accumulate in place: nothing for autograd to add
        small = None if acc else torch.empty((3, C), dtype=torch.float32, device=h.device)
        o = tg if acc else list(small)
        fan = ctx.fan
        if fan is not None:     # dh is also the gradient of the fanned-out addend of this layer's input: summed in the kernel
            if fan.buf is None:
                fan.buf = torch.empty_like(h)
            hip.check(L.hg_bias_relu_ln_bwd_ex(_ptr(h), 1.0, None, _ptr(bias), _ptr(gamma), _ptr(dy), R, C, float(ctx.eps), _ptr(dh),
                                               _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), 1 if acc else 0, _ptr(ws), ws_bytes,
                                               _ptr(fan.buf), 1 if fan.n == 0 else 0, _stream(h.device)),
                      "hg_bias_relu_ln_bwd_ex")
            fan.n += 1
        else:
            hip.check(L.hg_bias_relu_ln_bwd(_ptr(h), _ptr(bias), _ptr(gamma), _ptr(dy), R, C, float(ctx.eps), _ptr(dh),
                                            _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), 1 if acc else 0, _ptr(ws), ws_bytes,
                                            _stream(h.device)), "hg_bias_relu_ln_bwd")
        if acc:
            return dh, None, None, None, None, None, None
        return (dh, *_hand_out(list(small), tg), None, None, None)


class _LinearAddReluLn(torch.autograd.Function):
    """LayerNorm(relu(scale * (x @ W.T) + c + bias)): the GEMM writes x @ W.T, the addend c (beta = 1 in _LinearAddC, which
    costs a copy of c into the GEMM's output per call) and the scale enter in the LayerNorm kernel
    (hg_bias_relu_ln_fwd_ex / _bwd_ex).  Backward: the kernel returns the gradient of the pre-activation; the two GEMMs
    take ``scale`` as their alpha; c's gradient goes to its GradFan (summed over the applications) or to autograd."""

    @staticmethod
    def forward(ctx, x, weight, c, scale, bias, gamma, beta, eps, fan, acc_params):
        _require_gpu(x, "linear_add_relu_ln")
        x, c, bias, gamma, beta = _f32c(x), _f32c(c), _f32c(bias), _f32c(gamma), _f32c(beta)
        h = mm_nt(x, weight)
        R, C = h.shape
        out = torch.empty_like(h)
        hip.check(hip.lib().hg_bias_relu_ln_fwd_ex(_ptr(h), float(scale), _ptr(c), _ptr(bias), _ptr(gamma), _ptr(beta), R, C,
                                                   float(eps), _ptr(out), _stream(x.device)), "hg_bias_relu_ln_fwd_ex")
        ctx.save_for_backward(x, weight, h, c, bias, gamma)
        ctx.meta = (float(scale), float(eps), fan)
        ctx.acc = acc_params
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, h, c, bias, gamma = ctx.saved_tensors
        a, eps, fan = ctx.meta
        dy = _f32c(dy)
        R, C = h.shape
        dpre = torch.empty_like(h)
        L = hip.lib()
        ws_bytes = L.hg_bias_relu_ln_bwd_workspace_bytes(R, C)
        ws = _workspace(ws_bytes, h.device)
        tg = [_acc_target(p) for p in ctx.acc]
        acc = all(t is not None for t in tg)
        small = None if acc else torch.empty((3, C), dtype=torch.float32, device=h.device)
        o = tg if acc else list(small)
        if fan is not None and fan.buf is None:
            fan.buf = torch.empty_like(h)
        hip.check(L.hg_bias_relu_ln_bwd_ex(_ptr(h), a, _ptr(c), _ptr(bias), _ptr(gamma), _ptr(dy), R, C, eps, _ptr(dpre),
                                           _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), 1 if acc else 0, _ptr(ws), ws_bytes,
                                           _ptr(fan.buf) if fan is not None else None, 1 if (fan is not None and fan.n == 0) else 0,
                                           _stream(h.device)), "hg_bias_relu_ln_bwd_ex")
        if fan is not None:
            fan.n += 1
        dx = mm_nn(dpre, weight, alpha=a) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            gbuf = getattr(weight, "_eqh_gbuf", None)
            if gbuf is not None and _wgrad_deferred(dpre, x, a, gbuf):
                pass
            elif gbuf is not None and _wgrad_ok(dpre, x):
                wgrad(dpre, x, a, into=gbuf)
            elif gbuf is not None:
                gbuf.addmm_(dpre.t(), x, alpha=a)
            elif _wgrad_ok(dpre, x):
                dw = wgrad(dpre, x, a)
            else:
                dw = torch.addmm(weight, dpre.t(), x, beta=0.0, alpha=a)
        dc = dpre if (fan is None and ctx.needs_input_grad[2]) else None
        if acc:
            return dx, dw, dc, None, None, None, None, None, None, None
        return (dx, dw, dc, None, *_hand_out(list(small), tg), None, None, None)


def linear_add_relu_ln(x, weight, c, scale, bias, gamma, beta, eps: float = 1e-5, fan=None):
    """bias_relu_ln(linear_add(x, weight, c, scale), bias, gamma, beta) with the addend and the scale applied inside the
    LayerNorm kernel (2-D fp32 x, c on the GPU); see _LinearAddReluLn."""
    if torch.is_grad_enabled() and weight.requires_grad and weight.is_leaf and not hasattr(weight, "_eqh_transient"):
        LINEAR_PARAMS[id(weight)] = weight
    _note_acc(bias, gamma, beta)
    return _LinearAddReluLn.apply(x, weight, c, scale, bias, gamma, beta, eps, fan, (bias, gamma, beta))


class _GatherLnReduce(torch.autograd.Function):
    """out[r] = gamma * reduce_{q in row r of csr} xhat(relu(h[csr.col[q]] + bias)) + beta * [..]: the hidden layer of an
    MLP on dense rows followed by the gathered reduction that is its only consumer, one launch each way
    (hg_gather_ln_reduce_*; csrc/incidence.hip)."""

    @staticmethod
    def forward(ctx, h, bias, gamma, beta, csr, csr_t, mean, eps, acc_params):
        _require_gpu(h, "gather_ln_reduce")
        h, bias, gamma, beta = _f32c(h), _f32c(bias), _f32c(gamma), _f32c(beta)
        R, C = h.shape
        if csr_t.n_rows != R:
            raise ValueError("gather_ln_reduce: the transposed CSR must have one row per row of h")
        out = torch.empty((csr.n_rows, C), dtype=torch.float32, device=h.device)
        timed("k_gather_ln_fwd", segment_reduce_bytes(csr.nnz, csr.n_rows, C, True, True, False),
              lambda: hip.check(hip.lib().hg_gather_ln_reduce_fwd(_ptr(h), _ptr(bias), _ptr(gamma), _ptr(beta),
                                                                  _ptr(csr.rowptr), _ptr(csr.col), csr.n_rows, C, int(mean),
                                                                  float(eps), _ptr(out), _stream(h.device)),
                                "hg_gather_ln_reduce_fwd"))
        ctx.save_for_backward(h, bias, gamma)
        ctx.eps, ctx.acc, ctx.csr_t = eps, acc_params, csr_t
        ctx.ew = entry_weights(csr_t, csr) if mean else None    # once per batch (cached on the CSR)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, bias, gamma = ctx.saved_tensors
        dout = _f32c(dout)
        R, C = h.shape
        t = ctx.csr_t
        dh = torch.empty_like(h)
        L = hip.lib()
        ws_bytes = L.hg_gather_ln_reduce_bwd_workspace_bytes(R, C)
        ws = _workspace(ws_bytes, h.device)
        tg = [_acc_target(p) for p in ctx.acc]
        acc = all(x is not None for x in tg)
        small = None if acc else torch.empty((3, C), dtype=torch.float32, device=h.device)
        o = tg if acc else list(small)
        timed("k_gather_ln_bwd", segment_reduce_bytes(t.nnz, R, C, True, True, False) + 4 * t.nnz + 4 * C * R,
              lambda: hip.check(L.hg_gather_ln_reduce_bwd(_ptr(h), _ptr(bias), _ptr(gamma), _ptr(dout), _ptr(t.rowptr),
                                                          _ptr(t.col), _ptr(ctx.ew), R, C, float(ctx.eps), _ptr(dh),
                                                          _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), 1 if acc else 0, _ptr(ws),
                                                          ws_bytes, _stream(h.device)), "hg_gather_ln_reduce_bwd"))
        if acc:
            return (dh,) + (None,) * 8
        return (dh, *_hand_out(list(small), tg), None, None, None, None, None)


def gather_ln_reduce(h, bias, gamma, beta, csr: CSR, csr_t: CSR, reduce: str = "mean", eps: float = 1e-5):
    """reduce_gathered(bias_relu_ln(h, bias, gamma, beta), csr, csr_t, reduce) in one launch each way (2-D h)."""
    _note_acc(bias, gamma, beta)
    return _GatherLnReduce.apply(h, bias, gamma, beta, csr, csr_t, reduce == "mean", eps, (bias, gamma, beta))


class _BatchNormRows(torch.autograd.Function):
    """Training-mode BatchNorm1d over rows with the batch statistics taken over the masked (real) rows only and the running
    buffers updated in the same launch (hg_batch_norm_rows_*; csrc/bn_rows.hip)."""

    @staticmethod
    def forward(ctx, x, mask, gamma, beta, running_mean, running_var, n_tracked, momentum, eps, acc_params, relu=False):
        _require_gpu(x, "batch_norm_rows")
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        R, C = x.shape
        m = _f32c(mask).reshape(-1) if mask is not None else None
        y = torch.empty_like(x)
        stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
        L = hip.lib()
        ws_bytes = L.hg_batch_norm_rows_workspace_bytes(R, C)
        ws = _workspace(max(ws_bytes, 16), x.device)
        hip.check(L.hg_batch_norm_rows_fwd(_ptr(x), _ptr(m), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                           _ptr(n_tracked), float(momentum), float(eps), R, C, _ptr(y), _ptr(stats[0]),
                                           _ptr(stats[1]), 1 if relu else 0, _ptr(ws), ws_bytes, _stream(x.device)),
                  "hg_batch_norm_rows_fwd")
        ctx.save_for_backward(x, m, gamma, stats, beta)
        ctx.acc, ctx.relu = acc_params, bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, m, gamma, stats, beta = ctx.saved_tensors
        R, C = x.shape
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        small = torch.empty((2, C), dtype=torch.float32, device=x.device)
        L = hip.lib()
        ws_bytes = L.hg_batch_norm_rows_workspace_bytes(R, C)
        ws = _workspace(max(ws_bytes, 16), x.device)
        hip.check(L.hg_batch_norm_rows_bwd(_ptr(x), _ptr(dy), _ptr(m), _ptr(gamma), _ptr(stats[0]), _ptr(stats[1]), R, C,
                                           _ptr(dx), _ptr(small[0]), _ptr(small[1]), _ptr(beta), 1 if ctx.relu else 0, _ptr(ws),
                                           ws_bytes, _stream(x.device)), "hg_batch_norm_rows_bwd")
        dgam, dbet = _hand_out(list(small), [_acc_target(p) for p in ctx.acc])
        return dx, None, dgam, dbet, None, None, None, None, None, None, None


def batch_norm_rows(x, mask, bn, relu: bool = False):
    """Training-mode ``bn`` (nn.BatchNorm1d with running statistics) on 2-D fp32 rows ``x`` with the statistics over the rows
    where ``mask`` [R, 1] is > 0 (None: all rows); running_mean / running_var / num_batches_tracked are updated in place.
    ``relu``: relu(bn(x)) in the same launches."""
    _note_acc(bn.weight, bn.bias)
    return _BatchNormRows.apply(x, mask, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                bn.momentum, bn.eps, (bn.weight, bn.bias), bool(relu))


def batch_norm_rows_supported(x, bn) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and bn.training and bn.affine
            and bn.track_running_stats and bn.momentum is not None and x.shape[0] > 1)


class _LayerNormRows(torch.autograd.Function):
    """Plain LayerNorm over dense rows; one launch each way, dgamma/dbeta from the backward pass.  With ``passthrough`` the
    node also returns x itself for x's OTHER consumer (a residual), whose gradient the backward kernel adds to dx in the
    same pass instead of an autograd add over the [rows, C] tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, acc_params, passthrough=False):
        _require_gpu(x, "layer_norm_rows")
        xin = x
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        R, C = x.shape
        out = torch.empty_like(x)
        hip.check(hip.lib().hg_layer_norm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), R, C, float(eps), _ptr(out),
                                              _stream(x.device)), "hg_layer_norm_fwd")
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.acc = acc_params
        if passthrough:
            ctx.set_materialize_grads(False)
            return out, xin.view_as(xin)
        return out

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, gamma = ctx.saved_tensors
        if dy is None:
            return dpass, None, None, None, None, None
        R, C = x.shape
        dy, ld = _rows_ld(dy.reshape(R, C))          # (a column block of a wider gradient -- the backward of a cat -- is read in place)
        add = _f32c(dpass).reshape(R, C) if dpass is not None else None
        dx = torch.empty_like(x)
        L = hip.lib()
        ws_bytes = L.hg_layer_norm_bwd_workspace_bytes(R, C)
        ws = _workspace(ws_bytes, x.device)
        tg = [_acc_target(p) for p in ctx.acc]
        if all(t is not None for t in tg):
            hip.check(L.hg_layer_norm_bwd(_ptr(x), _ptr(gamma), _ptr(dy), ld, _ptr(add), R, C, float(ctx.eps), _ptr(dx), _ptr(tg[0]),
                                          _ptr(tg[1]), 1, _ptr(ws), ws_bytes, _stream(x.device)), "hg_layer_norm_bwd")
            return dx, None, None, None, None, None
        small = torch.empty((2, C), dtype=torch.float32, device=x.device)
        hip.check(L.hg_layer_norm_bwd(_ptr(x), _ptr(gamma), _ptr(dy), ld, _ptr(add), R, C, float(ctx.eps), _ptr(dx), _ptr(small[0]),
                                      _ptr(small[1]), 0, _ptr(ws), ws_bytes, _stream(x.device)), "hg_layer_norm_bwd")
        return (dx, *_hand_out(list(small), tg), None, None, None)


# --------------------------------------------------------------------------------------------
# public functional API
# --------------------------------------------------------------------------------------------
def incidence_ln_reduce(pa, qb, gamma, beta, ia32, ib32, csr_a: CSR, csr_b: CSR, out_csr: CSR, okey32,
                        reduce: str = "mean", eps: float = 1e-5):
    """reduce_{p in out row} LayerNorm(relu(pa[ia[p]] + qb[ib[p]])); csr_a / csr_b are the incidence
    CSRs keyed by ia / ib (needed by the backward), out_csr the one keyed by okey32."""
    _note_acc(gamma, beta)
    return _IncidenceLnReduce.apply(pa, qb, gamma, beta, ia32, ib32, csr_a, csr_b, out_csr, okey32,
                                    reduce == "mean", eps, (gamma, beta))


def bias_relu_ln(h, bias, gamma, beta, eps: float = 1e-5, fan=None):
    """LayerNorm(relu(h + bias)) for 2-D ``h`` [rows, C].  ``fan``: see linear_add / GradFan (h = linear_add(..., c, fan=fan))."""
    _note_acc(bias, gamma, beta)
    return _BiasReluLn.apply(h, bias, gamma, beta, eps, (bias, gamma, beta), fan)


class _ResidualMix(torch.autograd.Function):
    """c = a * X0 + (1 - a) * w_r * bias (hg_residual_mix_f32); backward: dX0 = a * dc and the bias gradient
    as a scaled, row-weighted column sum (batched with the other bias gradients of the step)."""

    @staticmethod
    def forward(ctx, x0, bias, rowptr, weight_mode, alpha, bias_param, passthrough=False):
        _require_gpu(x0, "residual_mix")
        x0, bias = _f32c(x0), _f32c(bias)
        R, C = x0.shape
        out = torch.empty_like(x0)
        hip.check(hip.lib().hg_residual_mix_f32(_ptr(x0), _ptr(bias), _ptr(rowptr), weight_mode, float(alpha), R, C,
                                                _ptr(out), _stream(x0.device)), "hg_residual_mix_f32")
        ctx.rowptr, ctx.mode, ctx.alpha, ctx.bias_param = rowptr, weight_mode, float(alpha), bias_param
        ctx.set_materialize_grads(False)
        if passthrough:     # x0 again, for its OTHER consumer: both gradients then meet here, in one kernel
            return out, x0.view_as(x0)
        return out

    @staticmethod
    def backward(ctx, dc, dpass=None):
        dx0 = db = None
        if dc is not None:
            dc = _f32c(dc)
            if ctx.needs_input_grad[0]:
                dx0 = dc * ctx.alpha if dpass is None else torch.add(dpass, dc, alpha=ctx.alpha)
            if ctx.needs_input_grad[1]:
                db = colsum(dc, ctx.rowptr, ctx.mode, into=_acc_target(ctx.bias_param), scale=1.0 - ctx.alpha)
        elif dpass is not None and ctx.needs_input_grad[0]:
            dx0 = dpass
        return dx0, db, None, None, None, None, None


def residual_mix(x0, bias, rowptr, weight_mode: int, alpha: float, passthrough: bool = False):
    """alpha * x0 + (1 - alpha) * w_r * bias for 2-D x0 [rows, C] (C % 4 == 0); w_r from the int32 CSR ``rowptr``
    (weight_mode 1: [row non-empty], 2: row length).  ``passthrough``: also return x0 itself as a second output of the
    same autograd node -- hand THAT to x0's other consumer and the two gradients of x0 are combined by one kernel here
    instead of a multiply plus autograd's add."""
    _note_acc(bias)
    return _ResidualMix.apply(x0, bias, rowptr, weight_mode, alpha, bias, passthrough)


def layer_norm_rows(x, gamma, beta, eps: float = 1e-5, passthrough: bool = False):
    """nn.LayerNorm over the last dim of 2-D ``x`` [rows, C] (C % 4 == 0, C <= 1024).  ``passthrough``: returns
    (LayerNorm(x), x) -- see _LayerNormRows."""
    _note_acc(gamma, beta)
    return _LayerNormRows.apply(x, gamma, beta, eps, (gamma, beta), passthrough)
