"""Readout head (pool + output MLP + MSE in one launch) and the fused MSE loss.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from .. import hip
from ._base import (ACC_PARAMS, LINEAR_PARAMS, _acc_target, _f32c, _hand_out, _ptr, _require_gpu, _stream, _workspace)


class _MseLoss(torch.autograd.Function):
    """F.mse_loss(pred, target) (mean) with forward value and gradient from ONE launch (eqh_mse_fwd_bwd)
    instead of six tiny elementwise / reduction launches."""

    @staticmethod
    def forward(ctx, pred, target):
        _require_gpu(pred, "mse_loss")
        pred, target = _f32c(pred), _f32c(target)
        n = pred.numel()
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred)
        hip.check(hip.lib().eqh_mse_fwd_bwd(_ptr(pred), _ptr(target), n, _ptr(loss), _ptr(grad), _stream(pred.device)),
                  "eqh_mse_fwd_bwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        return grad * dloss, None


def mse_loss(pred, target):
    """mean((pred - target)^2) for 1-D fp32 device tensors of up to 65 536 values (a batch of molecules)."""
    if pred.is_cuda and pred.dtype == torch.float32 and 0 < pred.numel() <= 65536 and not target.requires_grad:
        return _MseLoss.apply(pred.reshape(-1), target.reshape(-1))
    return F.mse_loss(pred, target)


_READOUT_STATE = {}


def _readout_state(device):
    key = torch.device(device).index
    if key not in _READOUT_STATE:
        _READOUT_STATE[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _READOUT_STATE[key]


class _ReadoutMse(torch.autograd.Function):
    """pool -> MLP(C -> H -> H -> 1, LN) -> MSE with loss, dx and all parameter gradients from ONE launch
    (hg_readout_mse_f32).  The gradients are computed in forward(); backward() hands them out -- scaled by the
    incoming gradient unless ``unit_grad`` says it is the implicit 1 of ``loss.backward()``, in which case the
    parameter gradients may already have been added to their persistent accumulators."""

    @staticmethod
    def forward(ctx, x, rowptr, n_graphs, n_real, target, eps, unit_grad, params, *weights):
        _require_gpu(x, "readout_mse")
        dev = x.device
        x, target = _f32c(x), _f32c(target)
        ws_t = [_f32c(w.detach()) for w in weights]
        H, C = ws_t[0].shape
        L = hip.lib()
        vp = ctypes.c_void_p * 10
        y = torch.empty(n_graphs, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dx = torch.empty_like(x)
        ws_bytes = L.hg_readout_mse_workspace_bytes(n_graphs, C, H)
        ws = _workspace(ws_bytes, dev)
        tg = [_acc_target(w) for w in params] if unit_grad else [None]
        in_place = all(t is not None for t in tg)
        grads = tg if in_place else [torch.empty_like(w) for w in ws_t]
        hip.check(L.hg_readout_mse_f32(_ptr(x), _ptr(rowptr), n_graphs, n_real, C, H, vp(*[w.data_ptr() for w in ws_t]),
                                       float(eps), _ptr(target), _ptr(y), _ptr(loss), _ptr(dx),
                                       vp(*[g.data_ptr() for g in grads]), 1 if in_place else 0, _ptr(ws), ws_bytes,
                                       _ptr(_readout_state(dev)), _stream(dev)), "hg_readout_mse_f32")
        ctx.unit_grad, ctx.in_place = unit_grad, in_place
        ctx.targets = tg if (unit_grad and not in_place) else [None] * 10
        ctx.held = (dx,) if in_place else (dx, *grads)
        ctx.mark_non_differentiable(y)
        ctx.set_materialize_grads(False)        # (no zero-filled gradient tensor for the predictions: one launch per step)
        return loss, y

    @staticmethod
    def backward(ctx, dloss, _dy):
        held = ctx.held
        if dloss is None:
            return (None,) * 18
        if not ctx.unit_grad:
            held = tuple(h * dloss for h in held)
        dx = held[0]
        dws = (None,) * 10 if ctx.in_place else tuple(_hand_out([g.view_as(g) for g in held[1:]], ctx.targets))
        return (dx, None, None, None, None, None, None, None, *dws)


def readout_mse_supported(x, mlp) -> bool:
    """Whether ops.readout_mse takes this pooled-MLP head: 2-D fp32 device rows, MLP of three Linears with
    LayerNorm hidden layers and one output, no active dropout, widths the kernel is built for."""
    lins = getattr(mlp, "lins", None)
    if lins is None or len(lins) != 3 or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32:
        return False
    norms = mlp.normalizations
    if mlp.InputNorm or not all(isinstance(n, torch.nn.LayerNorm) for n in norms[1:]):
        return False
    if mlp.training and mlp.dropout > 0:
        return False
    H, C = lins[0].weight.shape
    if lins[1].weight.shape != (H, H) or lins[2].weight.shape != (1, H) or x.shape[1] != C:
        return False
    if norms[1].eps != norms[2].eps:
        return False
    return bool(hip.lib().hg_readout_mse_supported(C, H))


def readout_mse(x, pool_rowptr, mlp, target, n_real=None, unit_grad=False):
    """(loss, predictions) of the readout head: x [N, C] node rows, pool_rowptr int32 [B+1] (sorted ``batch``),
    ``mlp`` the output MLP, ``target`` [>= n_real]; loss = mean over the first n_real molecules."""
    n_graphs = pool_rowptr.shape[0] - 1
    n_real = n_graphs if n_real is None else int(n_real)
    lins, norms = mlp.lins, mlp.normalizations
    weights = (lins[0].weight, lins[0].bias, norms[1].weight, norms[1].bias, lins[1].weight, lins[1].bias,
               norms[2].weight, norms[2].bias, lins[2].weight, lins[2].bias)
    if torch.is_grad_enabled():
        for w in weights:
            if w.requires_grad and w.is_leaf:
                (LINEAR_PARAMS if w.dim() == 2 else ACC_PARAMS)[id(w)] = w
    return _ReadoutMse.apply(x, pool_rowptr, n_graphs, n_real, target, norms[1].eps, unit_grad, weights, *weights)
