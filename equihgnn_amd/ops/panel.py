"""Row-panel kernels (csrc/panel.hip): weights packed once per step into bf16 planes in MFMA operand order, conv-sized
dense products fused with the row-wise work either side of them.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import torch

from .. import hip
from ._base import (_ptr, _row_view, _stream)

PANEL_WIDTHS = (64, 128, 256)


def panel_supported(C: int) -> bool:
    return C in PANEL_WIDTHS


def panel_pack(items, out=None, k_major: bool = False):
    """Pack weights for the panel kernels in ONE launch.  ``items`` = [(w, trans)] or [[(w, trans), ...]]: a 2-D fp32 weight
    view ``w`` (unit inner stride) used as B[k][n] = w[n, k] (``trans`` True: x @ w.T) or B[k][n] = w[k, n] (False: dy @ w);
    an inner list stacks its weights along K in one image; (w, trans, n_pad): N zero-padded up to n_pad columns.  Returns one uint8 image tensor per item (views of ``out`` if
    given: a uint8 device buffer of at least panel_pack_bytes(items) bytes).  ``k_major``: the image order the x6 GEMM's pre-split
    B operand takes ([k / 32][tile][k half][plane][lane]) instead of the panel kernels' [tile][k / 16][plane][lane]."""
    L = hip.lib()
    groups = [it if isinstance(it, list) else [it] for it in items]
    sizes = []
    for g in groups:
        ks = [(it[0].shape[1] if it[1] else it[0].shape[0]) for it in g]
        ns = {(it[2] if len(it) > 2 else (it[0].shape[0] if it[1] else it[0].shape[1])) for it in g}
        if len(ns) != 1:
            raise ValueError("panel_pack: weights stacked along K must share their N")
        sizes.append((sum(ks), ns.pop()))
    total = sum(K * N * 6 for K, N in sizes)
    dev = groups[0][0][0].device
    if out is None:
        out = torch.empty(total, dtype=torch.uint8, device=dev)
    elif out.numel() < total or out.dtype != torch.uint8:
        raise ValueError("panel_pack: output buffer too small")
    n = sum(len(g) for g in groups)
    arr = (hip.HgPanelPack * n)()
    keep, views, off, i = [], [], 0, 0
    for g, (K, N) in zip(groups, sizes):
        img = out[off:off + K * N * 6]
        views.append(img)
        k0 = 0
        for it in g:
            w, tr = it[0], it[1]
            w = _row_view(w.detach(), "panel_pack: w")
            keep.append(w)
            kk = w.shape[1] if tr else w.shape[0]
            if kk % 16 or N % 32:
                raise ValueError(f"panel_pack: K = {kk} must be a multiple of 16 and N = {N} of 32")
            arr[i].w, arr[i].ld, arr[i].dst = w.data_ptr(), w.stride(0), img.data_ptr()
            arr[i].K, arr[i].N, arr[i].trans, arr[i].kstep0, arr[i].ksteps_total = kk, N, 1 if tr else 0, k0 // 16, K // 16
            arr[i].n_valid = w.shape[0] if tr else w.shape[1]
            arr[i].k_major = 1 if k_major else 0
            k0 += kk
            i += 1
        off += K * N * 6
    hip.check(L.hg_panel_pack(n, arr, _stream(dev)), "hg_panel_pack")
    return views


def panel_pack_bytes(items) -> int:
    groups = [it if isinstance(it, list) else [it] for it in items]
    return sum(sum((it[0].shape[1] if it[1] else it[0].shape[0]) for it in g)
               * (g[0][2] if len(g[0]) > 2 else (g[0][0].shape[0] if g[0][1] else g[0][0].shape[1])) * 6 for g in groups)


def panel_gemm(a, wpack, C: int, alpha: float = 1.0, d=None, beta: float = 1.0, bias=None, relu: bool = False, out=None):
    """act(alpha * a @ B + beta * d + bias) for a [rows, C] and a packed [C x C] image (hg_panel_gemm_f32)."""
    a = _row_view(a, "panel_gemm: a")
    rows = a.shape[0]
    if a.shape[1] != C or not panel_supported(C):
        raise ValueError(f"panel_gemm: a must be [rows, {C}] with C in {PANEL_WIDTHS}")
    if out is None:
        out = torch.empty((rows, C), dtype=torch.float32, device=a.device)
    if d is not None:
        d = _row_view(d, "panel_gemm: d")
    hip.check(hip.lib().hg_panel_gemm_f32(_ptr(a), a.stride(0), rows, C, _ptr(wpack), float(alpha),
                                          _ptr(d) if d is not None else None, d.stride(0) if d is not None else 0, float(beta),
                                          _ptr(bias) if bias is not None else None, 1 if relu else 0, _ptr(out), out.stride(0),
                                          _stream(a.device)), "hg_panel_gemm_f32")
    return out


def panel_stream_supported(K: int, N: int) -> bool:
    return bool(hip.lib().hg_panel_stream_supported(int(K), int(N)))


def panel_stream_gemm(a, w, trans_b: bool = True, alpha: float = 1.0, d=None, beta: float = 1.0, bias=None, relu: bool = False, out=None):
    """act(alpha * a @ op(w) + beta * d + bias) for MANY rows a [rows, K] and a small weight (``trans_b``: w [N, K], an
    nn.Linear weight used as x W^T; else w [K, N], an input gradient dY W), K in {64, 128, 256}, N in {128, 256}: the weight is
    packed into bf16 planes (one launch) and the persistent row-panel kernel streams the rows (hg_panel_stream_gemm_f32)."""
    a = _row_view(a, "panel_stream_gemm: a")
    rows, K = a.shape
    N = w.shape[0] if trans_b else w.shape[1]
    if (w.shape[1] if trans_b else w.shape[0]) != K or not panel_stream_supported(K, N):
        raise ValueError(f"panel_stream_gemm: a [{rows}, {K}] x weight {tuple(w.shape)} (trans_b={trans_b}) is not a supported shape")
    (img,) = panel_pack([(w, bool(trans_b))])
    if out is None:
        out = torch.empty((rows, N), dtype=torch.float32, device=a.device)
    if d is not None and d is not out:
        d = _row_view(d, "panel_stream_gemm: d")
    if bias is not None:
        bias = bias.detach().contiguous()
    hip.check(hip.lib().hg_panel_stream_gemm_f32(_ptr(a), a.stride(0), rows, K, N, _ptr(img), float(alpha),
                                                 _ptr(d) if d is not None else None, d.stride(0) if d is not None else 0, float(beta),
                                                 _ptr(bias) if bias is not None else None, 1 if relu else 0, _ptr(out), out.stride(0),
                                                 _stream(a.device)), "hg_panel_stream_gemm_f32")
    return out


_CP_PTRS = ("in0", "in1", "in2", "in3", "rowptr", "col", "wq", "w0", "w1", "w2", "w3", "b0", "g0", "be0", "b1", "g1", "be1",
            "bias_out", "out0", "out1", "out2", "out3", "out4", "out5", "slab", "slab2", "acc_out", "dbias", "dgamma", "dbeta",
            "dbias2", "dgamma2", "dbeta2", "g_inc", "be_inc", "out6", "signal")


def conv_panel(stage: int, rows: int, C: int, device, eps: float = 1e-5, scale: float = 1.0, relu: bool = False,
               acc_first: bool = False, tail: bool = False, accumulate: bool = False, ld0: int = 0, eps_inc: float = 1e-5,
               **tensors):
    """One hg_conv_panel stage (include/equihgnn_hip.h lists the operands of each); ``tensors``: name -> device tensor or None."""
    a = hip.HgConvPanel()
    a.rows, a.C, a.eps, a.scale, a.eps_inc = rows, C, float(eps), float(scale), float(eps_inc)
    a.relu, a.acc_first, a.tail, a.accumulate, a.ld0 = int(relu), int(acc_first), int(tail), int(accumulate), int(ld0)
    for k, t in tensors.items():
        if k not in _CP_PTRS:
            raise TypeError(f"conv_panel: unknown operand {k}")
        if t is not None:
            setattr(a, k, t.data_ptr())
    hip.check(hip.lib().hg_conv_panel(stage, a, _stream(device)), f"hg_conv_panel(stage {stage})")


def conv_panel_slab(rows: int, C: int, device):
    """A slab for the vector gradients of an HG_CONV_B3 / HG_CONV_B1 stage (parked until defer_flush while deferred)."""
    from ._base import _workspace
    return _workspace(max(hip.lib().hg_conv_panel_slab_bytes(rows, C), 16), device)
