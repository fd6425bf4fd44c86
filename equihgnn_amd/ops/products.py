"""Dense products: the x6 GEMM (csrc/gemm_x6.hip), the fp32-MFMA dense batch, the dispatch between them and the
library, weight-level small products.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from .. import hip
from ._base import (_f32c, _ptr, _row_view, _stream, _workspace, timed)


# ------------------------------------------------------------------------------------------------------------------
# fp32 GEMM on the bf16 matrix cores (csrc/gemm_x6.hip): every dense product of the models goes through here
# ------------------------------------------------------------------------------------------------------------------
@dataclass
class GemmProblem:
    """c = act(alpha * op(a) @ op(b) + beta * d + bias); see hg_gemm_x6_batch in include/equihgnn_hip.h.
    ``trans_a``: a is stored [K, M]; ``trans_b``: b is stored [N, K] (an nn.Linear weight).  ``d`` may be ``out``."""

    a: torch.Tensor
    b: torch.Tensor
    trans_a: bool = False
    trans_b: bool = True
    bias: Optional[torch.Tensor] = None
    d: Optional[torch.Tensor] = None
    alpha: float = 1.0
    beta: float = 1.0
    relu: bool = False
    out: Optional[torch.Tensor] = None
    mean8: Optional[tuple] = None     # (p, seed tensor or None): frame-mean epilogue, the output is [M / 8, N]
    presplit: Optional[bool] = None   # b split into bf16 planes once per call (hg_panel_pack); None: by the rule of _presplit_ok


GEMM_TILE = 0          # 0: chosen per launch; 64 / 128 force a block tile (tools/gemm_bench.py)
# B operand (the weight of a Linear) split into its bf16 planes ONCE per call by hg_panel_pack, the stagers of the x6 kernel then
# copy planes instead of splitting the weight again for every row tile of the output (two thirds of their VALU work at the
# 128 x 256 tile) -- VERDICT r5 #2's first lever.  Bit-identical results; measured (round 6, profiles/r06_ab_runs.txt) it is
# SLOWER: the planes are 1.5 x the weight's bytes through L2 -> registers -> LDS and the split it saves was not what bounded a
# step ([246 k x 256].[256 x 256] 324 against 239-255 us, [15 k x 256].[256 x 768] 52 against 47-50).  OFF unless
# EQH_X6_PRESPLIT=1; from this many rows of a (the pack is one ~5 us launch).
X6_PRESPLIT = os.environ.get("EQH_X6_PRESPLIT") == "1"
X6_PRESPLIT_MIN_ROWS = int(os.environ.get("EQH_X6_PRESPLIT_MIN_ROWS", 16384))


def _presplit_ok(M, N, K, trans_a, b) -> bool:
    return (X6_PRESPLIT and not trans_a and K % 32 == 0 and M >= X6_PRESPLIT_MIN_ROWS and N * K <= (1 << 23)
            and b.stride(-1) == 1 and b.stride(0) % 4 == 0)


def gemm_supported(a, b, trans_a=False, trans_b=True) -> bool:
    """Shapes hg_gemm_x6_batch takes in place: 2-D fp32 device operands whose contiguous extents are multiples of 4."""
    if not (a.is_cuda and a.dim() == 2 and b.dim() == 2 and a.dtype == torch.float32 and b.dtype == torch.float32):
        return False
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    if (b.shape[1] if trans_b else b.shape[0]) != k:
        return False
    return n % 4 == 0 and k > 0 and (m % 4 == 0 if trans_a else k % 4 == 0) and (k % 4 == 0 or not trans_b)


def gemm_out_ok(t) -> bool:
    """A destination / addend view hg_gemm_x6_batch can write: unit inner stride, row stride a multiple of 4 floats, 16-byte
    aligned (a column block of a wider gradient buffer may be neither)."""
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def gemm_batch(problems):
    """Up to 8 GEMMs with the same operand layout in ONE launch; returns the outputs."""
    n = len(problems)
    assert 1 <= n <= 8
    arr = (hip.HgGemmProblem * n)()
    keep, outs = [], []
    dev = problems[0].a.device
    flops = 0
    for i, pr in enumerate(problems):
        a, b = _row_view(pr.a, "gemm: a"), _row_view(pr.b, "gemm: b")
        M, K = (a.shape[1], a.shape[0]) if pr.trans_a else a.shape
        N = b.shape[0] if pr.trans_b else b.shape[1]
        if (b.shape[1] if pr.trans_b else b.shape[0]) != K:
            raise ValueError(f"gemm: op(a) is [{M}, {K}] but b is {tuple(b.shape)} (trans_b={pr.trans_b})")
        rows_out = M // 8 if pr.mean8 is not None else M
        out = pr.out if pr.out is not None else torch.empty((rows_out, N), dtype=torch.float32, device=dev)
        assert out.shape == (rows_out, N) and out.stride(1) == 1 and out.dtype == torch.float32
        q = arr[i]
        q.a, q.lda, q.b, q.ldb = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0)
        q.c, q.ldc, q.m, q.n, q.k = out.data_ptr(), out.stride(0), M, N, K
        pre = pr.presplit if pr.presplit is not None else _presplit_ok(M, N, K, pr.trans_a, b)
        if i == 0:
            pre_all = pre
        if pre_all and pre:      # (one B form per launch: the first problem decides)
            from .panel import panel_pack
            n_pad = (N + 31) // 32 * 32
            (img,) = panel_pack([(b, bool(pr.trans_b), n_pad)], k_major=True)
            keep.append(img)
            q.b_packed = img.data_ptr()
        elif pre_all:
            raise ValueError("gemm_batch: the problems of one launch must all take the pre-split form of b, or none")
        q.trans_a, q.trans_b, q.relu = int(bool(pr.trans_a)), int(bool(pr.trans_b)), int(bool(pr.relu))
        q.alpha, q.beta = float(pr.alpha), float(pr.beta)
        if pr.mean8 is not None:
            drop_p, seed = pr.mean8
            assert M % 8 == 0 and pr.d is None and not pr.relu and not pr.trans_a
            q.mean_rows, q.drop_p = 8, float(drop_p)
            if seed is not None:
                assert seed.dtype == torch.int64 and seed.is_cuda
                keep.append(seed)
                q.drop_seed = seed.data_ptr()
        if pr.d is not None:
            d = pr.d if pr.d is out else _row_view(pr.d, "gemm: d")
            assert d.shape == (M, N)
            keep.append(d)
            q.d, q.ldd = d.data_ptr(), d.stride(0)
        if pr.bias is not None:
            bias = _f32c(pr.bias)
            assert bias.numel() == N
            keep.append(bias)
            q.bias = bias.data_ptr()
        keep.extend((a, b))
        outs.append(out)
        flops += 2 * M * N * K
    L = hip.lib()
    ws_bytes = L.hg_gemm_x6_workspace_bytes(n, arr, GEMM_TILE)
    ws = _workspace(ws_bytes, dev) if ws_bytes else None
    timed("k_gemm_x6", flops, lambda: hip.check(L.hg_gemm_x6_batch(n, arr, GEMM_TILE, _ptr(ws), ws_bytes, _stream(dev)),
                                               "hg_gemm_x6_batch"))
    return outs


# Products of MANY rows with a small weight (K, N <= 256): the persistent row-panel kernel (csrc/panel.hip, k_panel_stream:
# weights pre-split once per call and streamed from L2, two A images, separate multiplying and row wavefronts) as an alternative
# to the tiled x6 kernel.  OFF by default (EQH_PANEL_STREAM=1 turns it on): measured in round 6 it is within +-15 % of x6 --
# [246 k x 256].[256 x 256] 237-258 against 222 us, [1.97 M x 128].[128 x 256] 1121-1168 against 982-1025, [246 k x 64].[64 x 256]
# dy W 87 against 103 -- because with all 256 CUs streaming the same 384 KB image every panel the XCD's L2 (2 KB/clk) is asked
# for exactly its peak: the MFMA loop takes 9.5-11.7 k cycles per panel against 6.9 k on the 148 CUs of the one-panel kernels
# (tools/stream_stamps.py; profiles/r06_ab_runs.txt).
STREAM_MIN_ROWS = int(os.environ.get("EQH_STREAM_MIN_ROWS", 32768))
USE_PANEL_STREAM = os.environ.get("EQH_PANEL_STREAM") == "1"


def _stream_ok(a, b, trans_a, trans_b, out, d, mean8) -> bool:
    if not USE_PANEL_STREAM or trans_a or mean8 is not None or a.dim() != 2 or b.dim() != 2 or a.shape[0] < STREAM_MIN_ROWS:
        return False
    K = a.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    if (b.shape[1] if trans_b else b.shape[0]) != K or N != 256 or K not in (64, 128, 256):
        return False
    for t in (out, d):
        if t is not None and not gemm_out_ok(t):
            return False
    return a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and b.stride(-1) == 1 and b.stride(0) % 4 == 0


def gemm(a, b, trans_a=False, trans_b=True, bias=None, d=None, alpha=1.0, beta=1.0, relu=False, out=None, mean8=None, presplit=None):
    """One GEMM through hg_gemm_x6_batch (see GemmProblem) -- or, many rows against a small weight, through the streaming
    row-panel kernel (panel_stream_gemm)."""
    if _stream_ok(a, b, trans_a, trans_b, out, d, mean8):
        from .panel import panel_stream_gemm
        M, K = a.shape
        N = b.shape[0] if trans_b else b.shape[1]
        return timed("k_panel_stream", 2 * M * N * K,
                     lambda: panel_stream_gemm(a, b, trans_b, alpha=alpha, d=d, beta=beta, bias=bias, relu=relu, out=out))
    return gemm_batch([GemmProblem(a, b, trans_a, trans_b, bias, d, alpha, beta, relu, out, mean8, presplit)])[0]


# Where the x6 kernel replaces the library GEMM (measured on MI355X against the TunableOp-selected hipBLASLt kernels,
# tools/gemm_bench.py -> profiles/r03_gemm_bench.txt): from ~4 M output elements per launch it is 7-37 % faster
# ([15 k x 256].[256 x 256] 18.6 against 20.0 us, [15 k x 512].[512 x 1024] 84 against 133 us, 150-215 against 105-135
# TFLOP/s at the Molecule3D / PCQM / Equiformer sizes); below that -- the [4.7 k x 256] x [256 x 256] products of a QM9
# batch, one workgroup per CU and eight K steps -- the tuned library is 10-15 % ahead.
X6_MIN_OUTPUTS = int(os.environ.get("EQH_X6_MIN_OUTPUTS", 3_500_000))
X6_MAX_K = 8192
X6_DEEP_ROWS = 32768        # weight gradients dY^T X over at least this many rows: the split-K form of the x6 kernel
X6_WGRAD_OUTPUTS = 4_000_000   # x^T dy products with at least this many outputs (and >= 1024 rows): 128 x 128 tiles
X6_WGRAD_ROWS = int(os.environ.get("EQH_X6_WGRAD_ROWS", 8192))        # deferred weight gradients from this many rows up go to it in batches of up to 8 products
USE_X6 = os.environ.get("EQH_GEMM", "auto") != "library"


# judge a product by its work (outputs x K / 256) instead of its outputs: measured on the K = 2176 input gradient of the EGNN's
# first edge Linear ([4.7 k x 2176] . [2176 x 256]): x6 47.6 us against the library's 49.1 -- not worth a rule of its own; off
X6_K_WEIGHT = os.environ.get("EQH_X6_KWEIGHT", "0") == "1"


def _x6_ok(a, b, trans_b, out_elems, k) -> bool:
    if X6_K_WEIGHT:
        out_elems = out_elems * max(k, 256) // 256
    return (USE_X6 and out_elems >= X6_MIN_OUTPUTS and k <= X6_MAX_K and a.is_cuda and a.dtype == torch.float32
            and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2 and gemm_supported(a, b, False, trans_b))


def mm_nt(x, w, bias=None, d=None, alpha: float = 1.0, beta: float = 1.0, relu: bool = False):
    """act(alpha * x @ w.T + beta * d + bias) for x [M, K], w [N, K] (an nn.Linear weight or a view of one): the x6
    kernel where it is the faster one, else the library GEMM."""
    if _x6_ok(x, w, True, x.shape[0] * w.shape[0], x.shape[1]):
        return gemm(x, w, trans_b=True, bias=bias, d=d, alpha=alpha, beta=beta, relu=relu)
    if d is None and alpha == 1.0:
        if relu and bias is not None:
            return torch._addmm_activation(bias, x, w.t(), use_gelu=False)
        y = F.linear(x, w, bias)
        return torch.relu(y) if relu else y
    if d is not None:
        y = torch.addmm(d, x, w.t(), beta=beta, alpha=alpha)
    else:       # (beta = 0: the input is ignored -- alpha rides the GEMM instead of a scaling kernel)
        y = torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device)
        torch.addmm(y, x, w.t(), beta=0.0, alpha=alpha, out=y)
    if bias is not None:
        y = y + bias
    return torch.relu(y) if relu else y


def mm_nn(x, w, d=None, alpha: float = 1.0, beta: float = 1.0, out=None):
    """alpha * x @ w + beta * d for x [M, K], w [K, N] (an input gradient dY W, or a weight stored [in, out]); ``out``
    (which may be ``d``: accumulate) receives the result."""
    if _x6_ok(x, w, False, x.shape[0] * w.shape[1], x.shape[1]):
        return gemm(x, w, trans_b=False, d=d, alpha=alpha, beta=beta, out=out)
    if d is None:
        if out is None:
            out = torch.empty((x.shape[0], w.shape[1]), dtype=x.dtype, device=x.device)
        if alpha == 1.0:
            return torch.mm(x, w, out=out)
        return torch.addmm(out, x, w, beta=0.0, alpha=alpha, out=out)   # (beta = 0: the input is ignored)
    if out is not None and out is d:
        return d.addmm_(x, w, beta=beta, alpha=alpha)
    y = torch.addmm(d, x, w, beta=beta, alpha=alpha)
    if out is not None:
        out.copy_(y)
        return out
    return y


def small_mm_batch(problems):
    """hg_small_mm_batch: up to 8 small products in one launch.  Each problem is a dict with
    a, b (2-D tensors, any strides), ta / tb (use the transpose), c (2-D out, unit inner stride), accumulate, alpha,
    and optionally u, v (c += u v^T), x, z, y, acc_y (y (+)= op(a) x + z), w (w += u)."""
    n = len(problems)
    assert 1 <= n <= 8
    arr = (hip.HgSmallMM * n)()
    keep = []
    for q, pr in zip(arr, problems):
        a, b, c = pr["a"], pr["b"], pr["c"]
        if pr.get("ta"):
            a = a.t()
        if pr.get("tb"):
            b = b.t()
        M, K = a.shape
        N = b.shape[1]
        assert b.shape[0] == K and c.shape == (M, N) and c.stride(1) == 1 and a.dtype == b.dtype == c.dtype == torch.float32
        q.a, q.a_rs, q.a_cs = a.data_ptr(), a.stride(0), a.stride(1)
        q.b, q.b_rs, q.b_cs = b.data_ptr(), b.stride(0), b.stride(1)
        q.c, q.ldc, q.m, q.n, q.k = c.data_ptr(), c.stride(0), M, N, K
        q.alpha, q.accumulate_c, q.accumulate_y = float(pr.get("alpha", 1.0)), int(bool(pr.get("accumulate"))), int(bool(pr.get("acc_y")))
        for name in ("u", "v", "x", "z", "y", "w"):
            t = pr.get(name)
            if t is not None:
                assert t.dim() == 1 and t.stride(0) == 1 and t.dtype == torch.float32
                setattr(q, name, t.data_ptr())
                keep.append(t)
        keep.extend((a, b, c))
    dev = problems[0]["c"].device
    hip.check(hip.lib().hg_small_mm_batch(n, arr, _stream(dev)), "hg_small_mm_batch")
