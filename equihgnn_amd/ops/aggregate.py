"""CSR builds, gather / segmented-reduce aggregations (torch_scatter.scatter and X[idx] of conv.py), embedding sums,
neighbour search.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from .. import hip
from ._base import (
    ACC_PARAMS, _acc_target, _as2d, _contiguous_run, _f32c, _ptr, _require_gpu, _stacked_view, _stream,
    _workspace, timed)


# --------------------------------------------------------------------------------------------
# CSR
# --------------------------------------------------------------------------------------------
@dataclass
class CSR:
    """rowptr[n_rows+1], perm[nnz] (entry ids grouped by row, ascending inside a row) and
    col[nnz] (the other coordinate of each entry), all int32 on the device."""

    rowptr: torch.Tensor
    perm: torch.Tensor
    col: torch.Tensor
    n_rows: int
    nnz: int
    entry_w: Optional[torch.Tensor] = None    # per-entry mean weights 1 / deg(col[q]) w.r.t. the TRANSPOSED CSR (entry_weights)
    entry_w_of: Optional[torch.Tensor] = None  # the rowptr tensor of the partner CSR `entry_w` was computed against


def csr_build(key: torch.Tensor, other: Optional[torch.Tensor], n_rows: int, col_div: int = 1, counts=None) -> CSR:
    """hg_csr_build: COO (int64 or int32 keys) -> CSR.  ``other`` int64 or None (then col = perm//col_div).  ``counts``
    (int32 keys only): an int32 tensor [n_rows + 2] already holding the keys' histogram (knn(..., counts=...)); it is
    consumed, and the build skips its clear and histogram launches."""
    _require_gpu(key, "csr_build")
    assert key.dtype in (torch.int64, torch.int32) and key.dim() == 1
    key = key.contiguous()
    if other is not None:
        assert other.dtype == torch.int64 and other.shape == key.shape
        other = other.contiguous()
    nnz = key.numel()
    dev = key.device
    rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    L = hip.lib()
    ws_bytes = L.hg_csr_build_workspace_bytes(nnz, n_rows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    if counts is not None:
        assert key.dtype == torch.int32 and counts.dtype == torch.int32 and counts.numel() >= n_rows + 2 and counts.is_contiguous()
        hip.check(L.hg_csr_build_i32_counted(_ptr(key), _ptr(other), nnz, n_rows, col_div, _ptr(rowptr), _ptr(perm), _ptr(col),
                                             _ptr(counts), _ptr(ws), ws_bytes, _stream(dev)), "hg_csr_build_i32_counted")
        return CSR(rowptr, perm[:nnz], col[:nnz], n_rows, nnz)
    build = L.hg_csr_build if key.dtype == torch.int64 else L.hg_csr_build_i32
    hip.check(build(_ptr(key), _ptr(other), nnz, n_rows, col_div, _ptr(rowptr), _ptr(perm),
                    _ptr(col), _ptr(ws), ws_bytes, _stream(dev)), "hg_csr_build")
    return CSR(rowptr, perm[:nnz], col[:nnz], n_rows, nnz)


def csr_build_batch(problems):
    """hg_csr_build_batch: several COO -> CSR builds in three launches.  ``problems`` is a list of
    (key int64, other int64 or None, n_rows[, col_div]); returns the CSRs in the same order."""
    n = len(problems)
    dev = problems[0][0].device
    _require_gpu(problems[0][0], "csr_build_batch")
    keys, others, outs = [], [], []
    for pr in problems:
        key, other, n_rows = pr[0], pr[1], int(pr[2])
        assert key.dtype == torch.int64 and key.dim() == 1
        key = key.contiguous()
        if other is not None:
            assert other.dtype == torch.int64 and other.shape == key.shape
            other = other.contiguous()
        nnz = key.numel()
        keys.append(key)
        others.append(other)
        outs.append((torch.empty(n_rows + 1, dtype=torch.int32, device=dev),
                     torch.empty(max(nnz, 1), dtype=torch.int32, device=dev),
                     torch.empty(max(nnz, 1), dtype=torch.int32, device=dev), n_rows, nnz))
    i64, i32, vp = ctypes.c_int64 * n, ctypes.c_int32 * n, ctypes.c_void_p * n
    nnz_a = i64(*[o[4] for o in outs])
    rows_a = i64(*[o[3] for o in outs])
    div_a = i32(*[(int(pr[3]) if len(pr) > 3 else 1) for pr in problems])
    key_a = vp(*[k.data_ptr() for k in keys])
    oth_a = vp(*[(o.data_ptr() if o is not None else None) for o in others])
    rp_a, pm_a, cl_a = vp(*[o[0].data_ptr() for o in outs]), vp(*[o[1].data_ptr() for o in outs]), vp(*[o[2].data_ptr() for o in outs])
    L = hip.lib()
    ws_bytes = L.hg_csr_build_batch_workspace_bytes(n, nnz_a, rows_a)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    hip.check(L.hg_csr_build_batch(n, key_a, oth_a, nnz_a, rows_a, div_a, rp_a, pm_a, cl_a, _ptr(ws), ws_bytes,
                                   _stream(dev)), "hg_csr_build_batch")
    return [CSR(o[0], o[1][:o[4]], o[2][:o[4]], o[3], o[4]) for o in outs]


def index_aux(vertex, edges, batch, n_nodes: int, n_edges: int, rowptr_v, rowptr_e, by_v: Optional[CSR] = None,
              by_e: Optional[CSR] = None, zero_buf=None):
    """hg_index_aux: (v32, e32, batch32 or None, has_v [N] float, has_e [M] float).  With the two CSRs it also fills
    their ``entry_w`` (per-entry mean weights with respect to each other's rows) in the same launch.  ``zero_buf``: an
    int32 tensor cleared by the same launch (the counters of the neighbour search that follows, knn(..., counts=...))."""
    _require_gpu(vertex, "index_aux")
    dev = vertex.device
    vertex, edges = vertex.contiguous(), edges.contiguous()
    nnz = vertex.numel()
    v32 = torch.empty(nnz, dtype=torch.int32, device=dev)
    e32 = torch.empty(nnz, dtype=torch.int32, device=dev)
    b32 = torch.empty(n_nodes, dtype=torch.int32, device=dev) if batch is not None else None
    has_v = torch.empty(n_nodes, dtype=torch.float32, device=dev)
    has_e = torch.empty(n_edges, dtype=torch.float32, device=dev)
    col_v = col_e = ew_v = ew_e = None
    if by_v is not None and by_e is not None:
        col_v, col_e = by_v.col, by_e.col
        ew_v = torch.empty(max(by_v.nnz, 1), dtype=torch.float32, device=dev)
        ew_e = torch.empty(max(by_e.nnz, 1), dtype=torch.float32, device=dev)
        by_v.entry_w, by_e.entry_w = ew_v, ew_e
        by_v.entry_w_of, by_e.entry_w_of = by_e.rowptr, by_v.rowptr
    hip.check(hip.lib().hg_index_aux(_ptr(vertex), _ptr(edges), nnz, _ptr(batch.contiguous()) if batch is not None else None,
                                     n_nodes, n_edges, _ptr(rowptr_v), _ptr(rowptr_e), _ptr(v32), _ptr(e32), _ptr(b32),
                                     _ptr(has_v), _ptr(has_e), _ptr(col_v), _ptr(col_e), _ptr(ew_v), _ptr(ew_e), _ptr(zero_buf),
                                     zero_buf.numel() if zero_buf is not None else 0, _stream(dev)),
              "hg_index_aux")
    return v32, e32, b32, has_v, has_e


def segment_reduce_bytes(nnz: int, n_out: int, C: int, has_idx: bool, has_ptr: bool, has_w: bool) -> int:
    """Algorithmic bytes of one hg_segment_reduce_f32 launch: 4C*nnz gathered rows + 4*nnz index +
    4*(R+1) rowptr + 4C*R output (+ 8*nnz for the mean-weight rowptr reads of the backward form)."""
    return (4 * C * nnz + 4 * C * n_out + (4 * nnz if has_idx else 0) + (4 * (n_out + 1) if has_ptr else 0)
            + (8 * nnz if has_w else 0))


def _segment_reduce(src, idx, rowptr, wptr, n_out, mean: bool) -> torch.Tensor:
    """Raw launch of hg_segment_reduce_f32 on 2-D ``src`` [rows, C]."""
    _require_gpu(src, "segment_reduce")
    src = _f32c(src)
    C = src.shape[-1]
    out = torch.empty((n_out, C), dtype=torch.float32, device=src.device)

    def work():
        nnz = int(idx.numel()) if idx is not None else (int(n_out) if rowptr is None else int(src.shape[0]))
        return segment_reduce_bytes(nnz, int(n_out), C, idx is not None, rowptr is not None, wptr is not None)

    timed("k_segment_reduce" + ("<weighted>" if wptr is not None else ""), work,
          lambda: hip.check(hip.lib().hg_segment_reduce_f32(_ptr(src), _ptr(idx), _ptr(rowptr), _ptr(wptr), _ptr(out),
                                                            n_out, C, 1 if mean else 0, _stream(src.device)),
                            "hg_segment_reduce_f32"))
    return out


def entry_weights(csr: CSR, csr_t: CSR) -> torch.Tensor:
    """w[q] = 1 / max(deg_t(csr.col[q]), 1): the mean weights of csr's entries with respect to the rows of ``csr_t``
    (hg_entry_weights), cached on ``csr``."""
    if csr.entry_w is None or csr.entry_w_of is not csr_t.rowptr:
        # (the cache is only valid for the partner it was built against: another csr_t has other degrees)
        w = torch.empty(max(csr.nnz, 1), dtype=torch.float32, device=csr.col.device)
        hip.check(hip.lib().hg_entry_weights(_ptr(csr.col), _ptr(csr_t.rowptr), csr.nnz, _ptr(w), _stream(w.device)),
                  "hg_entry_weights")
        csr.entry_w, csr.entry_w_of = w, csr_t.rowptr
    return csr.entry_w


def _segment_reduce_w(src, csr: CSR, entry_w) -> torch.Tensor:
    """out[r] = sum_{q in row r} entry_w[q] * src[csr.col[q]] (hg_segment_reduce_w_f32)."""
    src = _f32c(src)
    C = src.shape[-1]
    out = torch.empty((csr.n_rows, C), dtype=torch.float32, device=src.device)
    timed("k_segment_reduce<weighted>", segment_reduce_bytes(csr.nnz, csr.n_rows, C, True, True, False) + 4 * csr.nnz,
          lambda: hip.check(hip.lib().hg_segment_reduce_w_f32(_ptr(src), _ptr(csr.col), _ptr(csr.rowptr), _ptr(entry_w),
                                                              _ptr(out), csr.n_rows, C, _stream(src.device)),
                            "hg_segment_reduce_w_f32"))
    return out


# --------------------------------------------------------------------------------------------
# autograd functions
# --------------------------------------------------------------------------------------------
class _ReduceGathered(torch.autograd.Function):
    """out[r] = reduce_{q in row r of csr} src[csr.col[q]] — gather + scatter fused
    (conv.py:172-173: ``scatter(W1(X)[..., vertex, :], edges)``).  Backward is the same kernel on
    the transposed CSR with the mean weights of the forward rows."""

    @staticmethod
    def forward(ctx, src, csr: CSR, csr_t: CSR, mean: bool):
        ctx.csr, ctx.csr_t, ctx.mean = csr, csr_t, mean
        if mean:
            ctx.ew = entry_weights(csr_t, csr)     # once per batch (cached on the CSR): built here, outside the backward
        return _segment_reduce(src, csr.col, csr.rowptr, None, csr.n_rows, mean)

    @staticmethod
    def backward(ctx, dout):
        csr, csr_t = ctx.csr, ctx.csr_t
        if ctx.mean:
            return _segment_reduce_w(dout, csr_t, ctx.ew), None, None, None
        dsrc = _segment_reduce(dout, csr_t.col, csr_t.rowptr, None, csr_t.n_rows, False)
        return dsrc, None, None, None


class _ReduceEntries(torch.autograd.Function):
    """out[r] = reduce_{q in row r} src[csr.perm[q]] with ``src`` holding one row per entry
    (torch_scatter.scatter of a per-incidence matrix, conv.py:91-93,97,177).  Backward is a row
    gather: dsrc[p] = dout[key[p]] / max(deg(key[p]), 1)."""

    @staticmethod
    def forward(ctx, src, csr: CSR, key32, mean: bool):
        ctx.csr, ctx.key32, ctx.mean = csr, key32, mean
        return _segment_reduce(src, csr.perm, csr.rowptr, None, csr.n_rows, mean)

    @staticmethod
    def backward(ctx, dout):
        csr = ctx.csr
        dsrc = _segment_reduce(dout, ctx.key32, None, csr.rowptr if ctx.mean else None, csr.nnz, False)
        return dsrc, None, None, None


class _GatherRows(torch.autograd.Function):
    """out[p] = src[key32[p]] (X[..., vertex, :], conv.py:90,96,172,175,176).  Backward is the
    segmented sum over the CSR keyed by the same index (what ATen does with index_put_
    accumulate, 14 % of the reference's mhnnm CPU step)."""

    @staticmethod
    def forward(ctx, src, key32, csr: CSR):
        ctx.csr = csr
        return _segment_reduce(src, key32, None, None, key32.numel(), False)

    @staticmethod
    def backward(ctx, dout):
        csr = ctx.csr
        return _segment_reduce(dout, csr.perm, csr.rowptr, None, csr.n_rows, False), None, None


class _EmbedSum(torch.autograd.Function):
    """out[n] = sum_f table_f[x[n, f]] over F embedding tables given as separate [rows_f, C] weights
    (hg_embed_sum_fwd/bwd on their row-wise concatenation).  When the weights lie back to back in memory
    (the graphed trainer lays all parameters out in one flat buffer) the concatenation is a view, and when
    their gradient accumulators do too the backward adds straight into them: no cat, no split, no copy."""

    @staticmethod
    def forward(ctx, x, offsets, *tables):
        _require_gpu(tables[0], "embed_sum")
        x = x.contiguous()
        table = _stacked_view(tables) if _contiguous_run(tables) else torch.cat([_f32c(t) for t in tables], 0)
        N, F = x.shape
        C = table.shape[1]
        off = (ctypes.c_int32 * F)(*offsets)
        out = torch.empty((N, C), dtype=torch.float32, device=table.device)
        hip.check(hip.lib().hg_embed_sum_fwd(_ptr(x), _ptr(table), off, F, N, C, table.shape[0],
                                             _ptr(out), _stream(table.device)), "hg_embed_sum_fwd")
        ctx.save_for_backward(x)
        ctx.offsets, ctx.rows, ctx.tables = offsets, table.shape[0], tables
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dout = _f32c(dout)
        N, F = x.shape
        C = dout.shape[1]
        L = hip.lib()
        off = (ctypes.c_int32 * F)(*ctx.offsets)
        accs = [_acc_target(t) for t in ctx.tables]
        have = all(a is not None for a in accs)
        direct = have and _contiguous_run(accs)
        dtable = _stacked_view(accs) if direct else torch.empty((ctx.rows, C), dtype=torch.float32, device=dout.device)
        ws_bytes = L.hg_embed_sum_bwd_workspace_bytes(N, C, ctx.rows)
        ws = _workspace(ws_bytes, dout.device)
        hip.check(L.hg_embed_sum_bwd(_ptr(x), _ptr(dout), off, F, N, C, ctx.rows, _ptr(dtable), 1 if direct else 0,
                                     _ptr(ws), ws_bytes, _stream(dout.device)), "hg_embed_sum_bwd")
        if direct:
            return (None, None) + (None,) * len(ctx.tables)
        parts = torch.split(dtable, [t.shape[0] for t in ctx.tables], 0)
        if have:   # accumulators present but scattered (the trainer's probe pass): add piece by piece
            for a, g in zip(accs, parts):
                a.add_(g)
            return (None, None) + (None,) * len(ctx.tables)
        return (None, None) + tuple(parts)


def reduce_gathered(src, csr: CSR, csr_t: CSR, reduce: str = "mean"):
    s2, lead = _as2d(src)
    out = _ReduceGathered.apply(s2, csr, csr_t, reduce == "mean")
    return out.reshape(*lead, *out.shape)


def reduce_entries(src, csr: CSR, key32, reduce: str = "mean"):
    s2, lead = _as2d(src)
    out = _ReduceEntries.apply(s2, csr, key32, reduce == "mean")
    return out.reshape(*lead, *out.shape)


def gather_rows(src, key32, csr: CSR):
    s2, lead = _as2d(src)
    out = _GatherRows.apply(s2, key32, csr)
    return out.reshape(*lead, *out.shape)


def embed_sum(x, tables, offsets=None):
    """out[n] = sum_f tables[f][x[n, f]] (ogb AtomEncoder order).  ``tables``: one [rows, C] weight or a
    sequence of F of them (the PARAMETERS, so that their gradient accumulators can be found)."""
    if x.dim() == 1:
        x = x[:, None]
    if torch.is_tensor(tables):
        tables = (tables,)
    tables = tuple(tables)
    if offsets is None:
        offsets, run = [], 0
        for t in tables:
            offsets.append(run)
            run += t.shape[0]
    if torch.is_grad_enabled():
        for t in tables:
            if t.requires_grad and t.is_leaf:
                ACC_PARAMS[id(t)] = t
    return _EmbedSum.apply(x, tuple(int(o) for o in offsets), *tables)


# Measured on MI355X (k = 16, both kernels with the k-th-distance bound of round 2; mode 0 / mode 1, eager launches
# including the grid build): 4.7 k atoms brute 43 / 50 us, grid 97 / 99; 8.3 k (QM9-like) 170 / 205 vs 124 / 126; 9.1 k
# (PCQM-like) 185 / 223 vs 194 / 199; 15 k 332 / 413 vs 251 / 258; 31 k 1028 / 1347 vs 418 / 436.  The molecules of a
# batch overlap around the origin, so the central cells stay crowded at the finest grid the LDS counters allow: the grid
# pays from ~10 k atoms.
KNN_GRID_MIN_POINTS = 10240


def knn(pos, k: int, mode: int, n_box=None, algorithm: str = "auto", counts=None):
    """(nbr int32 [N,k], key fp32 [N,k]); mode 0 = EGNN (squared distance, self included), 1 = Equiformer /
    FAFormer (true distance, self excluded).  No gradient (the reference feeds ``pos`` as data).
    ``algorithm``: "grid" (geo_knn_grid: cell grid, O(N)), "brute" (geo_knn) or "auto"; the two give identical
    results.  ``n_box``: optional int32 device tensor [1], the number of leading points that define the grid's
    bounding box (the real atoms of a padded batch).  ``counts``: a ZEROED int32 tensor [>= N]; the brute-force search adds 1
    to counts[j] for every list entry j (the histogram csr_build(..., counts=...) starts from) and the call returns
    (nbr, key, True); the grid search ignores it and returns (nbr, key, False)."""
    _require_gpu(pos, "knn")
    pos = _f32c(pos.detach())
    N = pos.shape[0]
    nbr = torch.empty((N, k), dtype=torch.int32, device=pos.device)
    dist = torch.empty((N, k), dtype=torch.float32, device=pos.device)
    L = hip.lib()
    if algorithm == "auto":
        algorithm = "grid" if KNN_GRID_MIN_POINTS <= N <= L.geo_knn_grid_max_points() else "brute"
    if algorithm == "grid":
        ws_bytes = L.geo_knn_grid_workspace_bytes(N)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=pos.device)
        hip.check(L.geo_knn_grid(_ptr(pos), N, k, mode, _ptr(n_box), _ptr(nbr), _ptr(dist), _ptr(ws), ws_bytes,
                                 _stream(pos.device)), "geo_knn_grid")
    elif counts is not None:
        assert counts.dtype == torch.int32 and counts.numel() >= N and counts.is_contiguous()
        hip.check(L.geo_knn_counted(_ptr(pos), N, k, mode, _ptr(nbr), _ptr(dist), _ptr(counts), _stream(pos.device)),
                  "geo_knn_counted")
        return nbr, dist, True
    else:
        hip.check(L.geo_knn(_ptr(pos), N, k, mode, _ptr(nbr), _ptr(dist), _stream(pos.device)), "geo_knn")
    return (nbr, dist, False) if counts is not None else (nbr, dist)


def scatter(src, index, dim: int = -1, out=None, dim_size=None, reduce: str = "sum"):
    """Drop-in for ``torch_scatter.scatter`` at the reference's call sites (conv.py:3,91-93,97,
    173,177): 1-D int64 ``index`` along ``dim=-2``.  Builds the CSR on the fly; the model classes
    instead build it once per batch (HyperIndex) and call reduce_entries / reduce_gathered."""
    if out is not None:
        raise NotImplementedError("out= is not used by the reference")
    if dim not in (-2, src.dim() - 2) or index.dim() != 1:
        raise NotImplementedError("only the reference's pattern (1-D index, dim=-2) is supported")
    if reduce not in ("sum", "add", "mean"):
        raise ValueError(reduce)
    if dim_size is None:
        dim_size = int(index.max()) + 1  # device sync, as in torch_scatter
    csr = csr_build(index, None, dim_size)
    return reduce_entries(src, csr, index.to(torch.int32), "mean" if reduce == "mean" else "sum")
