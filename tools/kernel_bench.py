#!/usr/bin/env python3
"""Per-kernel microbenchmark of libequihgnn_hip.so at the BASELINE shapes (one process, HIP events
on the launching stream, N back-to-back launches behind a busy prefix so the queue is GPU-bound).

    python tools/kernel_bench.py [--batch 256] [--flavour qm9] [--hidden 256] [--json out.json]

Prints one row per kernel: microseconds, algorithmic bytes / FLOPs (DESIGN.md §4 formulas), achieved
GB/s or TFLOP/s and the fraction of the bounding peak (HBM 8 TB/s, fp32 MFMA 157.3 TFLOP/s).
Working sets at these shapes are L2/Infinity-Cache resident: the bandwidth column is an on-chip
rate, the cache-exceeding figure is bench.py's `roofline.saturation`.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from equihgnn_amd import hip, ops  # noqa: E402
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch  # noqa: E402
from equihgnn_amd.index import HyperIndex  # noqa: E402

HBM, MFMA32 = 8000.0, 157.3


def timed(fn, reps=30):
    dev = torch.device("cuda")
    busy = torch.randn(4096, 4096, device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(busy, busy)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # microseconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--flavour", default="qm9")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda")
    C = a.hidden
    raw = synth_batch(a.batch, 2000, a.flavour)
    b = pad_batch(raw, *bucket_sizes(raw.num_nodes, raw.num_hyperedges, raw.nnz)).to(dev)
    ix = HyperIndex.from_batch(b)
    N, M, nnz = ix.N, ix.M, ix.nnz
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(N, C, device=dev, generator=g)
    E = torch.randn(M, C, device=dev, generator=g)
    rows = []

    def add(name, us, nbytes=None, flops=None):
        r = {"kernel": name, "us": round(us, 2)}
        if nbytes:
            r.update(bytes=int(nbytes), GBps=round(nbytes / us / 1e3, 1), frac_hbm=round(nbytes / us / 1e3 / HBM, 3))
        if flops:
            r.update(flops=int(flops), TFLOPs=round(flops / us / 1e6, 2), frac_mfma=round(flops / us / 1e6 / MFMA32, 3))
        rows.append(r)

    # index build
    add("hg_csr_build (by hyperedge)", timed(lambda: ops.csr_build(b.edge_index1, b.edge_index0, M)),
        nbytes=28 * nnz + 12 * M)
    add("geo_knn mode 0 (k=16)", timed(lambda: ops.knn(b.pos, 16, 0)), flops=8.0 * N * N)
    nbr, d2, csr_t = ix.knn(b.pos, 16, 0)
    # aggregation kernels
    seg_b = lambda n, r: 4 * C * n + 4 * n + 4 * (r + 1) + 4 * C * r
    add("hg_segment_reduce fwd v->e mean", timed(lambda: ops._segment_reduce(X, ix.by_e.col, ix.by_e.rowptr, None, M, True)),
        nbytes=seg_b(nnz, M))
    add("hg_segment_reduce bwd e->v weighted",
        timed(lambda: ops._segment_reduce(E, ix.by_v.col, ix.by_v.rowptr, ix.by_e.rowptr, N, False)),
        nbytes=seg_b(nnz, N) + 8 * nnz)
    add("hg_segment_reduce pool", timed(lambda: ops._segment_reduce(X, ix.pool.perm, ix.pool.rowptr, None, ix.B, False)),
        nbytes=seg_b(N, ix.B))
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    L = hip.lib()
    S = torch.empty(N, C, device=dev)
    p, st = ops._ptr, ops._stream(dev)
    add("hg_incidence_ln_reduce_fwd (e->v)", timed(lambda: L.hg_incidence_ln_reduce_fwd(
        p(X), p(E), p(ix.v32), p(ix.e32), p(ix.by_v.rowptr), p(ix.by_v.perm), p(gamma), p(beta), N, C, 1, 1e-5, p(S), st)),
        nbytes=4 * C * (2 * nnz + N) + 12 * nnz + 4 * (N + 1))
    dpa, dqb, dg = torch.empty_like(X), torch.empty_like(E), torch.empty(C, device=dev)
    wsb = L.hg_incidence_ln_reduce_bwd_workspace_bytes(N, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    add("hg_incidence_ln_reduce_bwd (both sides)", timed(lambda: L.hg_incidence_ln_reduce_bwd(
        p(X), p(E), p(ix.v32), p(ix.e32), p(ix.by_v.rowptr), p(ix.by_v.perm), N, p(ix.by_e.rowptr), p(ix.by_e.perm), M,
        p(ix.v32), p(ix.by_v.rowptr), p(S), p(gamma), C, 1, 1e-5, p(dpa), p(dqb), p(dg), 0, p(ws), wsb, st)),
        nbytes=2 * (4 * C * 3 * nnz) + 4 * C * (N + M) + 32 * nnz)
    out = torch.empty_like(X)
    add("hg_bias_relu_ln_fwd", timed(lambda: L.hg_bias_relu_ln_fwd(p(X), p(beta), p(gamma), p(beta), N, C, 1e-5, p(out), st)),
        nbytes=8 * C * N)
    small = torch.empty(3 * C, device=dev)
    wsb2 = L.hg_bias_relu_ln_bwd_workspace_bytes(N, C)
    ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
    add("hg_bias_relu_ln_bwd", timed(lambda: L.hg_bias_relu_ln_bwd(p(X), p(beta), p(gamma), p(S), N, C, 1e-5, p(out), p(small),
                                                                     p(small[C:]), p(small[2 * C:]), 0, p(ws2), wsb2, st)),
        nbytes=12 * C * N)
    wsb3 = L.hg_colsum_workspace_bytes(N, C)
    ws3 = torch.empty(wsb3, dtype=torch.uint8, device=dev)
    add("hg_colsum_f32", timed(lambda: L.hg_colsum_f32(p(X), None, 0, 1.0, N, C, 0, p(dg), p(ws3), wsb3, st)), nbytes=4 * C * N)
    xi = torch.stack([torch.randint(0, d, (N,), device=dev, generator=g) for d in (119, 5, 12, 12, 10, 6, 6, 2, 2)], 1)
    table = torch.randn(174, C, device=dev, generator=g)
    offs = (0, 119, 124, 136, 148, 158, 164, 170, 172)
    add("hg_embed_sum_fwd", timed(lambda: ops.embed_sum(xi, table, offs)), nbytes=(9 * 4 * C + 72 + 4 * C) * N)
    dtab = torch.empty_like(table)
    off9 = (ctypes.c_int32 * 9)(*offs)
    wsb5 = L.hg_embed_sum_bwd_workspace_bytes(N, C, 174)
    ws5 = torch.empty(wsb5, dtype=torch.uint8, device=dev)
    add("hg_embed_sum_bwd (+ slab reduction)", timed(lambda: L.hg_embed_sum_bwd(p(xi), p(X), off9, 9, N, C, 174, p(dtab), 0,
                                                                               p(ws5), wsb5, st)), nbytes=9 * 4 * C * N)
    # fused EGNN edge kernels
    H = 2 * (2 * C + 1)
    Hp = H + (-H) % 64
    ab = torch.randn(N, 2 * Hp, device=dev, generator=g)
    wd = torch.randn(Hp, device=dev, generator=g) * 0.1
    w2 = torch.randn(16, Hp, device=dev, generator=g) / Hp ** 0.5
    b2 = torch.zeros(16, device=dev)
    m, pre2 = torch.empty(N, 16, device=dev), torch.empty(N, 16, 16, device=dev)
    add("egnn_edge_fwd", timed(lambda: L.egnn_edge_fwd(p(ab), p(wd), p(w2), p(b2), p(nbr), p(d2), N, Hp, p(m), p(pre2), st)),
        nbytes=4 * Hp * N * 17, flops=N * 16 * Hp * (2 * 16 + 12))
    dm = torch.randn(N, 16, device=dev, generator=g)
    dab, dwd, dw2, dpre2 = torch.empty_like(ab), torch.empty_like(wd), torch.empty_like(w2), torch.empty_like(pre2)
    wsb4 = L.egnn_edge_bwd_workspace_bytes(N, Hp)
    ws4 = torch.empty(wsb4, dtype=torch.uint8, device=dev)
    add("egnn_edge_bwd (recv + send + slabs)", timed(lambda: L.egnn_edge_bwd(
        p(ab), p(wd), p(w2), p(nbr), p(d2), p(pre2), p(dm), 16, p(csr_t.rowptr), p(csr_t.perm), N, Hp, p(dab), p(dwd), p(dw2),
        p(dpre2), None, 0, p(ws4), wsb4, st)), flops=N * 16 * Hp * (3 * 2 * 16 + 40) * 1.0)
    # weight gradient of a C x C Linear: split-K kernel against the library GEMM
    dY = torch.randn(N, C, device=dev, generator=g)
    gacc = torch.zeros(C, C, device=dev)
    add("hg_wgrad_f32 [C x N].[N x C] (+ slab reduction)", timed(lambda: ops.wgrad(dY, X, into=gacc)), flops=2.0 * N * C * C)
    add("  library: gacc.addmm_(dY.t(), X)", timed(lambda: gacc.addmm_(dY.t(), X)), flops=2.0 * N * C * C)
    # the 21 deferred weight gradients of one backward pass (7 weights x 3 layer applications) in one launch
    dYs = [torch.randn(N, C, device=dev, generator=g) for _ in range(21)]
    Xs = [torch.randn(N, C, device=dev, generator=g) for _ in range(21)]
    dsts = [torch.zeros(C, C, device=dev) for _ in range(7)]

    def batch21():
        ops.defer_begin(dev)
        ops.wgrad_batch([(dYs[i], Xs[i], 1.0, dsts[i % 7]) for i in range(21)])
        ops.defer_flush(dev)

    add("hg_wgrad_batch_f32, 21 x [C x N].[N x C] (+ reduction)", timed(batch21, reps=10), flops=21 * 2.0 * N * C * C)
    print(f"shapes: N={N} M={M} nnz={nnz} B={ix.B} C={C} Hp={Hp}")
    for r in rows:
        bw = f"{r['GBps']:8.1f} GB/s ({100 * r['frac_hbm']:4.1f}% HBM)" if "GBps" in r else " " * 26
        fl = f"{r['TFLOPs']:6.2f} TF/s ({100 * r['frac_mfma']:4.1f}% fp32 MFMA)" if "TFLOPs" in r else ""
        print(f"{r['kernel']:42s} {r['us']:9.2f} us  {bw}  {fl}")
    if a.json:
        json.dump({"shapes": dict(N=N, M=M, nnz=nnz, B=ix.B, C=C, Hp=Hp), "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
