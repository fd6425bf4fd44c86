#!/usr/bin/env python3
"""In-graph cost of one conv-sized product [rows x C] . [C x C] (C = 256): library fp32 GEMM (TunableOp), the x6 kernel and
the panel kernel, each as a chain of N dependent launches captured in a hipGraph and replayed (what a launch costs inside the
replayed training step, not back-to-back eager launches).  python tools/panel_bench.py [rows]"""
import os
import sys
import time

os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "50")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/panel_bench_tunable_%d.csv")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equihgnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4864
C = 256
N = 40
torch.manual_seed(0)
x = torch.randn(rows, C, device=dev)
w = torch.randn(C, C, device=dev) * C ** -0.5
bias = torch.randn(C, device=dev)
(img,) = ops.panel_pack([(w, True)])
bufs = [torch.empty_like(x) for _ in range(2)]


def chain(fn):
    cur = x
    for i in range(N):
        out = bufs[i & 1]
        fn(cur, out)
        cur = out
    return cur


def lib(cur, out):
    torch.mm(cur, w.t(), out=out)


def lib_bias_relu(cur, out):
    torch._addmm_activation(bias, cur, w.t(), out=out)


def x6(cur, out):
    ops.gemm(cur, w, trans_a=False, trans_b=True, out=out)


def panel(cur, out):
    ops.panel_gemm(cur, img, C, out=out)


def panel_bias_relu(cur, out):
    ops.panel_gemm(cur, img, C, bias=bias, relu=True, out=out)


def empty(cur, out):
    ops.copy_many([out[:1]], [cur[:1]])


def timeit(name, fn):
    for _ in range(3):
        chain(fn)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(fn)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (20 * N) * 1e6)
    flops = 2.0 * rows * C * C
    print(f"{name:28s} {best:7.2f} us / launch   {flops / best / 1e6:7.1f} TFLOP/s", flush=True)
    return best


print(f"rows = {rows}, C = {C}, {N} dependent launches per graph")
timeit("tiny kernel (launch slot)", empty)
timeit("library fp32 (TunableOp)", lib)
timeit("library + bias + relu", lib_bias_relu)
timeit("x6", x6)
timeit("panel", panel)
timeit("panel + bias + relu", panel_bias_relu)
