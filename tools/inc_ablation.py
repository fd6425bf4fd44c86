#!/usr/bin/env python3
"""What bounds k_inc_fwd_col at the BASELINE batch (4.9 k rows, ~10 k incidences, C = 256)?  The same launch with parts of
its body compiled out, each as a chain of dependent launches inside a replayed hipGraph (the conditions of the training step):
  full       the product kernel
  no-LN      rows gathered and summed, LayerNorm arithmetic removed            (-DINC_ABLATE=1)
  chain      index chain (rowptr -> col) + the row's own operand + the store   (-DINC_ABLATE=2)
The algorithmic bytes of the launch over the `chain` time is the ceiling ANY kernel with this index chain and output has at
this size; over the `no-LN` time, the ceiling of a pure gather-reduce.  python tools/inc_ablation.py"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from equihgnn_amd import hip, ops
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
from equihgnn_amd.index import HyperIndex

dev = torch.device("cuda:0")
C = 256
host = synth_batch(256, 2000, "qm9")
b = pad_batch(host, *bucket_sizes(host.num_nodes, host.num_hyperedges, host.nnz)).to(dev)
ix = HyperIndex.from_batch(b)
N, M, nnz = ix.N, ix.M, ix.by_v.nnz
g = torch.Generator(device=dev).manual_seed(0)
pa, qb = torch.randn(N, C, device=dev, generator=g), torch.randn(M, C, device=dev, generator=g)
gam, bet = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
outs = [torch.empty(N, C, device=dev) for _ in range(2)]
work = 4 * C * (2 * nnz + N) + 12 * nnz + 4 * (N + 1) + 8 * C
print(f"rows {N}, hyperedges {M}, incidences {nnz}, algorithmic bytes per launch {work}")
NCH = 40


def build(flag):
    so = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"libinc_ablate{flag}.so")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", f"-DINC_ABLATE={flag}",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "equihgnn_amd", "csrc"),
                           os.path.join(ROOT, "equihgnn_amd", "csrc", "incidence.hip"), os.path.join(ROOT, "equihgnn_amd", "csrc", "api.hip"),
                           "-o", so])
    L = ctypes.CDLL(so)
    L.hg_incidence_ln_reduce_fwd_col.argtypes = hip.SIGNATURES["hg_incidence_ln_reduce_fwd_col"][1]
    return L


def timeit(name, L):
    def chain():
        cur = pa
        for i in range(NCH):            # each launch reads the previous launch's output as its row operand: a dependent chain
            o = outs[i & 1]
            assert L.hg_incidence_ln_reduce_fwd_col(cur.data_ptr(), qb.data_ptr(), ix.by_v.rowptr.data_ptr(), ix.by_v.col.data_ptr(), 1,
                                                    gam.data_ptr(), bet.data_ptr(), N, C, 1, 1e-5, o.data_ptr(),
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            cur = o
    for _ in range(3):
        chain()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            gr.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (20 * NCH) * 1e6)
    print(f"{name:8s} {best:6.2f} us / launch in a replayed graph   {work / best / 1e3:7.1f} GB/s algorithmic = {work / best / 1e3 / 8000:.3f} of 8 TB/s")


timeit("full", build(0))
timeit("no-LN", build(1))
timeit("chain", build(2))
