#!/usr/bin/env python3
"""Diagnostic: where a workgroup of k_gemm_x6 spends its cycles.  Builds csrc/gemm_x6.hip with -DGX_STAMPS into a
scratch library (the product library carries no stamps), runs one problem and prints, per role, the median cycle
counts of the pipeline phases (s_memtime) over the workgroups launched in the middle of the grid."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from equihgnn_amd import hip


def main():
    M, N, K, tile = [int(x) for x in sys.argv[1:5]]
    ta = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    tb = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libgemm_stamps.so")
    extra = [f"-D{d}" for d in os.environ.get("GX_DEFS", "").split() if d]
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DGX_STAMPS", *extra,
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "equihgnn_amd", "csrc"),
                           os.path.join(ROOT, "equihgnn_amd", "csrc", "gemm_x6.hip"), os.path.join(ROOT, "equihgnn_amd", "csrc", "api.hip"), "-o", so])
    L = ctypes.CDLL(so)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tb else (K, N), device=dev, generator=g)
    C = torch.empty(M, N, device=dev)
    tm = 128 if tile == 128 else 64
    n_blocks = ((M + tm - 1) // tm) * ((N + 63) // 64)
    buf = torch.zeros(n_blocks * 16 * 64, dtype=torch.int64, device=dev)
    pr = (hip.HgGemmProblem * 1)()
    q = pr[0]
    q.a, q.lda, q.b, q.ldb, q.c, q.ldc = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0)
    q.m, q.n, q.k, q.trans_a, q.trans_b, q.alpha, q.beta = M, N, K, ta, tb, 1.0, 0.0
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.hg_gemm_x6_batch.argtypes = [ctypes.c_int32, ctypes.POINTER(hip.HgGemmProblem), ctypes.c_int32, ctypes.c_void_p,
                                   ctypes.c_size_t, ctypes.c_void_p]
    for _ in range(int(os.environ.get('GX_WARM', '2000'))):      # ~0.5 s of back-to-back launches: the clock the chip HOLDS under this load
        assert L.hg_gemm_x6_batch(1, pr, tile, None, 0, stream) == 0
    assert L.hg_gemm_x6_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        assert L.hg_gemm_x6_batch(1, pr, tile, None, 0, stream) == 0
    e1.record()
    torch.cuda.synchronize()
    wall_us = e0.elapsed_time(e1) * 1e3 / 20
    ref = (A.t() if ta else A) @ (B.t() if tb else B)
    print("max |err|", float((C - ref).abs().max()))
    st = buf.cpu().numpy().reshape(n_blocks, 16, 64).astype(np.int64)
    nm = 8 if tile in (512, 513) else 4
    tn = {64: 64, 128: 64, 256: 128, 512: 256, 513: 128}[tile]
    tmm = {64: 64, 128: 128, 256: 128, 512: 128, 513: 256}[tile]
    real = ((M + tmm - 1) // tmm) * ((N + tn - 1) // tn)
    if tile in (256, 512, 513) and real > 256:
        real = 256          # one workgroup per CU walks the tiles: the stamps are those of a workgroup's LAST tile
    st = st[:real, :nm + 8]
    n_blocks = real
    t0 = st[:, :, 0].min()
    print(f"blocks {n_blocks}; kernel span {int(st[:, :, 31].max() - t0)} cycles; {wall_us:.1f} us per launch -> {float(st[:, :, 31].max() - t0) / wall_us:.0f} MHz if the span is the launch; "
          f"{2.0 * M * N * K / wall_us / 1e6:.1f} TFLOP/s")
    steps = min((K + 31) // 32, 13)                 # (the stamp record holds 32 slots per wavefront)
    lo, hi = n_blocks // 3, 2 * n_blocks // 3 + 1
    for role, waves in (("multiplier", range(0, nm)), ("stager g0", range(nm, nm + 4)), ("stager g1", range(nm + 4, nm + 8))):
        s = st[lo:hi][:, list(waves), :]
        med = lambda x: float(np.median(x))
        print(f"--- {role} (median over blocks {lo}..{hi})")
        print(f"  start -> first barrier reached : {med(s[:, :, 1] - s[:, :, 0]):9.0f}")
        print(f"  first barrier wait             : {med(s[:, :, 2] - s[:, :, 1]):9.0f}")
        work, wait = [], []
        for k in range(steps):
            prev = s[:, :, 2 + 2 * k]
            work.append(med(s[:, :, 3 + 2 * k] - prev))
            wait.append(med(s[:, :, 4 + 2 * k] - s[:, :, 3 + 2 * k]))
        print("  per step work :", " ".join(f"{w:6.0f}" for w in work))
        print("  per step wait :", " ".join(f"{w:6.0f}" for w in wait))
        if (K + 31) // 32 <= 13:
            print(f"  last barrier -> end            : {med(s[:, :, 31] - s[:, :, 2 + 2 * steps]):9.0f}")
        print(f"  lifetime                       : {med(s[:, :, 31] - s[:, :, 0]):9.0f}")
    # launch cadence: start times of consecutive blocks on the same slot are unknown; print the distribution of starts
    dt, dr = (st[:, 0, 36] - st[:, 0, 34]).astype(np.float64), (st[:, 0, 37] - st[:, 0, 35]).astype(np.float64)
    ok = dr > 0
    if ok.any():
        print(f"shader clock over a workgroup's life (s_memtime / s_memrealtime x 100 MHz), median over workgroups: "
              f"{float(np.median(dt[ok] / dr[ok]) * 100):.0f} MHz; life {float(np.median(dr[ok]) / 100):.1f} us, {float(np.median(dt[ok])):.0f} cycles")
    print("SIMD of wavefronts 0 .. (first blocks):", [[int(x) - 1 for x in st[b, :, 33]] for b in range(0, min(n_blocks, 3))])
    starts = np.sort(st[:, 0, 0] - t0)
    print("block start percentiles (cycles):", [int(np.percentile(starts, p)) for p in (0, 10, 25, 50, 75, 90, 100)])


if __name__ == "__main__":
    main()
