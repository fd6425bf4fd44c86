#!/usr/bin/env python3
"""Diagnostic: where a workgroup of k_panel_plain spends its cycles.  Builds csrc/panel.hip with -DPN_STAMPS into a scratch
library (the product library carries no stamps), runs one [rows x 256] . [256 x 256] product and prints the median cycle
counts (s_memtime) of the phases over all workgroups.  python tools/panel_stamps.py [rows]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from equihgnn_amd import hip, ops


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4864
    C = 256
    so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libpanel_stamps.so")
    extra = [f"-D{d}" for d in os.environ.get("PN_DEFS", "").split() if d]
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DPN_STAMPS", *extra,
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "equihgnn_amd", "csrc"),
                           os.path.join(ROOT, "equihgnn_amd", "csrc", "panel.hip"), os.path.join(ROOT, "equihgnn_amd", "csrc", "api.hip"), "-o", so])
    L = ctypes.CDLL(so)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(rows, C, device=dev, generator=g)
    w = torch.randn(C, C, device=dev, generator=g) * C ** -0.5
    (img,) = ops.panel_pack([(w, True)])
    out = torch.empty_like(x)
    nb = (rows + 31) // 32
    buf = torch.zeros(nb * 4 * 8, dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.hg_panel_gemm_f32.argtypes = hip.SIGNATURES["hg_panel_gemm_f32"][1]

    def run():
        assert L.hg_panel_gemm_f32(x.data_ptr(), C, rows, C, img.data_ptr(), 1.0, None, 0, 0.0, None, 0, out.data_ptr(), C, stream) == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert L.hg_panel_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
    print("max |err|", float((out - x @ w.t()).abs().max()))
    st = buf.cpu().numpy().reshape(nb, 4, 8).astype(np.int64)
    t0 = st[:, :, 0].min()
    print(f"workgroups {nb}; kernel span {int(st[:, :, 5].max() - t0)} cycles")
    med = lambda a: float(np.median(a))
    names = ["rows loaded, split, in LDS", "barrier wait", "MFMA loop (16 K steps)", "staging + barrier", "row epilogue + store"]
    for i, n in enumerate(names):
        print(f"  {n:30s}: {med(st[:, :, i + 1] - st[:, :, i]):8.0f}")
    print(f"  {'lifetime':30s}: {med(st[:, :, 5] - st[:, :, 0]):8.0f}")
    starts = np.sort(st[:, 0, 0] - t0)
    print("workgroup start percentiles (cycles):", [int(np.percentile(starts, q)) for q in (0, 10, 50, 90, 100)])
    ends = np.sort(st[:, :, 5].max(axis=1) - t0)
    print("workgroup end percentiles (cycles):", [int(np.percentile(ends, q)) for q in (0, 10, 50, 90, 100)])


if __name__ == "__main__":
    main()
