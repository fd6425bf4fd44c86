#!/usr/bin/env python3
"""Phases of one steady-state panel of k_panel_stream (the 4th panel of every workgroup), per role, from the -DPN_STAMPS build
(equihgnn_amd/libequihgnn_panel_stamps.so): s_memtime cycles, median over workgroups.  python tools/stream_stamps.py [rows K N]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from equihgnn_amd import build as _build, hip, ops

rows, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (246016, 256, 256)
L = ctypes.CDLL(_build.STAMPS_LIB)
L.hg_panel_stream_gemm_f32.argtypes = hip.SIGNATURES["hg_panel_stream_gemm_f32"][1]
dev = torch.device("cuda:0")
x = torch.randn(rows, K, device=dev)
w = torch.randn(N, K, device=dev) * K ** -0.5
bias = torch.randn(N, device=dev)
out = torch.empty(rows, N, device=dev)
(img,) = ops.panel_pack([(w, True)])
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
buf = torch.zeros(256 * 16 * 16, dtype=torch.int64, device=dev)
run = lambda: L.hg_panel_stream_gemm_f32(ops._ptr(x), K, rows, K, N, ops._ptr(img), 1.0, None, 0, 1.0, ops._ptr(bias), 1, ops._ptr(out), N, stream)
flags = int(os.environ.get("PS_FLAGS", "0"))      # 1: no result stores, 2: every panel re-reads the first panel's rows
assert L.hg_panel_debug_flags(flags) == 0
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    run()
ev1.record()
ev1.synchronize()
print(f"flags {flags}: {ev0.elapsed_time(ev1) / 5 * 1e3:.1f} us per launch")
assert L.hg_panel_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
buf.zero_()
assert run() == 0
torch.cuda.synchronize()
assert L.hg_panel_debug_stamps(ctypes.c_void_p(0)) == 0
t = buf.cpu().numpy().reshape(256, 16, 16).astype(np.int64)
t = t[t[:, 0, 0] > 0]
print(f"[{rows} x {K}] . [{K} x {N}]: {t.shape[0]} workgroups; 4th panel; cycles (median over workgroups and the role's wavefronts)")
for name, sl, labels in (("multiplying wavefronts", slice(0, 8), ["MFMA loop (incl. prime)", "wait at barrier B", "staging write + barrier A"]),
                         ("row wavefronts", slice(8, 16), ["next image (split, LDS)", "row loads issued + previous epilogue", "wait at barrier B", "wait at barrier A"])):
    tt = t[:, sl, :]
    idx = [0, 2, 3, 4] if name.startswith("mult") else [0, 1, 2, 3, 4]
    d = np.diff(tt[:, :, idx], axis=2)
    med = np.median(d.reshape(-1, d.shape[2]), axis=0)
    print(f"  {name}: " + " | ".join(f"{l} {v:.0f}" for l, v in zip(labels, med)) + f" | whole panel {np.median(tt[:, :, 4] - tt[:, :, 0]):.0f}")
