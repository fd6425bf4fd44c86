#!/usr/bin/env python3
"""Where the read-out head's launch spends its time: s_memtime stamps of the phases of k_readout_mse (wavefront 0 of every
workgroup) in a -DRH_STAMPS build of csrc/readout.hip, on a BASELINE-shaped batch (256 molecules, C = 256, H = 128).

    python tools/readout_stamps.py            (needs hipcc and an MI355X)
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from equihgnn_amd import hip, ops

NAMES = ["issue loads", "pool + vectors to LDS", "barrier", "layer 1 product", "LayerNorm 1", "layer 2 product", "LayerNorm 2",
         "output + dy", "LN2 backward + v3 slab", "dh1 product, v2 / w2 slabs", "barrier", "LN1 backward", "dx product",
         "dX rows", "v1 / w1 slabs"]


def main():
    so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libreadout_stamps.so")
    csrc = os.path.join(ROOT, "equihgnn_amd", "csrc")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DRH_STAMPS", "-ffp-contract=off",
                           "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(csrc, "readout.hip"), os.path.join(csrc, "api.hip"),
                           "-o", so])
    L = ctypes.CDLL(so)
    L.hg_readout_mse_f32.argtypes = hip.SIGNATURES["hg_readout_mse_f32"][1]
    L.hg_readout_mse_workspace_bytes.restype = ctypes.c_size_t
    L.hg_readout_mse_workspace_bytes.argtypes = hip.SIGNATURES["hg_readout_mse_workspace_bytes"][1]
    dev = torch.device("cuda:0")
    B, C, H = 257, 256, 128
    g = torch.Generator().manual_seed(0)
    sizes = torch.randint(9, 28, (B,), generator=g)
    rowptr = torch.cat((torch.zeros(1, dtype=torch.int64), sizes.cumsum(0))).int().to(dev)
    N = int(rowptr[-1])
    x = torch.randn(N, C, generator=g).to(dev)
    y = torch.randn(B, generator=g).to(dev)
    shapes = [(H, C), (H,), (H,), (H,), (H, H), (H,), (H,), (H,), (1, H), (1,)]
    ws_t = [torch.randn(*s, generator=g).to(dev) * 0.1 for s in shapes]
    grads = [torch.zeros_like(w) for w in ws_t]
    vp = ctypes.c_void_p * 10
    pred, loss, dx = torch.empty(B, device=dev), torch.empty((), device=dev), torch.empty_like(x)
    wsb = L.hg_readout_mse_workspace_bytes(B, C, H)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    state = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_wg = 80
    buf = torch.zeros(n_wg * 32, dtype=torch.int64, device=dev)

    def run():
        rc = L.hg_readout_mse_f32(ops._ptr(x), ops._ptr(rowptr), B, B - 1, C, H, vp(*[w.data_ptr() for w in ws_t]), 1e-5, ops._ptr(y),
                                  ops._ptr(pred), ops._ptr(loss), ops._ptr(dx), vp(*[g_.data_ptr() for g_ in grads]), 1, ops._ptr(ws), wsb,
                                  ops._ptr(state), stream)
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert L.hg_readout_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    spans = []
    for _ in range(5):
        buf.zero_()
        run()
        torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(n_wg, 32)[:, :15].astype(np.int64)
        t = t[t[:, 0] > 0]
        spans.append(t)
    assert L.hg_readout_debug_stamps(ctypes.c_void_p(0)) == 0
    t = spans[-1]
    print(f"{t.shape[0]} workgroups; shader cycles (median over workgroups), 16 molecules per workgroup")
    d = np.median(np.diff(t, axis=1), axis=0)
    for i, v in enumerate(d):
        print(f"  {NAMES[i + 1] if i + 1 < len(NAMES) else i:32s} {v:8.0f}")
    print(f"  {'whole kernel (stamp 0 -> 14)':32s} {np.median(t[:, 14] - t[:, 0]):8.0f}   first start -> last end {t[:, 14].max() - t[:, 0].min()}")


if __name__ == "__main__":
    main()
