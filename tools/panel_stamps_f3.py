#!/usr/bin/env python3
"""Diagnostic: phases of k_conv_f3 (no tail) by s_memtime stamps; builds csrc/panel.hip with -DPN_STAMPS into a scratch library."""
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4736
    C = 256
    so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libpanel_stamps.so")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DPN_STAMPS",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "equihgnn_amd", "csrc"),
                           os.path.join(ROOT, "equihgnn_amd", "csrc", "panel.hip"), os.path.join(ROOT, "equihgnn_amd", "csrc", "api.hip"), "-o", so])
    L = ctypes.CDLL(so)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    s_, cw = rn(rows, C), rn(rows, C)
    w23, w3b = rn(C, C) * C ** -0.5, rn(C, C) * C ** -0.5
    b3a, g3, be3, b3b = rn(C), rn(C), rn(C), rn(C)
    i23, i3b = ops.panel_pack([(w23, True), (w3b, True)])
    u, x3, xn = torch.empty_like(s_), torch.empty_like(s_), torch.empty_like(s_)
    nb = (rows + 31) // 32
    buf = torch.zeros(nb * 8 * 16, dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = hip.HgConvPanel()
    a.rows, a.C, a.eps, a.scale, a.relu, a.tail = rows, C, 1e-5, 0.5, 1, 0
    for k, t in dict(in0=s_, in1=cw, w0=i23, b0=b3a, g0=g3, be0=be3, w1=i3b, bias_out=b3b, out0=u, out1=x3, out2=xn).items():
        setattr(a, k, t.data_ptr())
    L.hg_conv_panel.argtypes = hip.SIGNATURES["hg_conv_panel"][1]
    run = lambda: L.hg_conv_panel(hip.HG_CONV_F3, a, stream)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    assert L.hg_panel_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    assert run() == 0
    torch.cuda.synchronize()
    nw = int(L.hg_panel_waves())
    st = buf.cpu().numpy().reshape(nb, 8, 16).astype(np.int64)[:, :nw, :]
    print(f"wavefronts per panel: {nw}")
    med = lambda x: float(np.median(x))
    names = ["rows loaded, split, A image", "barrier", "MFMA 1", "staging + prime + barrier", "row phase (LN, stores, split)", "barrier",
             "MFMA 2 + staging", "barrier + bias / ReLU + store"]
    for i, n in enumerate(names):
        print(f"  {n:34s}: {med(st[:, :, i + 1] - st[:, :, i]):8.0f}")
    print(f"  {'stamp 0 -> 8':34s}: {med(st[:, :, 8] - st[:, :, 0]):8.0f}")


if __name__ == "__main__":
    main()
