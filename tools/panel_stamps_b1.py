#!/usr/bin/env python3
"""Diagnostic: phases of k_conv_b1 with the B3 tail (the longest panel launch) by s_memtime stamps; builds csrc/panel.hip with
-DPN_STAMPS into a scratch library.  python tools/panel_stamps_b1.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from equihgnn_amd import hip, ops
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
from equihgnn_amd.index import HyperIndex


def main():
    C = 256
    so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libpanel_stamps.so")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DPN_STAMPS",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "equihgnn_amd", "csrc"),
                           os.path.join(ROOT, "equihgnn_amd", "csrc", "panel.hip"), os.path.join(ROOT, "equihgnn_amd", "csrc", "api.hip"), "-o", so])
    L = ctypes.CDLL(so)
    dev = torch.device("cuda:0")
    host = synth_batch(256, 2000, "qm9")
    b = pad_batch(host, *bucket_sizes(host.num_nodes, host.num_hyperedges, host.nnz)).to(dev)
    ix = HyperIndex.from_batch(b)
    N, M = ix.N, ix.M
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    W = [rn(C, C) * C ** -0.5 for _ in range(5)]
    i12, i3b, i23 = ops.panel_pack([(W[0], False), (W[1], False), (W[2], False)])
    (istack,) = ops.panel_pack([[(W[3], False), (W[4], False)]])
    vecs = [rn(C) for _ in range(4)]
    dqb, h1, dpa, xprev, u = rn(M, C), rn(N, C), rn(N, C), rn(N, C), rn(N, C)
    ew = ops.entry_weights(ix.by_v, ix.by_e)
    outs = [torch.empty(N, C, device=dev) for _ in range(6)]
    acc = torch.zeros(N, C, device=dev)
    slab1, slab2 = ops.conv_panel_slab(N, C, dev), ops.conv_panel_slab(N, C, dev)
    v1, v3 = torch.zeros(3, C, device=dev), torch.zeros(3, C, device=dev)
    nb = (N + 31) // 32
    buf = torch.zeros(nb * 8 * 16, dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = hip.HgConvPanel()
    a.rows, a.C, a.eps, a.scale, a.tail, a.acc_first = N, C, 1e-5, 0.5, 1, 0
    for k, t in dict(in0=dqb, w3=i12, rowptr=ix.by_v.rowptr, col=ix.by_v.col, wq=ew, in1=h1, b0=vecs[0], g0=vecs[1], in2=dpa, w0=istack,
                     out0=outs[0], slab=slab1, dbias=v1[0], dgamma=v1[1], dbeta=v1[2], in3=xprev, w1=i3b, w2=i23, out5=u, b1=vecs[2],
                     g1=vecs[3], out2=outs[2], out3=outs[3], out4=outs[4], acc_out=acc, slab2=slab2, dbias2=v3[0], dgamma2=v3[1],
                     dbeta2=v3[2]).items():
        setattr(a, k, t.data_ptr())
    L.hg_conv_panel.argtypes = hip.SIGNATURES["hg_conv_panel"][1]
    run = lambda: L.hg_conv_panel(hip.HG_CONV_B1, a, stream)
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
    names = ["prime w12, load h / dpa, gather dqb sums", "a_put + product w12 + staging", "prime stacked, LN1 bwd, a_put x 2",
             "barrier + stacked product (K = 512) + staging", "dX tile, slab 1", "tail: B3 (2 products, LN3 bwd, slab 2)"]
    for i, n in enumerate(names):
        print(f"  {n:46s}: {med(st[:, :, i + 1] - st[:, :, i]):8.0f}")
    print(f"  {'total':46s}: {med(st[:, :, 6] - st[:, :, 0]):8.0f}")
    tail = [(5, 8, "tail: prime W3b, load u / mask, a_put"), (8, 9, "barrier + product W3b + staging"), (9, 10, "prime w23 + barrier"),
            (10, 11, "LN3 bwd, stores, acc_out, a_put"), (11, 12, "barrier + product w23 + staging"), (12, 13, "barrier + ds store"),
            (13, 6, "slab 2")]
    for a_, b_, n in tail:
        print(f"    {n:44s}: {med(st[:, :, b_] - st[:, :, a_]):8.0f}")


if __name__ == "__main__":
    main()
