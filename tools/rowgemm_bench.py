#!/usr/bin/env python3
"""The Equiformer row products alone (csrc/rowgemm.hip): hg_rowgemm_fwd / _bwd on 2432 nodes x 16 entries for the shapes of
BASELINE config 3, receiver rows (16 entries each) and sender-like rows (random keys: Poisson(16) entries), the node matrices
rotated over three buffers; microseconds and TB/s of matrix bytes.  EQH_ROWGEMM_LDS=1 times round 4's workgroup-per-row kernels.

    python tools/rowgemm_bench.py            (profiles/r05_c3_ab_runs.txt holds a run of each)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from equihgnn_amd import hip, ops
dev = torch.device('cuda:0')
N, K = 2432, 16
E = N * K
g = torch.Generator(device=dev).manual_seed(0)
recv = torch.arange(0, E + 1, K, dtype=torch.int32, device=dev)
key = torch.randint(0, N, (E,), device=dev, generator=g)
csr = ops.csr_build(key, None, N)
L_ = hip.lib(); st = ops._stream(dev)
def t(fn, reps=30):
    for _ in range(3): fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for Kd, L in ((64, 256), (192, 64), (256, 64), (64, 64)):
    ws = [torch.randn(N, Kd, L, device=dev) for _ in range(3)]
    z = torch.randn(E, Kd, device=dev); dout = torch.randn(E, L, device=dev)
    out = torch.zeros(E, L, device=dev); dz = torch.zeros(E, Kd, device=dev)
    for name, rp, pm in (("recv", recv, None), ("send", csr.rowptr, csr.perm)):
        p = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
        f = t(lambda i: L_.hg_rowgemm_fwd(p(z), p(ws[i % 3]), p(rp), p(pm), N, Kd, L, p(out), 1, st))
        bz = t(lambda i: L_.hg_rowgemm_bwd(p(z), p(ws[i % 3]), p(dout), p(rp), p(pm), N, Kd, L, p(dz), 0, None, st))
        bw = t(lambda i: L_.hg_rowgemm_bwd(p(z), None, p(dout), p(rp), p(pm), N, Kd, L, None, 0, p(ws[i % 3]), st))
        mb = N * Kd * L * 4 / 1e6
        print(f"Kd {Kd:3d} L {L:3d} {name}: fwd {f:6.1f} us ({mb/f:5.2f} TB/s)  bwd_z {bz:6.1f} ({mb/bz:5.2f})  bwd_w {bw:6.1f} ({mb/bw:5.2f})   matrix {mb:.0f} MB", flush=True)
