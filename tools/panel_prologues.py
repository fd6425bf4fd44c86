#!/usr/bin/env python3
"""HBM-side figure of the panel kernels' gather PROLOGUES (the aggregation work that used to be k_gather_ln_fwd / _bwd and
k_inc_fwd_col launches and now runs inside k_conv_f2 / k_conv_b1 / k_conv_f3): per-wavefront s_memtime stamps around each
prologue (a -DPN_STAMPS build of csrc/panel.hip into a scratch library), on the index of the BASELINE batch.  A launch is one
round of workgroups (~150 on 256 CUs), so a prologue's duration is the median over its wavefronts; its algorithmic bytes are
the launch's (all panels).  Writes the JSON that bench.py reports as roofline.kernels.panel_prologue:

    python tools/panel_prologues.py profiles/r05_panel_prologue.json
"""
import ctypes
import json
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

HBM_PEAK_GBS = 8000.0


def measure(method="egnn_equihnns", batch=256, flavour="qm9", C=256, dev=None, verbose=False):
    """The three prologues on the index of one synthetic batch of the workload, through the stamped build of the panel kernels
    (equihgnn_amd/libequihgnn_panel_stamps.so, built by equihgnn_amd.build next to the product library; compiled here into
    TMPDIR when it is missing and hipcc is at hand).  Returns the dict bench.py reports."""
    from equihgnn_amd import build as _build
    so = _build.STAMPS_LIB
    if not os.path.exists(so):
        so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libpanel_stamps.so")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DPN_STAMPS",
                               "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "equihgnn_amd", "csrc"),
                               os.path.join(ROOT, "equihgnn_amd", "csrc", "panel.hip"), os.path.join(ROOT, "equihgnn_amd", "csrc", "api.hip"), "-o", so])
    L = ctypes.CDLL(so)
    L.hg_conv_panel.argtypes = hip.SIGNATURES["hg_conv_panel"][1]
    dev = torch.device("cuda:0") if dev is None else dev
    host = synth_batch(batch, 2000, flavour)
    b = pad_batch(host, *bucket_sizes(host.num_nodes, host.num_hyperedges, host.nnz)).to(dev)
    ix = HyperIndex.from_batch(b)
    N, M, nnz = ix.N, ix.M, ix.by_v.nnz
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    W = [rn(C, C) * C ** -0.5 for _ in range(7)]
    i12t, i23t, i3bt = ops.panel_pack([(W[0], True), (W[1], True), (W[2], True)])
    i12, i3b, i23 = ops.panel_pack([(W[0], False), (W[2], False), (W[1], False)])
    (istack,) = ops.panel_pack([[(W[3], False), (W[4], False)]])
    vecs = [rn(C) for _ in range(8)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nw = int(L.hg_panel_waves())
    # the shader clock under this load: s_memtime ticks are shader cycles (MI355X_MICROARCH.md), eqh_clock_probe gives MHz
    probe = torch.zeros(2, dtype=torch.int64, device=dev)

    def mhz():
        hip.check(hip.lib().eqh_clock_probe(ops._ptr(probe), 20, ops._stream(dev)), "eqh_clock_probe")
        cyc, ticks = (int(v) for v in probe.tolist())
        return cyc / max(ticks, 1) * int(hip.lib().eqh_wall_clock_khz()) / 1e3

    def stamped(stage, rows, fields):
        nb = (rows + 31) // 32
        buf = torch.zeros(nb * 8 * 16, dtype=torch.int64, device=dev)
        a = hip.HgConvPanel()
        for k, v in fields.items():
            setattr(a, k, v.data_ptr() if torch.is_tensor(v) else v)
        a.rows, a.C, a.eps, a.eps_inc, a.scale = rows, C, 1e-5, 1e-5, 0.5
        run = lambda: L.hg_conv_panel(stage, a, stream)
        null = ctypes.c_void_p(0)
        assert L.hg_panel_debug_stamps(null) == 0
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        clocks = []
        spans = []
        for _ in range(5):
            buf.zero_()
            assert L.hg_panel_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
            assert run() == 0
            torch.cuda.synchronize()
            clocks.append(mhz())
            spans.append(buf.cpu().numpy().reshape(nb, 8, 16).astype(np.int64)[:, :nw, :])
        assert L.hg_panel_debug_stamps(null) == 0
        return spans, float(np.median(clocks))

    new = lambda r: torch.empty(r, C, device=dev)
    res = {}

    def report(name, spans, clock, s0, s1, alg_bytes, what):
        cyc = float(np.median([np.median(st[:, :, s1] - st[:, :, s0]) for st in spans]))
        us = cyc / clock
        gbs = alg_bytes / us / 1e3
        res[name] = {"prologue_cycles": round(cyc), "shader_clock_mhz": round(clock), "prologue_us": round(us, 2),
                     "alg_bytes_per_launch": int(alg_bytes), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "launches_per_step": 3, "what": what}
        if verbose:
            print(f"{name:12s} {cyc:8.0f} cycles = {us:6.2f} us  {alg_bytes / 1e6:7.2f} MB  {gbs:7.1f} GB/s  {gbs / HBM_PEAK_GBS:.3f} of 8 TB/s")

    # ---- F2: hbar[e] = mean over the hyperedge's nodes of h1n (stamps 0 -> 1: gather, mean, hbar store, A image) -------------
    h1n, hbar, qb = rn(N, C), new(M), new(M)
    spans, clock = stamped(hip.HG_CONV_F2, M, dict(in0=h1n, rowptr=ix.by_e.rowptr, col=ix.by_e.col, w0=i12t, bias_out=vecs[0],
                                                   out0=hbar, out1=qb))
    report("k_conv_f2", spans, clock, 0, 1, ops.segment_reduce_bytes(nnz, M, C, True, True, False),
           "gathered mean over the hyperedge's nodes (conv.py:172-173; k_gather_ln_fwd's reduce): gathered rows + col + rowptr in, hbar out")
    # ---- F3: s[v] = gamma2 mean_e xhat(relu(pa[v] + qb[e])) + beta2 (stamps 0 -> 1: incidence prologue, s store, A image) ----
    pa, cw, s = rn(N, C), rn(N, C), new(N)
    u, x3, xn = new(N), new(N), new(N)
    spans, clock = stamped(hip.HG_CONV_F3, N, dict(in0=pa, in2=qb, rowptr=ix.by_v.rowptr, col=ix.by_v.col, g_inc=vecs[1],
                                                   be_inc=vecs[2], out6=s, in1=cw, w0=i23t, b0=vecs[3], g0=vecs[4], be0=vecs[5],
                                                   w1=i3bt, bias_out=vecs[6], out0=u, out1=x3, out2=xn, relu=1, tail=0))
    from equihgnn_amd.ops.rows import inc_fwd_col_bytes
    report("k_conv_f3", spans, clock, 0, 1, inc_fwd_col_bytes(nnz, N, C) + 4 * C * N,
           "per-incidence hidden layer + hyperedge -> node mean (conv.py:175-177; k_inc_fwd_col's bytes) + the cw rows loaded beside it")
    # ---- B1: gathered weighted sum of dqb over the node's hyperedges (stamps 0 -> 1), h1 / dpa rows loaded beside it ----------
    dqb, h1, dpa, xprev = rn(M, C), rn(N, C), rn(N, C), rn(N, C)
    ew = ops.entry_weights(ix.by_v, ix.by_e)
    outs = [new(N) for _ in range(5)]
    acc = torch.zeros(N, C, device=dev)
    slab1, slab2 = ops.conv_panel_slab(N, C, dev), ops.conv_panel_slab(N, C, dev)
    v1, v3 = torch.zeros(3, C, device=dev), torch.zeros(3, C, device=dev)
    spans, clock = stamped(hip.HG_CONV_B1, N, dict(
        in0=dqb, w3=i12, rowptr=ix.by_v.rowptr, col=ix.by_v.col, wq=ew, in1=h1, b0=vecs[0], g0=vecs[1], in2=dpa, w0=istack,
        out0=outs[0], slab=slab1, dbias=v1[0], dgamma=v1[1], dbeta=v1[2], in3=xprev, w1=i3b, w2=i23, out5=u, b1=vecs[2], g1=vecs[3],
        out2=outs[1], out3=outs[2], out4=outs[3], acc_out=acc, slab2=slab2, dbias2=v3[0], dgamma2=v3[1], dbeta2=v3[2], tail=1,
        acc_first=0))
    report("k_conv_b1", spans, clock, 0, 1, 4 * C * nnz + 8 * nnz + 4 * (N + 1) + 8 * C * N,
           "weighted gather of dqb over the node's hyperedges (backward of the gathered mean; k_gather_ln_bwd's reduce): gathered rows + "
           "col + weight per entry + rowptr, with the h1 and dpa rows of the panel loaded beside it (the sums stay in registers)")
    tot_b = sum(v["alg_bytes_per_launch"] * v["launches_per_step"] for v in res.values())
    tot_us = sum(v["prologue_us"] * v["launches_per_step"] for v in res.values())
    out = {"workload": {"method": method, "batch": batch, "flavour": flavour}, "wavefronts_per_panel": nw, "nodes": N,
           "hyperedges": M, "incidences": nnz,
           "method": "s_memtime stamps around the prologue of every wavefront (PN_STAMPS build), median over wavefronts and 5 launches; "
                     "cycles / shader clock (eqh_clock_probe after each launch); stand-alone launches on the BASELINE batch's index",
           "kernels": res,
           "all": {"us_per_step": round(tot_us, 2), "achieved": round(tot_b / tot_us / 1e3, 1),
                   "frac": round(tot_b / tot_us / 1e3 / HBM_PEAK_GBS, 4)}}
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    out = measure(verbose=True)
    print(json.dumps(out["all"]))
    if out_path:
        with open(out_path, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
