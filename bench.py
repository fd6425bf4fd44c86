#!/usr/bin/env python3
"""bench.py — training molecules/s of the egnn_equihnns hot path on synthetic QM9-like batches.

    python bench.py --gpus N --steps K --warmup W           (N=1 default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + MSE + backward + gradient all-reduce (RCCL) + Adam on ONE batch per rank
(BASELINE.json configs[1]: QM9-like, --method egnn_equihnns, batch 256 per rank, hidden 256 —
scripts/run_qm9_3d.sh hyper-parameters).  Batches are pre-collated and resident in HBM before
the timed region; the per-batch index build (CSR sort, kNN) is INSIDE the step.  Prints ONE
JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Library-GEMM algorithm selection: let PyTorch's TunableOp pick, per GEMM shape, the fastest
# rocBLAS / hipBLASLt solution during the (untimed) warm-up; +13 % step throughput on MI355X for the
# fp32 [~4.8k x 256] x [256 x 256] shapes of this model (profiles/README.md).  Opt out: EQH_NO_TUNABLEOP=1.
def _argv_value(flag, default):
    return sys.argv[sys.argv.index(flag) + 1] if flag in sys.argv[:-1] else default


# Methods that run eagerly on unpadded batches see new GEMM shapes at every step: tuning each of them would
# cost seconds per step, so they would only use the committed selections.  None is left: the BatchNorm models
# and FAFormer take their batch / cloud statistics over the real atoms of a padded batch.
_EAGER_METHODS = ()
_static_shapes = _argv_value("--method", "egnn_equihnns") not in _EAGER_METHODS and "--no-graph" not in sys.argv
if not os.environ.get("EQH_NO_TUNABLEOP"):
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1" if _static_shapes else "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "50")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME",
                          os.path.join(os.environ.get("TMPDIR", "/tmp"), "eqh_tunableop_%d.csv"))
    # seed every rank's results file with the selections committed for the BASELINE shapes, so only
    # shapes that are new (another batch size / method) are tuned during warm-up
    _seed = os.path.join(os.path.dirname(os.path.abspath(__file__)), "equihgnn_amd", "tuned",
                         "tunableop_gfx950.csv")
    _dst = os.environ["PYTORCH_TUNABLEOP_FILENAME"].replace("%d", os.environ.get("LOCAL_RANK", "0"))
    if os.path.exists(_seed) and not os.path.exists(_dst):
        try:
            import shutil
            shutil.copyfile(_seed, _dst)
        except OSError:
            pass

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--method", default="egnn_equihnns")
    p.add_argument("--batch", type=int, default=256, help="molecules per rank per step")
    p.add_argument("--flavour", default="qm9")
    p.add_argument("--pool", type=int, default=8, help="distinct pre-collated batches per rank")
    p.add_argument("--c4-steps", type=int, default=10,
                   help="timed steps of the strong-scaling BASELINE config 4 leg (PCQM-like, global batch 1024); 0 = skip")
    p.add_argument("--blocks", type=int, default=15,
                   help="timed blocks of --steps steps each (every block bracketed by barrier + synchronize); the line "
                        "reports the MEDIAN block with min / max / spread, so that 1-2 %% steps are resolvable")
    p.add_argument("--no-collective-probe", action="store_true",
                   help="skip `collective_probe` (N=1 only: the multi-rank step driven through a one-rank RCCL group)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-pipeline", action="store_true", help="skip the loader-fed run (`pipeline` in the JSON line)")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--cpu-batch", type=int, default=0,
                   help="molecules in the CPU-baseline sample batch (0: the GPU batch, or a bounded sample for the methods "
                        "whose oracle does not fit a CPU at the full batch)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip `other_configs` (N=1, default workload only: BASELINE configs 0, 2 and 4 -- mhnnm, equiformer_equihnns, "
                        "faformer_equihnns -- timed for a few graph-replayed steps each after the headline)")
    p.add_argument("--other-steps", type=int, default=10, help="timed steps per block of each `other_configs` entry")
    p.add_argument("--timeline-replays", type=int, default=30,
                   help="replays of the time-stamped graph that `roofline` averages over")
    p.add_argument("--wgrad-side-stream", action="store_true",
                   help="issue weight-gradient GEMMs on a second stream (parallel graph branch)")
    p.add_argument("--no-graph", action="store_true",
                   help="eager launches instead of hipGraph replay of the step")
    p.add_argument("--only-roofline", action="store_true",
                   help="only the scatter-kernel replay of `roofline` (the command the rocprofv3 --pmc passes profile)")
    p.add_argument("--only-saturation", action="store_true",
                   help="run just the cache-exceeding scatter probe (used for the rocprofv3 --pmc passes)")
    return p.parse_args()


class ClockProbe:
    """Shader clock the chip holds, read right after a timed block (outside the timed region): eqh_clock_probe, a one-
    wavefront kernel that brackets a 20 us wait with s_memtime / s_memrealtime (MI355X_MICROARCH.md, DVFS give-back 6)."""

    def __init__(self, dev):
        from equihgnn_amd import hip
        self.lib = hip.lib()
        self.dev = dev
        self.out = torch.zeros(2, dtype=torch.int64, device=dev)
        self.khz = int(self.lib.eqh_wall_clock_khz())

    def mhz(self):
        from equihgnn_amd import hip, ops
        hip.check(self.lib.eqh_clock_probe(ops._ptr(self.out), 20, ops._stream(self.dev)), "eqh_clock_probe")
        cyc, ticks = (int(v) for v in self.out.tolist())
        return round(cyc / max(ticks, 1) * self.khz / 1e3, 0)


def block_stats(ms):
    """min / median / max of the timed blocks (ms per step) and their spread (max - min) / median."""
    v = sorted(ms)
    med = v[len(v) // 2]
    return {"min": v[0], "median": med, "max": v[-1], "spread_pct": round((v[-1] - v[0]) / med * 100, 2), "blocks": len(v)}


def seg_reduce_bytes(nnz, n_out, C, has_idx, has_ptr, has_w, n_src):
    """Algorithmic bytes of one hg_segment_reduce_f32 launch:
    4C*nnz gathered rows + 4*nnz index + 4*(R+1) rowptr + 4C*R output (+ the mean-weight
    rowptr reads of the backward form)."""
    b = 4 * C * nnz + 4 * C * n_out
    if has_idx:
        b += 4 * nnz
    if has_ptr:
        b += 4 * (n_out + 1)
    if has_w:
        b += 8 * nnz
    return b


SCATTER_KERNELS = ("k_segment_reduce", "k_segment_reduce<weighted>", "k_gather_ln_fwd", "k_gather_ln_bwd", "k_inc_fwd",
                   "k_inc_fwd_col", "k_inc_bwd_both")
# FAFormer's HBM-bound frame kernels (fa_former_layer.py:61-120,241-289) join the scatter kernels in its line
FRAME_KERNELS = ("k_frame_hidden_fwd", "k_frame_hidden_bwd", "k_drop_mean_fwd", "k_drop_mean_bwd")
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16; the x6 GEMM spends six bf16 MFMAs per fp32 product


def measure_in_graph(method, batch_size, flavour, dev, replays=30, seed=2000):
    """In-graph durations of the hand-written aggregation kernels of ONE real training step, measured live: the
    step is captured into a hipGraph exactly as in the timed run, with device-side time stamps (eqh_stamp, one-thread
    kernels storing the 100 MHz wall clock) around every launch of k_segment_reduce / k_inc_fwd / k_inc_bwd_both and
    the EGNN edge kernels; the graph is replayed `replays` times and the stamps are read after each replay.  HIP
    events cannot do this -- launches inside a replayed graph are invisible to the launching stream.  A bracket
    contains the kernel plus one launch slot; four back-to-back stamp pairs at the head of the graph measure that slot
    and it is subtracted.  The rocprofv3 --kernel-trace --stats summary of this same command is in profiles/."""
    from equihgnn_amd import ops
    from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.registry import default_args
    from equihgnn_amd.trainer import GraphedTrainStep

    ns = default_args(method=method, batch_size=batch_size)
    torch.manual_seed(0)
    model = MODELS[method](1, ns).to(dev)
    host = synth_batch(batch_size, seed, flavour)
    b = pad_batch(host, *bucket_sizes(host.num_nodes, host.num_hyperedges, host.nnz)).packed().to(dev)
    b.num_real_graphs = batch_size
    tr = GraphedTrainStep(model, lr=ns.lr, weight_decay=ns.wd, collective=False)   # rank-local: the other ranks wait
    # The measurement graph builds its index at the head of the step (as rounds 1-5 did), NOT on the side stream: the stamps then
    # bracket each kernel running ALONE, which is what a roofline fraction is about.  In the timed run the next batch's index build
    # overlaps whatever follows the read-out head (k_conv_b3, the first k_inc_bwd_both ...) and stretches those launches while
    # shortening the step (index_build in the JSON line).
    tr.index_prefetch = False
    tr.step(b)                                   # eager bootstrap
    tl = ops.Timeline(dev)
    ops.TIMELINE = tl
    try:
        tr.step(b)                               # capture (stamps included) + first replay
    finally:
        ops.TIMELINE = None
    acc = {}
    for _ in range(replays):
        tr.step(b)
        torch.cuda.synchronize(dev)
        for i, (name, work, us) in enumerate(tl.read_us()):
            acc.setdefault(i, [name, work, 0.0])[2] += us / replays
    tr.close()
    rows = [acc[i] for i in sorted(acc)]
    slot = [r[2] for r in rows if r[0] == "stamp_pair"]
    floor = sum(slot) / max(len(slot), 1)
    per = {}
    for name, work, us in rows:
        if name == "stamp_pair":
            continue
        d = per.setdefault(name, {"launches_per_step": 0, "work": 0, "us": 0.0})
        d["launches_per_step"] += 1
        d["work"] += work
        d["us"] += max(us - floor, 0.05)
    return per, floor


def measure_pipeline(method, batch_size, flavour, dev, rank, epochs=3, batches_per_epoch=100):
    """The same training step fed by the data pipeline instead of a resident pool: a MolStore of synthetic molecules,
    `fit.BucketedLoader` (array-operation collate into pinned packed staging buffers on a prefetch thread, one
    static bucket per epoch, one asynchronous host-to-device copy per batch) and GraphedTrainStep.  The first epoch
    captures; the following ones are timed.  Also the loader's own single-thread collate rate."""
    import numpy as np

    from equihgnn_amd.batch import MolStore, synth_molecule
    from equihgnn_amd.fit import BucketedLoader
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.registry import default_args
    from equihgnn_amd.trainer import GraphedTrainStep, with_next

    rng = np.random.default_rng(4242 + rank)
    store = MolStore([synth_molecule(rng, flavour) for _ in range(batch_size * batches_per_epoch)])
    ns = default_args(method=method, batch_size=batch_size)
    torch.manual_seed(0)
    model = MODELS[method](1, ns).to(dev)
    tr = GraphedTrainStep(model, lr=ns.lr, weight_decay=ns.wd)
    loader = BucketedLoader(store, batch_size, True, seed=1, device=dev)
    # untimed: eager bootstrap, then epochs until every bucket of the loader's ladder has been captured (a capture costs
    # ~0.3 s; a training run pays it once per bucket in its first epoch or two)
    seen = -1
    for _ in range(12):
        for b, nxt in with_next(loader):
            tr.step(b, nxt)
        if len(tr.slots) == seen and not tr.calibrating:     # (and the trainer has chosen its form of the index build)
            break
        seen = len(tr.slots)
    torch.cuda.synchronize(dev)
    c0, s0 = loader.collated, loader.collate_seconds
    t0 = time.perf_counter()
    n = 0
    for _ in range(epochs - 1):
        for b, nxt in with_next(loader):        # one batch of look-ahead: the next batch's index is built beside this step
            tr.step(b, nxt)
            n += b.num_real_graphs
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    n_graphs = len(tr.slots)
    # the same captured step on batches that are already resident (the loader's own device buffers, same padded bucket):
    # what the loader-fed rate is to be compared with -- the bucket of a whole epoch is a few per cent larger than the
    # bucket of the headline's pool of 8 batches, which is padding, not loader overhead
    resident = []
    for b in loader:
        resident.append(b)
        if len(resident) == 4:
            break
    ring = lambda i: resident[(i + 1) % len(resident)]
    for i, b in enumerate(resident):
        tr.step(b, ring(i))
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        for i, b in enumerate(resident):
            tr.step(b, ring(i))
    torch.cuda.synchronize(dev)
    res_ms = (time.perf_counter() - t1) / (reps * len(resident)) * 1e3
    # the evaluation pass of every epoch (main.py:65-87): forward-only hipGraph replay (trainer.GraphedEvalStep) against eager
    from equihgnn_amd.trainer import GraphedEvalStep
    model.eval()
    ev = GraphedEvalStep(model)
    for b in resident:
        ev(b)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    for _ in range(reps):
        for b in resident:
            ev(b)
    torch.cuda.synchronize(dev)
    eval_ms = (time.perf_counter() - t2) / (reps * len(resident)) * 1e3
    with torch.no_grad():
        for b in resident:
            b._hyper_index = None
            model(b)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        for b in resident:
            b._hyper_index = None
            model(b)
        torch.cuda.synchronize(dev)
        eager_eval_ms = (time.perf_counter() - t3) / len(resident) * 1e3
    model.train()
    loader.close()
    tr.close()
    return {"value": round(n / el, 1), "unit": "molecules/s", "ms_per_step": round(el / (n / batch_size) * 1e3, 3),
            "resident_same_bucket_ms_per_step": round(res_ms, 3),
            "eval_ms_per_batch": round(eval_ms, 3), "eval_molecules_per_s": round(batch_size / eval_ms * 1e3, 1),
            "eval_eager_ms_per_batch": round(eager_eval_ms, 3),
            "eval_what": "forward-only hipGraph replay of the evaluation pass (trainer.GraphedEvalStep) on resident padded "
                         "batches; eager = the same forward pass launched from Python",
            "fraction_of_resident_same_bucket": round(res_ms / (el / (n / batch_size) * 1e3), 4),
            "bucket": list(resident[0].x.shape[:1]) + [int(resident[0].edge_attr.shape[0]), int(resident[0].edge_index0.shape[0])],
            "what": "training steps fed by MolStore -> BucketedLoader (prefetch thread, pinned packed staging, one H2D "
                    "copy per batch) -> GraphedTrainStep; PCIe transfer and host collate included",
            "graphs_captured": n_graphs, "graphs_captured_while_timed": n_graphs - seen,
            "host_collate_molecules_per_s": round((loader.collated - c0) / max(loader.collate_seconds - s0, 1e-9), 1),
            "host_collate_threads": 1}


def measure_operator_boundary(batch, dev, C=256, reps=30):
    """The operator-level drop-in of INTEGRATION.md section 3 on its own: ``ops.scatter(src, index, dim=-2, dim_size, reduce)``
    -- torch_scatter.scatter's signature at the reference's call sites (conv.py:91-93,97,173,177) -- on the incidence list of one
    batch of this workload: every call builds its CSR from the unsorted int64 index (the models build it once per batch and
    never take this route), then runs the segmented reduction.  Eager launches, microseconds per call."""
    from equihgnn_amd import ops
    v, e = batch.edge_index0, batch.edge_index1
    keep = v >= 0                                            # (padded batches carry null incidences)
    v, e = v[keep].contiguous(), e[keep].contiguous()
    n_nodes, n_he = int(batch.x.shape[0]), int(batch.edge_attr.shape[0])
    xs = torch.randn(n_nodes, C, device=dev)
    out = {}
    for label, src_rows, index, size in (("node_to_hyperedge_mean", xs[v], e, n_he), ("hyperedge_to_node_sum", torch.randn(n_he, C, device=dev)[e], v, n_nodes)):
        red = "mean" if "mean" in label else "sum"
        for _ in range(3):
            ops.scatter(src_rows, index, dim=-2, dim_size=size, reduce=red)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.scatter(src_rows, index, dim=-2, dim_size=size, reduce=red)
        torch.cuda.synchronize(dev)
        us = (time.perf_counter() - t0) / reps * 1e6
        nnz = int(index.numel())
        alg = 4 * C * nnz + 8 * nnz + 4 * C * size
        out[label] = {"us_per_call": round(us, 1), "incidences": nnz, "rows_out": size, "alg_bytes": alg,
                      "achieved_gbs": round(alg / us / 1e3, 1)}
    out["what"] = ("ops.scatter(src, index, dim=-2, dim_size, reduce): torch_scatter.scatter's signature, CSR built per call (4 eager "
                   "launches) + one segmented reduction; the model classes build the CSR once per batch instead")
    return out


def scatter_roofline(per, floor, names=SCATTER_KERNELS):
    """`roofline` of the bench line: the HBM-bound node<->hyperedge aggregation kernels as they run inside the
    replayed training step: algorithmic bytes (formulas in DESIGN.md §4) / in-graph time."""
    kernels = {}
    tot_b = tot_us = 0.0
    n = 0
    for name in names:
        d = per.get(name)
        if not d:
            continue
        gbs = d["work"] / d["us"] / 1e3
        kernels[name] = {"launches_per_step": d["launches_per_step"],
                         "avg_launch_us": round(d["us"] / d["launches_per_step"], 2),
                         "alg_bytes_per_launch": int(d["work"] / d["launches_per_step"]),
                         "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        tot_b += d["work"]
        tot_us += d["us"]
        n += d["launches_per_step"]
    achieved = tot_b / tot_us / 1e3 if tot_us > 0 else 0.0
    return {"bound": "hbm", "kernel": "HBM-bound kernels inside the replayed step: " + " + ".join(kernels),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "launches_per_step": n, "avg_launch_us": round(tot_us / max(n, 1), 2),
            "alg_bytes_per_launch": int(tot_b / max(n, 1)),
            "method": "device time stamps around each launch inside the hipGraph, averaged over replays; "
                      f"launch slot {floor:.2f} us (stamp pair) subtracted",
            "kernels": kernels}


def measure_scatter_roofline(model, batch, dev):
    """The scatter kernel alone, warm and back to back (NOT the in-step figure: see measure_in_graph): every
    hg_segment_reduce_f32 launch of one training step is recorded, then replayed 20x behind a busy prefix between
    HIP events on the launching stream.  Kept as `roofline.back_to_back` for comparison with round 1."""
    from equihgnn_amd import hip, ops

    calls = []
    real = ops._segment_reduce

    def spy(src, idx, rowptr, wptr, n_out, mean):
        out = real(src, idx, rowptr, wptr, n_out, mean)
        nnz = int(idx.numel()) if idx is not None else (int(n_out) if rowptr is None else int(src.shape[0]))
        calls.append((src.detach(), idx, rowptr, wptr, int(n_out), bool(mean), nnz))
        return out

    ops._segment_reduce = spy
    try:
        batch._hyper_index = None
        out = model(batch)
        torch.nn.functional.mse_loss(out, batch.y).backward()
    finally:
        ops._segment_reduce = real
    torch.cuda.synchronize(dev)

    L = hip.lib()
    stream = torch.cuda.current_stream(dev)
    busy = torch.randn(4096, 4096, device=dev)
    reps = 20
    tot_bytes = tot_ms = 0.0
    for (src, idx, rowptr, wptr, n_out, mean, nnz) in calls:
        src = src.contiguous()
        C = src.shape[-1]
        o = torch.empty((n_out, C), dtype=torch.float32, device=dev)
        args = (ops._ptr(src), ops._ptr(idx), ops._ptr(rowptr), ops._ptr(wptr), ops._ptr(o), n_out, C,
                1 if mean else 0, ops._stream(dev))
        L.hg_segment_reduce_f32(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(busy, busy)  # ~1 ms of queued work: the launches below are enqueued behind it
        e0.record(stream)
        for _ in range(reps):
            L.hg_segment_reduce_f32(*args)
        e1.record(stream)
        e1.synchronize()
        tot_ms += e0.elapsed_time(e1) / reps
        tot_bytes += seg_reduce_bytes(nnz, n_out, C, idx is not None, rowptr is not None,
                                      wptr is not None, src.shape[0])
    n = max(len(calls), 1)
    avg_ms = tot_ms / n
    achieved = (tot_bytes / n) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"kernel": "k_segment_reduce (hg_segment_reduce_f32), eager launches of one step replayed back to back",
            "achieved": round(achieved, 1), "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "launches": len(calls), "avg_launch_us": round(avg_ms * 1e3, 2),
            "alg_bytes_per_launch": int(tot_bytes / n)}


def pmc_traffic(method, batch, flavour, kernel_names):
    """HBM bytes per launch of the scatter kernels from the PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), as
    collected by separate rocprofv3 --pmc passes over `bench.py --only-roofline` and committed under profiles/ (a bench
    run cannot collect counters itself).  Read from the newest profiles/r*_pmc_scatter_workload.json whose workload and
    kernel set match THIS run -- never a constant in this file, which would go stale with the kernels."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_scatter_workload.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        wl = d.get("workload", {"method": "egnn_equihnns", "batch": 256, "flavour": "qm9"})   # (the r01 / r02 files)
        if (wl.get("method"), wl.get("batch"), wl.get("flavour")) != (method, batch, flavour):
            continue
        if set(d.get("kernels", {})) != set(kernel_names):
            continue
        tot = d.get("all_scatter_kernels", {}).get("hbm_bytes_per_launch")
        if tot is not None:
            return int(tot), os.path.relpath(path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2 x FETCH + WRITE, offline)"
    return None, None


def panel_prologues(method, batch, flavour, dev=None):
    """The gather prologues of the panel kernels (k_conv_f2: gathered mean; k_conv_f3: per-incidence hidden layer + mean;
    k_conv_b1: weighted gather of the backward) are aggregation work that has no launch of its own, so device stamps around
    launches cannot see it.  MEASURED IN THIS RUN: tools/panel_prologues.measure brackets each prologue with s_memtime stamps
    inside the -DPN_STAMPS build of the panel kernels (equihgnn_amd/libequihgnn_panel_stamps.so, built by
    __graft_entry__.build() beside the product library) on the index of a batch of this workload.  Only when that library is
    absent: the newest committed profiles/r*_panel_prologue.json of the same workload, labelled as such."""
    import glob
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import panel_prologues as pp_tool
        d = pp_tool.measure(method, batch, flavour, dev=dev)
        d["source"] = "measured in this run (tools/panel_prologues.measure: in-kernel s_memtime stamps, -DPN_STAMPS build loaded beside the product library)"
        return d
    except Exception as exc:  # noqa: BLE001 -- the stamped build is a measurement aid: its absence must not cost the headline
        why = f"{type(exc).__name__}: {exc}"[:200]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_panel_prologue.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        wl = d.get("workload", {})
        if (wl.get("method"), wl.get("batch"), wl.get("flavour")) == (method, batch, flavour):
            d["source"] = os.path.relpath(path, ROOT) + f" (offline file: the live measurement failed -- {why})"
            return d
    return None


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
INFINITY_CACHE_GATHER_GBS = 8600.0   # MI355X_MICROARCH.md "Indexed rows: gather": 38 MB table, uniformly random rows, chip-wide


def saturation_probe(dev, log2_nodes=20, C=256, reps=10, seed=0):
    """The scatter kernel alone at a size far beyond the 256 MiB Infinity Cache:
    N = 2^20 nodes, nnz = 2.3 N incidences over M = 1.1 N hyperedges, C = 256 (1 GiB of node
    features).  Forward node->hyperedge gathered mean and its backward (hyperedge->node, weighted),
    each timed with HIP events over `reps` back-to-back launches."""
    from equihgnn_amd import hip, ops

    N = 1 << log2_nodes
    M = int(1.1 * N)
    nnz = int(2.3 * N)
    g = torch.Generator(device=dev).manual_seed(seed)
    v = torch.randint(0, N, (nnz,), device=dev, generator=g)
    e = torch.randint(0, M, (nnz,), device=dev, generator=g)
    by_e = ops.csr_build(e, v, M)
    by_v = ops.csr_build(v, e, N)
    X = torch.randn(N, C, device=dev, generator=g)
    dE = torch.randn(M, C, device=dev, generator=g)
    L = hip.lib()
    stream = torch.cuda.current_stream(dev)
    res = {"nodes": N, "hyperedges": M, "incidences": nnz, "C": C}
    for name, (src, idx, ptr, wptr, rows) in {
            "fwd_v2e_mean": (X, by_e.col, by_e.rowptr, None, M),
            "bwd_e2v_weighted": (dE, by_v.col, by_v.rowptr, by_e.rowptr, N)}.items():
        out = torch.empty((rows, C), dtype=torch.float32, device=dev)
        args = (ops._ptr(src), ops._ptr(idx), ops._ptr(ptr), ops._ptr(wptr), ops._ptr(out), rows, C,
                1 if wptr is None else 0, ops._stream(dev))
        for _ in range(2):
            L.hg_segment_reduce_f32(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            L.hg_segment_reduce_f32(*args)
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        b = seg_reduce_bytes(nnz, rows, C, True, True, wptr is not None, src.shape[0])
        res[name] = {"us": round(ms * 1e3, 1), "alg_bytes": b, "GBps": round(b / (ms * 1e-3) / 1e9, 1),
                     "frac_of_8TBps": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return res


def fused_saturation(dev, log2_nodes=20, C=256, reps=3, seed=1):
    """The FUSED aggregation kernels of the step (k_gather_ln_fwd / _bwd, k_inc_fwd_col, k_inc_bwd_both) at a size far
    beyond the Infinity Cache: N = 2^20 nodes, M = 1.1 N hyperedges, nnz = 2.3 N incidences, C = 256.  Each launch is
    bracketed by device time stamps (ops.Timeline, as inside the replayed step); a kernel takes milliseconds here, so
    eager launches measure the device."""
    from equihgnn_amd import ops

    N = 1 << log2_nodes
    M, nnz = int(1.1 * N), int(2.3 * N)
    g = torch.Generator(device=dev).manual_seed(seed)
    v = torch.randint(0, N, (nnz,), device=dev, generator=g)
    e = torch.randint(0, M, (nnz,), device=dev, generator=g)
    by_e, by_v = ops.csr_build(e, v, M), ops.csr_build(v, e, N)
    v32, e32 = v.to(torch.int32), e.to(torch.int32)
    mk = lambda *shape: torch.randn(*shape, device=dev, generator=g).requires_grad_()
    h, pa, qb = mk(N, C), mk(N, C), mk(M, C)
    bias, g1, b1, g2, b2 = mk(C), mk(C), mk(C), mk(C), mk(C)
    tl = ops.Timeline(dev, capacity=256)
    acc = {}
    for rep in range(reps + 1):
        tl.reset()
        for _ in range(2):
            tl.pair("stamp_pair")
        ops.TIMELINE = tl
        try:
            hbar = ops.gather_ln_reduce(h, bias, g1, b1, by_e, by_v, "mean", 1e-5)
            s_ = ops.incidence_ln_reduce(pa, qb, g2, b2, v32, e32, by_v, by_e, by_v, v32, "mean", 1e-5)
            (hbar.sum() + s_.sum()).backward()
        finally:
            ops.TIMELINE = None
        rows = tl.read_us()
        for t in (h, pa, qb, bias, g1, b1, g2, b2):
            t.grad = None
        if rep == 0:
            continue                      # warm-up (first-touch of the gradient buffers)
        floor = sum(us for n_, _, us in rows if n_ == "stamp_pair") / 2
        for n_, work, us in rows:
            if n_ != "stamp_pair":
                d = acc.setdefault(n_, [work, 0.0])
                d[1] += max(us - floor, 0.05) / reps
    out = {"nodes": N, "hyperedges": M, "incidences": nnz, "C": C}
    for n_, (work, us) in acc.items():
        gbs = work / us / 1e3
        out[n_] = {"us": round(us, 1), "alg_bytes": int(work), "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4)}
    return out


# The TRUE reference's own CPU path (the reference's model files imported in the build container, 8 host
# cores, same hyper-parameters, full training step).  It cannot travel to the GPU box, so the figures are quoted, labelled,
# beside the timed restatement ("port"); egnn_equihnns at batch 256 allocates a zero-filled [1, N, N, C] buffer in
# batched_index_select's backward (egnn_layer.py:18-32), which the restatement does not reproduce.
REFERENCE_TRUE_CPU = {
    ("egnn_equihnns", 256, "qm9"): (9.9, "25.9 s / step; 22 GB transient allocation (egnn_layer.py:18-32)"),
    ("mhnnm", 256, "qm9"): (590.0, "0.40-0.47 s / step"),
    ("mhnnm", 32, "qm9"): (279.0, "0.115 s / step"),
    ("equiformer_equihnns", 32, "qm9"): (1.0, "hidden 256, batch 32: 30.7 s / step (batch 128 does not fit: 262 KB of radial weights per edge)"),
    ("faformer_equihnns", 64, "pcqm"): (11.0, "hidden 256, batch 64 (QM9-like molecules): 5.85 s / step"),
}


def reference_true(method, batch, flavour):
    key = (method, batch, flavour)
    if key not in REFERENCE_TRUE_CPU:   # the nearest measured batch of the same method
        cands = [k for k in REFERENCE_TRUE_CPU if k[0] == method]
        if not cands:
            return None
        key = min(cands, key=lambda k: abs(k[1] - batch))
    v, note = REFERENCE_TRUE_CPU[key]
    return {"value": v, "unit": "molecules/s", "cores": 8, "kind": "reference", "batch": key[1],
            "where": "build container (not the GPU box), BASELINE.md 2: the reference's own model files on CPU", "note": note}


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (the
    GPU box exposes 256 logical CPUs but grants a 16-core share per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    if os.environ.get("EQH_CPU_CORES"):
        n = int(os.environ["EQH_CPU_CORES"])
    return max(1, min(n, 16 if n > 64 else n))


def cpu_baseline(method, args_ns, batch_cpu, seconds):
    """The oracle (kind "port": this repo's CPU restatement, pinned to the reference by the
    golden vectors) timed on the host cores: same batch shape, same full training step."""
    from oracle import ref_models as O

    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = O.MODELS[method](1, args_ns)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(model(batch_cpu), batch_cpu.y)
        loss.backward()
        opt.step()

    tw = time.perf_counter()
    step()  # warm-up
    tw = time.perf_counter() - tw
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 50 or (tw > seconds and n >= 1):
            break
    mol_s = n * batch_cpu.y.shape[0] / el
    return {"value": round(mol_s, 2), "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"{n} full training steps (fwd+MSE+bwd+Adam) of the CPU oracle on one "
                      f"B={batch_cpu.y.shape[0]} batch of the same workload, after 1 warm-up; "
                      f"{el:.1f} s"}


class _StdoutToStderr:
    """RCCL announces itself on STDOUT when a communicator is created ("RCCL version : ...", five lines); this script's
    stdout is ONE JSON line.  File-descriptor level: the banner comes from C code."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) through
    torch.distributed.run, as the reference's Trainer(devices="auto", strategy="ddp...") does (main.py:271-283).
    Runs BEFORE anything in this process touches the GPU (importing torch and counting devices does not), and as a
    child process -- a process that has initialised HIP must never be replaced by exec."""
    import socket
    import subprocess

    n_dev = torch.cuda.device_count()
    shared = os.environ.get("EQH_BACKEND", "nccl") != "nccl"      # gloo rehearsal: ranks may share one device
    if n_dev < a.gpus and not shared:
        print(f"bench.py: --gpus {a.gpus} but only {n_dev} device(s) are visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" is RCCL on ROCm.  EQH_BACKEND=gloo exists only to rehearse the multi-rank code path
        # on a one-GPU box (several ranks sharing one device cannot form an RCCL communicator).
        backend = os.environ.get("EQH_BACKEND", "nccl")
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            t0 = torch.zeros(1, device=dev)
            dist.all_reduce(t0)                     # (the communicator exists now, whatever the backend's laziness)
            torch.cuda.synchronize(dev)

    if a.only_saturation:
        sat = saturation_probe(dev)
        sat["fused_kernels"] = fused_saturation(dev)
        print(json.dumps({"saturation": sat}), flush=True)
        return

    from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.registry import default_args
    from equihgnn_amd.trainer import GraphedTrainStep, TrainStep

    if a.wgrad_side_stream:
        from equihgnn_amd import ops as _ops
        _ops.WGRAD_ON_SIDE_STREAM = True
    args_ns = default_args(method=a.method, batch_size=a.batch)
    if a.only_roofline:
        per, floor = measure_in_graph(a.method, a.batch, a.flavour, dev, a.timeline_replays)
        print(json.dumps({"roofline": scatter_roofline(per, floor), "all_kernels_us": {k: round(v["us"], 2) for k, v in per.items()}}),
              flush=True)
        return
    use_graph = (not a.no_graph) and a.method not in _EAGER_METHODS

    probe = ClockProbe(dev)

    def timed_run(method, batch, flavour, steps, warmup, cfg_id, trainer_kw=None, blocks=None):
        """K timed steps of one workload on this rank (barrier + synchronize on both sides, MAX over ranks)."""
        ns = default_args(method=method, batch_size=batch)
        torch.manual_seed(0)
        model = MODELS[method](1, ns).to(dev)
        host = [synth_batch(batch, cfg_id * 1000 + rank * 100 + i, flavour) for i in range(a.pool)]
        # hipGraph replay needs static shapes: the collate stage pads every batch to the bucket of the
        # largest one (one dummy molecule owns the padding; exact for the LayerNorm models, and for the
        # BatchNorm ones because their statistics count the real atoms only).
        if use_graph:
            ext = [bucket_sizes(b.num_nodes, b.num_hyperedges, b.nnz) for b in host]
            tgt = tuple(max(e[i] for e in ext) for i in range(3))
            if world > 1:   # every rank pads to the same bucket (one capture shape per job, like one DDP bucket plan)
                tt = torch.tensor(tgt, dtype=torch.int64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tgt = tuple(int(v) for v in tt.tolist())
            batches = [pad_batch(b, *tgt).packed().to(dev) for b in host]   # one staging buffer per batch
            for b in batches:
                b.num_real_graphs = batch
            trainer = GraphedTrainStep(model, lr=ns.lr, weight_decay=ns.wd, **(trainer_kw or {}))
            # set-up, not warm-up: the eager bootstrap step (lays out the flat buffers, tunes unseen GEMM shapes)
            # and the capture step, so that the W warm-up and K timed steps below are all graph replays whatever W is
            for i in range(2):
                trainer.step(batches[i % a.pool])
            # every step also gets the batch of the NEXT step: its index (CSR sorts, kNN) is built on a side stream while
            # this step runs -- one index build per step inside the timed region, as before, but off the critical path
            step = lambda j: trainer.step(batches[j % a.pool], batches[(j + 1) % a.pool])
        else:
            batches = [b.to(dev) for b in host]
            trainer = TrainStep(model, lr=ns.lr, weight_decay=ns.wd)

            def fresh(b):  # every step sees a "new" batch: the index (CSR sort, kNN) is rebuilt
                b._hyper_index = None

            trainer.on_batch = fresh
            step = lambda j: trainer.step(batches[j % a.pool])
        base = 2
        # set-up as well: the trainer's choice between building the next index ahead and building it in the step (it times a
        # window of steps in each form, GraphedTrainStep._calibrate: the same number of steps on every rank)
        while getattr(trainer, "calibrating", False):
            step(base)
            base += 1
        for i in range(warmup):
            step(base + i)
        els, clocks = [], []
        for blk in range(max(1, blocks or a.blocks)):
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(steps):
                loss = step(base + warmup + blk * steps + i)
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)      # the slowest rank's clock, per block
            els.append(float(t.item()))
            clocks.append(probe.mhz())        # after the block's clock was read: not part of any timed region
        return {"el": sorted(els)[len(els) // 2], "els": els, "loss": float(loss), "host": host, "batches": batches,
                "model": model, "trainer": trainer, "args": ns, "clocks": clocks}

    run = timed_run(a.method, a.batch, a.flavour, a.steps, a.warmup, 2)
    tr0 = run["trainer"]
    index_info = None
    if isinstance(tr0, GraphedTrainStep):
        on = any(s_.get("prefetch") is not None for s_ in tr0.slots.values())
        index_info = {"mode": "prefetch" if on else "in_step_graph", "steps_with_index_built_ahead": tr0.prefetch_hits,
                      "steps_that_built_it_first": tr0.prefetch_misses,
                      "index_graph_alone_us": None if tr0.index_build_us is None else round(tr0.index_build_us, 1),
                      "policy": tr0.prefetch_policy, "calibration": tr0.calibration,
                      "what": ("the per-batch index (three CSR sorts, kNN, transposed kNN CSR) of the NEXT step's batch is built by "
                               "its own small hipGraph on a side stream while this step's graph runs; one build per step, all "
                               "inside the timed region (device-wide synchronize at both ends); the step graph starts with "
                               "one batched copy of the finished index" if on else
                               "the per-batch index is built at the head of the step's hipGraph")}
    el, loss, host_batches, batches, model = run["el"], run["loss"], run["host"], run["batches"], run["model"]
    block_ms = [round(e / a.steps * 1e3, 4) for e in run["els"]]
    block_clocks = run["clocks"]
    run_args = run["args"]
    mode = getattr(run["trainer"], "collective_mode", "none")
    mode_err = getattr(run["trainer"], "capture_error", None)
    observed_world = dist.get_world_size() if world > 1 else 1      # what the (RCCL) process group reports
    assert observed_world == a.gpus, (observed_world, a.gpus)

    # The multi-rank step on this one GPU (N=1 only): a ONE-rank RCCL process group drives GraphedTrainStep's collective
    # path -- broadcast, flat-gradient all-reduce, 1 / world scale -- once with the all-reduce captured into the step's graph
    # and once in the split form (graph A -> eager all-reduce -> graph B), same workload and batches as the headline.
    coll = None
    if world == 1 and use_graph and not a.no_collective_probe:
        import socket
        coll = {"what": "the headline step through a 1-rank 'nccl' (RCCL) group with force_collective: all-reduce inside the "
                        "step's one hipGraph (in_graph) and eagerly between two hipGraphs (two_graph)",
                "single_rank_ms_per_step": round(el / a.steps * 1e3, 4)}
        try:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            with _StdoutToStderr():
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
                t0 = torch.zeros(1, device=dev)
                dist.all_reduce(t0)
                torch.cuda.synchronize(dev)
            try:
                ref = timed_run(a.method, a.batch, a.flavour, a.steps, a.warmup, 2, blocks=min(a.blocks, 7),
                                trainer_kw=dict(collective=False))
                coll["single_rank_again_ms_per_step"] = round(ref["el"] / a.steps * 1e3, 4)
                del ref
                for label, kw in (("in_graph", dict(force_collective=True)),
                                  ("two_graph", dict(force_collective=True, graph_collective=False))):
                    r = timed_run(a.method, a.batch, a.flavour, a.steps, a.warmup, 2, trainer_kw=kw, blocks=min(a.blocks, 7))
                    coll[f"{label}_ms_per_step"] = round(r["el"] / a.steps * 1e3, 4)
                    coll[f"{label}_mode"] = r["trainer"].collective_mode
                    if r["trainer"].capture_error:
                        coll[f"{label}_capture_error"] = r["trainer"].capture_error[:300]
                    coll[f"{label}_final_loss"] = round(r["loss"], 6)
                    r["trainer"].close()
                    del r
                coll["two_graph_over_single"] = round(coll["two_graph_ms_per_step"] / coll["single_rank_again_ms_per_step"], 4)
                coll["in_graph_over_single"] = round(coll["in_graph_ms_per_step"] / coll["single_rank_again_ms_per_step"], 4)
            finally:
                dist.destroy_process_group()
        except Exception as exc:     # (an RCCL that cannot initialise on this box must not cost the headline)
            coll["error"] = f"{type(exc).__name__}: {exc}"[:400]

    # BASELINE config 4 as a STRONG-scaling point: PCQM4Mv2-like molecules, global batch 1024 split over the ranks
    # (SURVEY.md §8d), beside the weak-scaling headline above.  Reported inside the same JSON line.
    strong = None
    if a.method == "egnn_equihnns" and a.flavour == "qm9" and a.c4_steps > 0 and 1024 % world == 0:
        del run
        r4 = timed_run("egnn_equihnns", 1024 // world, "pcqm", a.c4_steps, 2, 4, blocks=min(a.blocks, 5))
        strong = {"workload": "PCQM4Mv2-like synthetic molecules, --method egnn_equihnns, GLOBAL batch 1024 "
                              f"({1024 // world}/rank), hidden 256", "scaling": "strong", "n_gpus": observed_world,
                  "steps": a.c4_steps, "value": round(1024 * a.c4_steps / r4["el"], 1), "unit": "molecules/s",
                  "ms_per_step": round(r4["el"] / a.c4_steps * 1e3, 3), "final_loss": round(r4["loss"], 6),
                  "timed_blocks": block_stats([round(e / a.c4_steps * 1e3, 4) for e in r4["els"]])}
        del r4

    # BASELINE.json's other configurations on this one GPU (scripts/run_qm9_3d.sh:10-31 hyper-parameters, hidden 256): a few
    # graph-replayed training steps each, same harness as the headline (no CPU leg, no roofline): driver-visible ms per step.
    others = None
    if (world == 1 and use_graph and not a.no_other_configs and a.method == "egnn_equihnns" and a.flavour == "qm9"
            and a.batch == 256):
        others = {}
        for label, (m_, b_, fl_, cfg_, what_) in {
                "c0_mhnnm_b32": ("mhnnm", 32, "qm9", 10, "BASELINE configs[0]: QM9-like, --method mhnnm, batch 32 (the reference's CPU-runnable case, here on the GPU)"),
                "c0_mhnnm_b256": ("mhnnm", 256, "qm9", 11, "configs[0]'s method at the headline's batch 256"),
                "c2_equiformer_equihnns_b128": ("equiformer_equihnns", 128, "qm9", 3, "BASELINE configs[2]: QM9-like, --method equiformer_equihnns, batch 128"),
                "c4_faformer_equihnns_b512": ("faformer_equihnns", 512, "pcqm", 5, "BASELINE configs[4]: Molecule3D / PCQM-like molecules, --method faformer_equihnns, batch 512 per GPU")}.items():
            try:
                r_ = timed_run(m_, b_, fl_, a.other_steps, 3, cfg_, blocks=3)
                ms_ = r_["el"] / a.other_steps * 1e3
                others[label] = {"workload": what_, "ms_per_step": round(ms_, 3), "value": round(b_ / ms_ * 1e3, 1), "unit": "molecules/s",
                                 "steps": a.other_steps, "blocks_ms_per_step": [round(e / a.other_steps * 1e3, 3) for e in r_["els"]],
                                 "final_loss": round(r_["loss"], 6), "launch": "hipGraph replay (padded static shapes)",
                                 "index_build": getattr(r_["trainer"], "calibration", None)}
                r_["trainer"].close()
                del r_
            except Exception as exc:  # noqa: BLE001 -- a side measurement must not cost the headline line
                others[label] = {"workload": what_, "error": f"{type(exc).__name__}: {exc}"[:300]}
            torch.cuda.empty_cache()

    result = None
    if rank == 0:
        n_nodes = sum(b.num_nodes for b in host_batches) / a.pool
        nnz = sum(b.nnz for b in host_batches) / a.pool
        result = {
            "metric": "training molecules/sec on QM9 egnn_equihnns" if a.method == "egnn_equihnns"
            else f"training molecules/sec ({a.method})",
            "value": round(world * a.batch * a.steps / el, 1),
            "unit": "molecules/s",
            "n_gpus": observed_world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 3),
            "timed_blocks_ms_per_step": block_ms,        # every block = --steps steps between barrier + synchronize; value = median
            "timed_blocks": block_stats(block_ms),
            "shader_clock_mhz": {"after_each_block": block_clocks, "min": min(block_clocks), "max": max(block_clocks),
                                 "how": "eqh_clock_probe right after each timed block: d s_memtime / d s_memrealtime x 100 MHz"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.flavour}-like synthetic molecules, --method {a.method}, "
                                   f"batch {a.batch}/rank, hidden 256, 3 layers (scripts/run_qm9_3d.sh); "
                                   "full training step fwd+MSE+bwd+all-reduce+Adam",
                       "batch_per_rank": a.batch, "global_batch": a.batch * world,
                       "avg_nodes": round(n_nodes, 1), "avg_incidences": round(nnz, 1),
                       "parallelism": f"dp{world}",
                       "launch": "hipGraph replay (padded static shapes)" if use_graph else "eager",
                       "collective": (f"one flat-gradient all-reduce per step, backend {dist.get_backend()}, "
                                      + {"in_graph": "captured as a node of the step's one hipGraph",
                                         "split": "eager between two hipGraphs" + (f" (in-graph capture failed: {mode_err})" if mode_err else "")}.get(mode, mode)
                                      if world > 1 else "none (single rank)")},
            "final_loss": round(float(loss), 6),
        }
        if index_info is not None:
            result["index_build"] = index_info
        if others is not None:
            result["other_configs"] = others
        if strong is not None:
            result["strong_scaling_c4"] = strong
        if coll is not None:
            result["collective_probe"] = coll
        if not a.no_roofline and use_graph:
            per, floor = measure_in_graph(a.method, a.batch, a.flavour, dev, a.timeline_replays)
            names = SCATTER_KERNELS + (FRAME_KERNELS if a.method == "faformer_equihnns" else ())
            result["roofline"] = scatter_roofline(per, floor, names)
            traffic, src = pmc_traffic(a.method, a.batch, a.flavour, result["roofline"]["kernels"].keys())
            result["roofline"]["traffic"] = traffic
            if traffic is not None:
                result["roofline"]["traffic_source"] = src
            pp = panel_prologues(a.method, a.batch, a.flavour, dev) if "k_conv_f2" in per else None
            if pp is not None:
                # the aggregation work inside the panel launches, next to the stamped launches: bytes / time over both
                rk = result["roofline"]
                rk["kernels"]["panel_prologue"] = {"kernels": pp["kernels"], "all": pp["all"], "source": pp["source"],
                                                   "wavefronts_per_panel": pp.get("wavefronts_per_panel")}
                b_in = sum(v["alg_bytes_per_launch"] * v["launches_per_step"] for k_, v in rk["kernels"].items() if k_ != "panel_prologue")
                us_in = sum(v["avg_launch_us"] * v["launches_per_step"] for k_, v in rk["kernels"].items() if k_ != "panel_prologue")
                b_pp = sum(v["alg_bytes_per_launch"] * v["launches_per_step"] for v in pp["kernels"].values())
                us_pp = sum(v["prologue_us"] * v["launches_per_step"] for v in pp["kernels"].values())
                if us_in + us_pp > 0:
                    gbs = (b_in + b_pp) / (us_in + us_pp) / 1e3
                    rk["including_panel_prologues"] = {
                        "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "us_per_step": round(us_in + us_pp, 1), "alg_bytes_per_step": int(b_in + b_pp),
                        "what": "every node <-> hyperedge aggregation of the step: the stamped launches above (live) + the gather "
                                "prologues of k_conv_f2 / k_conv_f3 / k_conv_b1 (in-kernel stamps of the -DPN_STAMPS build, measured in this run; see panel_prologue.source)"}
            if "k_gemm_x6" in per:      # the dense products that run on the bf16 matrix cores (csrc/gemm_x6.hip)
                gx = per["k_gemm_x6"]
                tf = gx["work"] / gx["us"] / 1e6
                result["roofline"]["gemm_x6"] = {
                    "bound": "mfma", "launches_per_step": gx["launches_per_step"], "us": round(gx["us"], 1),
                    "fp32_flops": int(gx["work"]), "achieved": round(tf, 1), "unit": "TFLOP/s (2 M N K per second)",
                    "peak": round(BF16_MFMA_PEAK_TFLOPS / 6, 1), "frac": round(tf / (BF16_MFMA_PEAK_TFLOPS / 6), 4),
                    "note": "six bf16 MFMAs per fp32 product: peak = dense bf16 peak / 6; the fp32-input MFMA peak is "
                            f"{FP32_MFMA_PEAK_TFLOPS} TFLOP/s"}
            pk = {}
            for name in ("k_conv_f1", "k_conv_f2", "k_conv_f3", "k_conv_b3", "k_conv_b2", "k_conv_b1", "k_node_f", "k_node_b"):
                if name in per:     # row-panel kernels (csrc/panel.hip): conv-sized products + the row work between them, x6 arithmetic
                    tf = per[name]["work"] / per[name]["us"] / 1e6
                    pk[name] = {"launches_per_step": per[name]["launches_per_step"], "us": round(per[name]["us"], 1),
                                "fp32_flops": int(per[name]["work"]), "achieved": round(tf, 1),
                                "frac": round(tf / (BF16_MFMA_PEAK_TFLOPS / 6), 4)}
            if pk:
                tot_f = sum(v["fp32_flops"] for v in pk.values())
                tot_us = sum(v["us"] for v in pk.values())
                result["roofline"]["panel_kernels"] = {
                    "bound": "mfma", "unit": "TFLOP/s (2 M N K per second)", "peak": round(BF16_MFMA_PEAK_TFLOPS / 6, 1),
                    "launches_per_step": sum(v["launches_per_step"] for v in pk.values()), "us": round(tot_us, 1),
                    "achieved": round(tot_f / tot_us / 1e6, 1), "frac": round(tot_f / tot_us / 1e6 / (BF16_MFMA_PEAK_TFLOPS / 6), 4),
                    "note": "32-row panels, 8 wavefronts each: ~150 workgroups on 256 CUs at the BASELINE batch, each streaming the "
                            "whole [C x C] weight image (384 KB) through its CU's vector memory path per product -- that and the 192 "
                            "MFMAs per SIMD (2.6 us) bound a product; the row phases between products (split into bf16 planes, "
                            "LayerNorm, gathers) are bound by the same 148 CUs' VALU and vector-memory rate (DESIGN.md section 4)",
                    "kernels": pk}
            mf = {}
            for name in ("k_rowgemm_fwd", "k_rowgemm_bwd"):     # Equiformer's radial tensor product
                if name in per:
                    # A node's [Kd, L] matrix is used once per 16 entries: the launches are STREAMS over the node matrices (a few
                    # fp32 MFMAs ride along: 2 E Kd L flops = 8 flops per matrix byte at 16 entries per node), so they are booked
                    # against HBM: matrix bytes = flops / 8 (the forward and dz passes read them, the dw pass writes them).
                    tf = per[name]["work"] / per[name]["us"] / 1e6
                    gbs = per[name]["work"] / 8.0 / per[name]["us"] / 1e3
                    mf[name] = {"launches_per_step": per[name]["launches_per_step"], "us": round(per[name]["us"], 1),
                                "matrix_bytes": int(per[name]["work"] / 8.0), "achieved": round(gbs, 1), "unit": "GB/s",
                                "peak": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4),
                                "mfma_flops": int(per[name]["work"]), "mfma_tflops": round(tf, 1)}
            if mf:
                mf["bound"] = "hbm"
                result["roofline"]["rowgemm_kernels"] = mf
            edge = {}
            for name in ("egnn_edge_fwd", "egnn_edge_bwd"):
                if name in per:
                    # What bounds them (round 5, profiles/r05_ab_runs.txt: with the SiLU / split / MFMA work compiled out the
                    # forward keeps 46 of its 58 us): the GATHER of the neighbours' rows -- 16 rows of 4 Hp bytes per node, no
                    # locality (the kNN graph ignores molecule boundaries), served by the Infinity Cache, whose measured ceiling
                    # for uniformly random rows is 8.6 TB/s (MI355X_MICROARCH.md, "Indexed rows: gather").  Bytes per launch:
                    # forward N 16 Hp 4 (the senders' B rows) = flops / 8; backward twice that (B rows by receiver, A rows by sender).
                    tf = per[name]["work"] / per[name]["us"] / 1e6
                    gbytes = per[name]["work"] / (8.0 if name == "egnn_edge_fwd" else 12.0)
                    gbs = gbytes / per[name]["us"] / 1e3
                    edge[name] = {"us": round(per[name]["us"], 1), "gather_bytes": int(gbytes), "achieved": round(gbs, 1), "unit": "GB/s",
                                  "peak": INFINITY_CACHE_GATHER_GBS, "frac": round(gbs / INFINITY_CACHE_GATHER_GBS, 4),
                                  "mfma_flops": int(per[name]["work"]), "mfma_tflops": round(tf, 1),
                                  "where": "inside the replayed step"}
            if edge:
                edge["bound"] = "gather"
                edge["note"] = ("random-row gather from the Infinity Cache (peak: the microarchitecture guide's measured 8.6 TB/s for a 38 MB "
                                "table); the forward multiplies on the bf16 pipe (3 planes, 6 MFMAs), the backward on the fp32 MFMA; the "
                                "forward's VALU floor (2 transcendentals + the 3-plane split per hidden unit and edge) is ~37 us at this batch")
                result["roofline"]["edge_kernels"] = edge
            if world == 1:
                b2b = measure_scatter_roofline(model, host_batches[0].to(dev), dev)
                if b2b["launches"] > 0:      # (methods on the merged conv path launch no plain k_segment_reduce: nothing to report)
                    result["roofline"]["back_to_back"] = b2b
                result["roofline"]["operator_boundary"] = measure_operator_boundary(batches[0], dev)
                result["roofline"]["saturation"] = saturation_probe(dev)
                result["roofline"]["saturation"]["fused_kernels"] = fused_saturation(dev)
        if not a.no_pipeline and use_graph and world == 1:
            result["pipeline"] = measure_pipeline(a.method, a.batch, a.flavour, dev, rank)
            result["pipeline"]["fraction_of_resident"] = round(result["pipeline"]["value"] / result["value"], 4)
        if world == 1 and not a.no_cpu_baseline:
            # equiformer_equihnns at hidden 256 materialises 9.7 GB of radial weights per pair type at batch 128 on the
            # CPU (as the reference does); FAFormer at 15 k atoms a dense [N, N] search: bounded samples for those
            cb = a.cpu_batch or {"equiformer_equihnns": 8, "faformer_equihnns": 64}.get(a.method, a.batch)
            sample = host_batches[0] if cb == a.batch else synth_batch(cb, 2000, a.flavour)
            result["cpu_baseline"] = cpu_baseline(a.method, run_args, sample, a.cpu_seconds)
            ref = reference_true(a.method, a.batch, a.flavour)
            if ref is not None:
                result["cpu_baseline"]["reference_true"] = ref
            if strong is not None:      # config 4's own CPU figure: a bounded sample of its PCQM-like molecules
                c4_sample = synth_batch(128, 4000, "pcqm")
                strong["cpu_baseline"] = cpu_baseline("egnn_equihnns", default_args(method="egnn_equihnns", batch_size=128),
                                                      c4_sample, min(a.cpu_seconds, 10.0))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
