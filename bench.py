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
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--wgrad-side-stream", action="store_true",
                   help="issue weight-gradient GEMMs on a second stream (parallel graph branch)")
    p.add_argument("--no-graph", action="store_true",
                   help="eager launches instead of hipGraph replay of the step")
    p.add_argument("--only-roofline", action="store_true",
                   help="only the scatter-kernel replay of `roofline` (the command the rocprofv3 --pmc passes profile)")
    p.add_argument("--only-saturation", action="store_true",
                   help="run just the cache-exceeding scatter probe (used for the rocprofv3 --pmc passes)")
    return p.parse_args()


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


def measure_scatter_roofline(model, batch, dev):
    """Record every hg_segment_reduce_f32 launch of one training step, then replay each launch
    behind a busy prefix (so the queue is GPU-bound) bracketed by HIP events on the launching
    stream.  Returns the roofline dict for that kernel."""
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
    return {"bound": "hbm", "kernel": "k_segment_reduce (hg_segment_reduce_f32)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "launches_per_step": len(calls), "avg_launch_us": round(avg_ms * 1e3, 2),
            "alg_bytes_per_launch": int(tot_bytes / n)}


# HBM bytes per launch of k_segment_reduce from the PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), collected
# with rocprofv3 on `python3 bench.py --only-roofline` at the BASELINE workload: profiles/r01_pmc_scatter_workload.json.
# A bench run cannot collect counters itself; the figure is attached only to the workload it was measured on.
PMC_TRAFFIC_BYTES_PER_LAUNCH = {("egnn_equihnns", 256, "qm9"): 11706695}


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak


def measure_edge_kernels(batch, dev, hidden=256, reps=20):
    """The other hand-written hot kernels of the step, the fused EGNN edge update (egnn_edge_fwd / _bwd),
    timed live with HIP events on the batch's own neighbour graph.  FLOPs per launch:
    forward N*16*Hp*(2*16 + 12), backward N*16*Hp*(3*2*16 + 40); they are bound by fp32 MFMA + VALU issue
    (the fp32 MFMA shares the VALU pipe, profiles/r01_edge_fwd_notes.md), so the fraction of the MFMA peak
    is an upper-bound style figure, not a bandwidth one."""
    from equihgnn_amd import hip, ops
    from equihgnn_amd.index import HyperIndex

    batch._hyper_index = None
    ix = HyperIndex.from_batch(batch)
    nbr, d2, csr_t = ix.knn(batch.pos, 16, 0)
    N = ix.N
    H = 2 * (2 * hidden + 1)
    Hp = H + (-H) % 64
    g = torch.Generator(device=dev).manual_seed(0)
    ab = torch.randn(N, 2 * Hp, device=dev, generator=g)
    wd = torch.randn(Hp, device=dev, generator=g) * 0.1
    w2 = torch.randn(16, Hp, device=dev, generator=g) / Hp ** 0.5
    b2 = torch.zeros(16, device=dev)
    m, pre2 = torch.empty(N, 16, device=dev), torch.empty(N, 16, 16, device=dev)
    dm = torch.randn(N, 16, device=dev, generator=g)
    dab, dwd, dw2, dpre2 = torch.empty_like(ab), torch.empty_like(wd), torch.empty_like(w2), torch.empty_like(pre2)
    L, p, st = hip.lib(), ops._ptr, ops._stream(dev)
    wsb = L.egnn_edge_bwd_workspace_bytes(N, Hp)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    fwd = lambda: L.egnn_edge_fwd(p(ab), p(wd), p(w2), p(b2), p(nbr), p(d2), N, Hp, p(m), p(pre2), st)
    bwd = lambda: L.egnn_edge_bwd(p(ab), p(wd), p(w2), p(nbr), p(d2), p(pre2), p(dm), 16, p(csr_t.rowptr), p(csr_t.perm),
                                  N, Hp, p(dab), p(dwd), p(dw2), p(dpre2), None, 0, p(ws), wsb, st)
    busy = torch.randn(4096, 4096, device=dev)
    stream = torch.cuda.current_stream(dev)
    out = {}
    for name, fn, flops in (("egnn_edge_fwd", fwd, N * 16 * Hp * (2 * 16 + 12)),
                            ("egnn_edge_bwd", bwd, N * 16 * Hp * (3 * 2 * 16 + 40))):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(busy, busy)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        tf = flops / us / 1e6
        out[name] = {"us": round(us, 1), "flops": int(flops), "achieved": round(tf, 1), "unit": "TFLOP/s",
                     "peak": FP32_MFMA_PEAK_TFLOPS, "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}
    out["bound"] = "mfma"
    out["nodes"], out["Hp"] = N, Hp
    return out


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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" is RCCL on ROCm.  EQH_BACKEND=gloo exists only to rehearse the multi-rank code path
        # on a one-GPU box (several ranks sharing one device cannot form an RCCL communicator).
        backend = os.environ.get("EQH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if a.only_saturation:
        print(json.dumps({"saturation": saturation_probe(dev)}), flush=True)
        return

    from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.registry import default_args
    from equihgnn_amd.trainer import GraphedTrainStep, TrainStep

    if a.wgrad_side_stream:
        from equihgnn_amd import ops as _ops
        _ops.WGRAD_ON_SIDE_STREAM = True
    args_ns = default_args(method=a.method, batch_size=a.batch)
    torch.manual_seed(0)
    model = MODELS[a.method](1, args_ns).to(dev)
    cfg_id = 2
    host_batches = [synth_batch(a.batch, cfg_id * 1000 + rank * 100 + i, a.flavour) for i in range(a.pool)]
    if a.only_roofline:
        print(json.dumps({"roofline": measure_scatter_roofline(model, host_batches[0].to(dev), dev)}), flush=True)
        return
    # hipGraph replay needs static shapes: the collate stage pads every batch to the bucket of the
    # largest one (one dummy molecule owns the padding; exact for the LayerNorm models, and for the
    # BatchNorm ones because their statistics count the real atoms only).
    use_graph = (not a.no_graph) and a.method not in _EAGER_METHODS
    if use_graph:
        ext = [bucket_sizes(b.num_nodes, b.num_hyperedges, b.nnz) for b in host_batches]
        tgt = tuple(max(e[i] for e in ext) for i in range(3))
        batches = [pad_batch(b, *tgt).packed().to(dev) for b in host_batches]   # one staging buffer per batch
        for b in batches:
            b.num_real_graphs = a.batch
        trainer = GraphedTrainStep(model, lr=args_ns.lr, weight_decay=args_ns.wd)
    else:
        batches = [b.to(dev) for b in host_batches]
        trainer = TrainStep(model, lr=args_ns.lr, weight_decay=args_ns.wd)

        def fresh(b):  # every step sees a "new" batch: the index (CSR sort, kNN) is rebuilt
            b._hyper_index = None

        trainer.on_batch = fresh
    if use_graph:
        # set-up, not warm-up: the eager bootstrap step (lays out the flat buffers, tunes unseen GEMM shapes)
        # and the capture step, so that the W warm-up and K timed steps below are all graph replays whatever W is
        for i in range(2):
            trainer.step(batches[i % a.pool])
    for i in range(a.warmup):
        trainer.step(batches[i % a.pool])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = trainer.step(batches[i % a.pool])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())

    result = None
    if rank == 0:
        n_nodes = sum(b.num_nodes for b in host_batches) / a.pool
        nnz = sum(b.nnz for b in host_batches) / a.pool
        result = {
            "metric": "training molecules/sec on QM9 egnn_equihnns" if a.method == "egnn_equihnns"
            else f"training molecules/sec ({a.method})",
            "value": round(world * a.batch * a.steps / el, 1),
            "unit": "molecules/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.flavour}-like synthetic molecules, --method {a.method}, "
                                   f"batch {a.batch}/rank, hidden 256, 3 layers (scripts/run_qm9_3d.sh); "
                                   "full training step fwd+MSE+bwd+all-reduce+Adam",
                       "batch_per_rank": a.batch, "global_batch": a.batch * world,
                       "avg_nodes": round(n_nodes, 1), "avg_incidences": round(nnz, 1),
                       "parallelism": f"dp{world}",
                       "launch": "hipGraph replay (padded static shapes)" if use_graph else "eager"},
            "final_loss": round(float(loss), 6),
        }
        if not a.no_roofline:
            result["roofline"] = measure_scatter_roofline(model, host_batches[0].to(dev), dev)
            result["roofline"]["traffic"] = PMC_TRAFFIC_BYTES_PER_LAUNCH.get((a.method, a.batch, a.flavour))
            if result["roofline"]["traffic"] is not None:
                result["roofline"]["traffic_source"] = "profiles/r01_pmc_scatter_workload.json (rocprofv3 --pmc, offline)"
            if world == 1:
                result["roofline"]["saturation"] = saturation_probe(dev)
                if a.method == "egnn_equihnns":
                    result["roofline"]["edge_kernels"] = measure_edge_kernels(batches[0], dev)
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(a.method, args_ns, host_batches[0], a.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
