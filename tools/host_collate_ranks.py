#!/usr/bin/env python3
"""The host side of the N-GPU data-parallel run, measured without GPUs: N loader processes side by side, each exactly
what one rank of `bench.py --gpus N` / `fit.Fitter` runs on the host -- a MolStore shard read with DistributedSampler
semantics, `fit.BucketedLoader`'s prefetch thread collating 256-molecule batches into packed staging buffers (array
operations, main.py:227-229's per-molecule Python collate replaced), a consumer thread taking them -- with
OMP_NUM_THREADS = cores // N as `bench.launch_ranks` sets it.  Nothing touches a device: the question is whether N ranks'
collate threads, competing for the box's cores, each still outrun the rate ONE GPU consumes (BENCH_r04: 191 k
molecules/s), so that the >= 6x scaling target is not lost on the CPU (risk vi).

    python tools/host_collate_ranks.py --ranks 8 --seconds 8 --out profiles/r05_host_8rank_collate.json
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def usable_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def worker(rank, ranks, batch, seconds, flavour, molecules, barrier, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    from equihgnn_amd.batch import MolStore, synth_molecule
    from equihgnn_amd.fit import BucketedLoader
    torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "1")))
    rng = np.random.default_rng(4242)          # every rank holds the same dataset and reads its own shard of it
    store = MolStore([synth_molecule(rng, flavour) for _ in range(molecules)])
    loader = BucketedLoader(store, batch, True, seed=1, device=None, rank=rank, world=ranks)
    n = 0
    for b in loader:                           # first epoch: allocates the staging ring (as a run's first epoch does)
        n += 1
    barrier.wait()
    c0, s0 = loader.collated, loader.collate_seconds
    t0 = time.perf_counter()
    mols = 0
    while time.perf_counter() - t0 < seconds:
        for b in loader:
            mols += b.num_real_graphs if getattr(b, "num_real_graphs", None) else batch
            if time.perf_counter() - t0 >= seconds:
                break
    el = time.perf_counter() - t0
    loader.close()
    q.put({"rank": rank, "consumed_molecules_per_s": mols / el,
           "collate_thread_molecules_per_s": (loader.collated - c0) / max(loader.collate_seconds - s0, 1e-9),
           "batches_first_epoch": n})


def run(ranks, batch, seconds, flavour, molecules):
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(max(1, cores // ranks))      # what bench.launch_ranks gives every rank
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(ranks), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, ranks, batch, seconds, flavour, molecules, barrier, q)) for r in range(ranks)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join()
    res.sort(key=lambda d: d["rank"])
    return cores, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--flavour", default="qm9")
    ap.add_argument("--molecules", type=int, default=256 * 8 * 12, help="dataset size: 12 batches per rank and epoch at 8 ranks")
    ap.add_argument("--gpu-rate", type=float, default=191042.9, help="molecules/s one GPU consumes (BENCH_r04.json headline)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cores, res = run(a.ranks, a.batch, a.seconds, a.flavour, a.molecules)
    rates = [r["consumed_molecules_per_s"] for r in res]
    out = {"what": f"{a.ranks} host-only loader processes side by side (MolStore shard -> BucketedLoader prefetch thread -> packed staging "
                   f"buffers -> consumer), batch {a.batch}/rank, {a.flavour}-like molecules, no device",
           "cores": cores, "ranks": a.ranks, "omp_num_threads_per_rank": max(1, cores // a.ranks), "seconds": a.seconds,
           "per_rank_molecules_per_s": [round(r, 1) for r in rates],
           "per_rank_collate_thread_molecules_per_s": [round(r["collate_thread_molecules_per_s"], 1) for r in res],
           "min_molecules_per_s": round(min(rates), 1), "sum_molecules_per_s": round(sum(rates), 1),
           "one_gpu_consumes_molecules_per_s": a.gpu_rate, "min_over_gpu_rate": round(min(rates) / a.gpu_rate, 3),
           "floor": 1.3, "holds": bool(min(rates) >= 1.3 * a.gpu_rate)}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
