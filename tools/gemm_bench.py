#!/usr/bin/env python3
"""hg_gemm_x6_batch (fp32 GEMM on the bf16 matrix cores, csrc/gemm_x6.hip) against the library fp32 GEMM (torch.mm) on
the shapes of the BASELINE configs: microseconds and TFLOP/s (2 M N K), interleaved rounds in one process, median."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the library side gets what bench.py gives it: TunableOp's per-shape selection, tuned on first use (outside the graphs)
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "50")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(os.environ.get("TMPDIR", "/tmp"), "gemm_bench_tunableop_%d.csv"))
import torch

from equihgnn_amd import ops

# (name, M, N, K, trans_a, trans_b)
SHAPES = [("c2 conv fwd x W^T", 4736, 256, 256, 0, 1), ("c2 conv dgrad dY W", 4736, 256, 256, 0, 0),
          ("c2 conv wgrad dY^T X", 256, 256, 4736, 1, 0),
          ("c2 egnn ab fwd", 4736, 2176, 256, 0, 1), ("c2 egnn ab dgrad", 4736, 256, 2176, 0, 0),
          ("c2 egnn ab wgrad", 2176, 256, 4736, 1, 0), ("c2 node mlp", 4736, 512, 272, 0, 1),
          ("c4 conv fwd", 31232, 256, 256, 0, 1), ("c4 egnn ab fwd", 31232, 2176, 256, 0, 1),
          ("c3 P/Q node product", 2432, 16384, 64, 0, 1), ("c3 big", 38912, 1024, 256, 0, 1),
          ("c5 frame fc1", 1971840, 256, 128, 0, 1), ("c5 frame dgrad", 1971840, 128, 256, 0, 0),
          ("c5 frame wgrad", 256, 128, 1971840, 1, 0), ("c3 dgrad K=16384", 2312, 256, 16384, 0, 0),
          ("c3 wgrad N=16384", 256, 16384, 2312, 1, 0),
          ("c5 fc", 245760, 256, 256, 0, 1), ("c5 fc dgrad", 245760, 256, 256, 0, 0), ("c5 wgrad", 256, 256, 245760, 1, 0),
          ("square 4096", 4096, 4096, 4096, 0, 1),
          ("c4 egnn ab dgrad", 31232, 256, 2176, 0, 0), ("c4 egnn ab wgrad", 2176, 256, 31232, 1, 0),
          ("c5 node fwd", 15488, 256, 256, 0, 1), ("c5 ffn up", 15488, 1024, 512, 0, 1), ("c5 qkv", 15488, 768, 256, 0, 1),
          ("c5 mid wgrad", 1024, 512, 15488, 1, 0),
          ("c3 attn dgrad K=4096", 2432, 256, 4096, 0, 0), ("c3 attn wgrad N=4096", 256, 4096, 2432, 1, 0),
          ("c3 attn dgrad 3N rows", 7296, 256, 4096, 0, 0), ("c3 attn wgrad 3N rows", 256, 4096, 7296, 1, 0),
          ("c3 attn fwd", 2432, 4096, 256, 0, 0), ("c3 attn fwd 3N", 7296, 4096, 256, 0, 0),
          ("c3 pooled fwd Y W^T", 2432, 256, 16384, 0, 1), ("c3 pooled dgrad dp W", 2432, 16384, 256, 0, 0),
          ("c3 P/Q stacked 2N rows", 4864, 16384, 256, 0, 0), ("c3 dgrad stacked 2N", 4864, 256, 16384, 0, 1),
          ("c3 wgrad stacked 2N", 256, 16384, 4864, 1, 0)]


def timeit(fn, reps):
    """microseconds per call of `fn` inside a replayed hipGraph of `reps` back-to-back calls (device time: eager
    launches of 10-microsecond kernels measure the host)."""
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default=None, help="substring of the shape names to run")
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    for name, M, N, K, ta, tb in SHAPES:
        if a.only and a.only not in name:
            continue
        A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
        B = torch.randn((N, K) if tb else (K, N), device=dev, generator=g)
        out = torch.empty(M, N, device=dev)
        Al, Bl = (A.t() if ta else A), (B.t() if tb else B)
        fl = 2.0 * M * N * K
        reps = max(3, min(50, int(2e10 / fl)))
        variants = {"lib": lambda: torch.mm(Al, Bl, out=out)}
        for tile in (0, 64, 128, 256, 512, 513):      # 0: the launch's own choice (hg_gemm_x6_choose_tile)
            def f(tile=tile):
                ops.GEMM_TILE = tile
                ops.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
            variants[f"x6_t{tile}"] = f
        for f in variants.values():
            f()
        torch.cuda.synchronize()
        t = {k: [] for k in variants}
        for _ in range(a.rounds):
            for k, f in variants.items():
                t[k].append(timeit(f, reps))
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        row = {"shape": name, "M": M, "N": N, "K": K, "trans_a": ta, "trans_b": tb,
               **{k + "_us": round(v, 2) for k, v in med.items()}, **{k + "_tflops": round(fl / v / 1e6, 1) for k, v in med.items()}}
        rows.append(row)
        print(f"{name:24s} M={M:7d} N={N:5d} K={K:7d}  " + "  ".join(f"{k}: {v:9.1f} us {fl / v / 1e6:6.1f} TF" for k, v in med.items()),
              flush=True)
    ops.GEMM_TILE = 0
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
