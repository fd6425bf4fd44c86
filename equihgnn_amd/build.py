"""Build libequihgnn_hip.so (gfx950) in-tree with hipcc.

    python -m equihgnn_amd.build          # incremental
    python -m equihgnn_amd.build --force

The .so is git-ignored but travels with the gpurun snapshot; nothing is JIT-compiled at run time.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libequihgnn_hip.so")
# measurement build of the panel kernels (-DPN_STAMPS: per-wavefront s_memtime stamps around the phases of a panel's life);
# bench.py loads it BESIDE the product library to time the aggregation prologues that have no launch of their own
STAMPS_LIB = os.path.join(PKG, "libequihgnn_panel_stamps.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libequihgnn_hip.so)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMPS_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(STAMPS_LIB))
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    bdir = os.path.join(PKG, "build")
    os.makedirs(bdir, exist_ok=True)
    hipcc = _hipcc()
    common = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", INCLUDE, "-I", CSRC,
              "-ffp-contract=off",  # a*b+c stays two roundings unless written fmaf(): the kNN
                                    # distances must match the reference's fp32 CPU arithmetic
              "-Wall", "-Wno-unused-function"]
    procs = []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(h)
                        for h in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")))):
            continue
        cmd = [hipcc, *common, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    cmd = [hipcc, *common, "-shared", "-DPN_STAMPS", os.path.join(CSRC, "panel.hip"), os.path.join(CSRC, "api.hip"), "-o", STAMPS_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
