"""Shared by the golden-vector generator and the tests that consume the vectors.

Weights are never stored: both sides derive them from a seed with numpy's PCG64
generator, walking ``state_dict()`` in its (deterministic) key order.  Every branch is made
live (the reference initialises EGNN Linears at N(0,1e-3) and zero-initialises several
Equiformer projections, which would make a fixture test nothing — SURVEY.md §9).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))


def golden_args(method: str, hidden: int, **kw) -> SimpleNamespace:
    a = dict(method=method, All_num_layers=3, MLP1_num_layers=2, MLP2_num_layers=2,
             MLP3_num_layers=2, MLP4_num_layers=2, output_num_layers=3, MLP_hidden=hidden,
             output_hidden=hidden // 2, aggregate="mean", normalization="ln",
             activation="relu", dropout=0.0, lr=1e-4, wd=0.0, batch_size=8)
    a.update(kw)
    return SimpleNamespace(**a)


def fill_state_dict(model: torch.nn.Module, seed: int) -> None:
    """Overwrite every floating-point parameter/buffer with seeded values, in place.  Each tensor
    has its own generator keyed by (seed, crc32(name)), so the values do not depend on the order in
    which an implementation registers its parameters."""
    import zlib

    sd = model.state_dict()
    new = {}
    for name, t in sd.items():
        if not torch.is_floating_point(t):
            new[name] = t.clone()
            continue
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        z = rng.standard_normal(shape if len(shape) else (1,)).astype(np.float32).reshape(shape)
        if name == "equiformer_layer.basis:(1,1)" or leaf == "beta":
            new[name] = t.clone()  # data / fixed-zero buffers stay as constructed
        elif leaf == "running_var":
            v = np.abs(z) + 0.5
            new[name] = torch.from_numpy(v)
        elif leaf in ("running_mean", "bias"):
            new[name] = torch.from_numpy(0.1 * z)
        elif leaf in ("gamma", "scale") or ".transforms." in name or (leaf == "weight" and len(shape) == 1):
            new[name] = torch.from_numpy(1.0 + 0.1 * z)  # norm scales
        elif "embedding" in name or "bond_encoder" in name:
            new[name] = torch.from_numpy(0.5 * z)
        elif len(shape) >= 2:
            fan_in = shape[1] if leaf == "weight" else shape[0]  # nn.Linear: [out,in]; eq. Linear: [in,out]
            new[name] = torch.from_numpy(z / np.sqrt(max(fan_in, 1)))
        else:
            new[name] = torch.from_numpy(0.1 * z)
    model.load_state_dict(new, strict=True)


def load_case(name: str) -> dict:
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def batch_from_case(case: dict):
    """Rebuild the HBatch-like namespace from a stored case."""
    from equihgnn_amd.batch import HBatch

    t = lambda k: torch.from_numpy(case["in_" + k])
    return HBatch(x=t("x"), pos=t("pos"), edge_index0=t("edge_index0"),
                  edge_index1=t("edge_index1"), edge_attr=t("edge_attr"), n_e=t("n_e"),
                  e_order=t("e_order"), batch=t("batch"), y=t("y"),
                  num_nodes=int(case["in_x"].shape[0]),
                  num_hyperedges=int(case["in_edge_attr"].shape[0]),
                  num_graphs=int(case["in_y"].shape[0]))
