#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own model
files (read from /root/reference at run time, never copied) on seeded synthetic batches.

Run in the build container only:   python tests/golden/make_golden.py
(the GPU box has no /root/reference; it consumes the committed .npz files).

The reference's third-party dependencies are absent from this image.  The
stand-ins below are this repo's own code, written from the packages' documented semantics;
they exist only so that the reference's files import.  What is pinned by the vectors is the
reference's own arithmetic in equihgnn/models/** (wrappers, MHNN(S)Conv, MLP, EGNN,
Equiformer); the stand-ins are cross-checked by property tests in tests/test_standins.py.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("EQUIHGNN_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from common import fill_state_dict, golden_args  # noqa: E402

from equihgnn_amd.batch import ATOM_FEATURE_DIMS, collate, synth_molecule  # noqa: E402


# ------------------------------------------------------------------------------------------
# stand-ins for absent third-party packages
# ------------------------------------------------------------------------------------------
def _standin_scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter for the call pattern of conv.py (1-D index along ``dim``)."""
    assert out is None and index.dim() == 1
    dim = dim if dim >= 0 else src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    shape = list(src.shape)
    shape[dim] = dim_size
    res = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, idx, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        cnt = torch.bincount(index, minlength=dim_size).clamp(min=1).to(src.dtype)
        return res / cnt.view(view)
    raise ValueError(reduce)


def _standin_global_add_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return _standin_scatter(x, batch, dim=-2, dim_size=size, reduce="sum")


class _StandinAtomEncoder(torch.nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.atom_embedding_list = torch.nn.ModuleList()
        for d in ATOM_FEATURE_DIMS:
            emb = torch.nn.Embedding(d, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)

    def forward(self, x):
        acc = 0
        for i in range(x.shape[1]):
            acc += self.atom_embedding_list[i](x[:, i])
        return acc


def _standin_get_at(pattern, tensor, indices):
    """einx.get_at for the three patterns the Equiformer uses
    (equiformer_layer.py:356,1331-1335): gather along the bracketed axis."""
    pat = pattern.replace(" ", "")
    if pat == "b[i]dm,bjk->bjkdm":
        b = torch.arange(tensor.shape[0], device=tensor.device)[:, None, None]
        return tensor[b, indices]
    if pat == "bi[j]c,bik->bikc":
        return tensor.gather(2, indices[..., None].expand(*indices.shape, tensor.shape[-1]))
    if pat == "bi[j],bik->bik":
        return tensor.gather(2, indices)
    raise NotImplementedError(pattern)


def _standin_to_dense_batch(x, batch=None, fill_value=0.0, max_num_nodes=None, batch_size=None):
    """torch_geometric.utils.to_dense_batch: [N, ...] + sorted batch vector -> [B, Nmax, ...], mask."""
    if batch is None:
        batch = x.new_zeros(x.shape[0], dtype=torch.long)
    b = int(batch.max()) + 1 if batch_size is None else batch_size
    counts = torch.bincount(batch, minlength=b)
    nmax = int(counts.max()) if max_num_nodes is None else max_num_nodes
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(x.shape[0]) - start[batch]
    out = x.new_full((b, nmax, *x.shape[1:]), fill_value)
    mask = torch.zeros(b, nmax, dtype=torch.bool)
    out[batch, pos] = x
    mask[batch, pos] = True
    return out, mask


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_standins():
    _module("torch_scatter", scatter=_standin_scatter)
    _module("ogb")
    _module("ogb.graphproppred")
    _module("ogb.graphproppred.mol_encoder", AtomEncoder=_StandinAtomEncoder)
    tg = _module("torch_geometric")
    tg.nn = _module("torch_geometric.nn", global_add_pool=_standin_global_add_pool)
    tg.utils = _module("torch_geometric.utils", to_dense_batch=_standin_to_dense_batch)
    import typing

    _module("beartype", beartype=lambda f: f)
    bt = _module("beartype.typing")
    for k in ("Dict", "Optional", "Tuple", "Union", "List", "Callable"):
        setattr(bt, k, getattr(typing, k))
    _module("einx", get_at=_standin_get_at)
    _module("opt_einsum", contract=lambda eq, *ops, **kw: torch.einsum(eq, *ops))

    class _Never(torch.nn.Module):  # num_linear_attn_heads=0 -> never instantiated
        def __init__(self, *a, **k):
            raise RuntimeError("TaylorSeriesLinearAttn stand-in must not be constructed")

    _module("taylor_series_linear_attention", TaylorSeriesLinearAttn=_Never)


def reconstructed_J():
    """The reference's data/J_dense.pt is a missing blob (.MISSING_LARGE_BLOBS).  J_l is the
    real-spherical-harmonic representation of the x<->y... axis swap used by the ZYZ Wigner-D
    construction (irr_repr.py:23-32).  SURVEY.md §8c documents the reconstruction and its
    checks; J2 only feeds the (1,1) basis, which is dead for the type-0 output."""
    s3 = np.sqrt(3.0) / 2.0
    J0 = torch.tensor([[1.0]], dtype=torch.float64)
    J1 = torch.tensor([[0, 1, 0], [1, 0, 0], [0, 0, -1]], dtype=torch.float64)
    J2 = torch.tensor([[0, 0, 0, -1, 0], [0, 1, 0, 0, 0], [0, 0, -0.5, 0, -s3],
                       [-1, 0, 0, 0, 0], [0, 0, -s3, 0, 0.5]], dtype=torch.float64)
    return [J0, J1, J2]


def import_reference(names=("equihnn_egnn", "mhnn")):
    """Import the reference's model files without executing equihgnn/models/__init__.py
    (which eagerly imports PyG/torch_cluster-dependent models)."""
    install_standins()
    os.environ["CLEAR_CACHE"] = "1"
    sys.path.insert(0, REF)
    for pkg, sub in (("equihgnn.models", "equihgnn/models"),
                     ("equihgnn.models.layers", "equihgnn/models/layers"),
                     ("equihgnn.models.layers.equiformer", "equihgnn/models/layers/equiformer")):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[pkg] = m
    if "equihnn_equiformer" in names:
        real_load = torch.load

        def patched_load(path, *a, **k):
            if str(path).endswith("J_dense.pt"):
                return reconstructed_J()
            return real_load(path, *a, **k)

        torch.load = patched_load
        try:
            importlib.import_module("equihgnn.models.layers.equiformer.irr_repr")
        finally:
            torch.load = real_load
    for n in names:
        importlib.import_module("equihgnn.models." + n)
    return importlib.import_module("equihgnn.common.registry").registry


# ------------------------------------------------------------------------------------------
# cases
# ------------------------------------------------------------------------------------------
def make_batch(seed, n_mols, flavour="qm9", with_isolated=True):
    rng = np.random.default_rng(seed)
    mols = [synth_molecule(rng, flavour) for _ in range(n_mols)]
    mols[0] = synth_molecule(rng, flavour, n_atoms=9, force_conj=True)  # conj hyperedge, order>=3
    # the reference's mhnn / egnn_equihnn fail unless the LAST molecule has a hyperedge of order > 2
    mols[-1] = synth_molecule(rng, flavour, force_conj=True)
    if with_isolated:
        lone = synth_molecule(rng, flavour, n_atoms=3, force_conj=False)
        # a one-atom, zero-hyperedge molecule: its node row has no incidence at all
        lone.x, lone.pos = lone.x[:1], np.zeros((1, 3), np.float32) + 0.25
        lone.edge_index0 = lone.edge_index0[:0]
        lone.edge_index1 = lone.edge_index1[:0]
        lone.edge_attr = lone.edge_attr[:0]
        lone.e_order = lone.e_order[:0]
        mols.insert(n_mols // 2, lone)
    return collate(mols)


def run_case(registry, method, hidden, seed, n_mols, train_mode=True, store_grads=True,
             flavour="qm9"):
    torch.manual_seed(0)
    args = golden_args(method, hidden)
    model = registry.get_model_class(method)(1, args)
    fill_state_dict(model, seed)
    model.train(train_mode)
    data = make_batch(seed, n_mols, flavour)

    taps = {}
    hooks = []

    def tap(name, pick=lambda o: o):
        counter = {"n": 0}

        def fn(_m, _i, o):
            key = name if name != "conv" else f"conv{counter['n']}"
            counter["n"] += 1
            taps[key] = pick(o).detach().clone()
        return fn

    hooks.append(model.atom_encoder.register_forward_hook(tap("atom_encoder")))
    if hasattr(model, "egnn_layer"):
        hooks.append(model.egnn_layer.register_forward_hook(tap("front_end", lambda o: o[0][0])))
    if hasattr(model, "fa_former"):
        hooks.append(model.fa_former.register_forward_hook(tap("front_end", lambda o: o[0][0])))
    if hasattr(model, "equiformer_layer"):
        hooks.append(model.equiformer_layer.register_forward_hook(
            tap("front_end", lambda o: o.type0[0])))
    if hasattr(model, "conv"):
        # the wrapper applies act() after conv; tap the conv output itself
        # MHNNSConv returns X; MHNNConv returns (X, E): tap the node features
        hooks.append(model.conv.register_forward_hook(
            tap("conv", lambda o: (o[0] if isinstance(o, tuple) else o).reshape(-1, (o[0] if isinstance(o, tuple) else o).shape[-1]))))
    if hasattr(model, "batch_norms"):
        for i, bn in enumerate(model.batch_norms):
            hooks.append(bn.register_forward_hook(tap(f"bn{i}")))
    hooks.append(model.mlp_out.register_forward_pre_hook(
        lambda _m, i: taps.__setitem__("pool", i[0].detach().reshape(-1, i[0].shape[-1]).clone())))

    # record the neighbour lists the reference's own topk call produces
    real_topk = torch.Tensor.topk
    rec = {}

    def spy_topk(self, *a, **k):
        r = real_topk(self, *a, **k)
        rec.setdefault("idx", r.indices.detach().clone())
        rec.setdefault("val", r.values.detach().clone())
        return r

    torch.Tensor.topk = spy_topk
    try:
        out = model(data)
    finally:
        torch.Tensor.topk = real_topk
    loss = torch.nn.functional.mse_loss(out, data.y)
    loss.backward()
    for h in hooks:
        h.remove()

    case = {"meta_method": np.array(method), "meta_hidden": np.array(hidden),
            "meta_seed": np.array(seed), "meta_train": np.array(int(train_mode))}
    for k in ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "n_e", "e_order", "batch", "y"):
        case["in_" + k] = getattr(data, k).numpy()
    for k, v in taps.items():
        case["tap_" + k] = v.numpy()
    if "idx" in rec:
        case["knn_idx"] = rec["idx"].reshape(-1, rec["idx"].shape[-1]).numpy()
        case["knn_val"] = rec["val"].reshape(-1, rec["val"].shape[-1]).numpy()
    case["out"] = out.detach().numpy()
    case["loss"] = loss.detach().numpy()
    names, has_grad, stats = [], [], []
    for n, p in model.named_parameters():
        names.append(n)
        has_grad.append(p.grad is not None)
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        stats.append([float(g.sum()), float(g.abs().sum()), float(g.norm())])
        if store_grads and p.grad is not None and p.numel() <= 20000:
            case["grad_" + n] = g.numpy()
        elif p.grad is not None:
            flat = g.reshape(-1)
            case["gradhead_" + n] = flat[: min(64, flat.numel())].numpy()
    case["grad_names"] = np.array(names)
    case["grad_present"] = np.array(has_grad)
    case["grad_stats"] = np.array(stats, dtype=np.float64)
    if any(k.endswith("running_mean") for k in model.state_dict()):
        for k, v in model.state_dict().items():
            if "running_" in k:
                case["buf_" + k] = v.numpy()
    return case


CASES = [
    # name, method, hidden, seed, n_mols, train_mode, store_grads
    ("mhnnm_c64_train", "mhnnm", 64, 11, 6, True, True),
    ("mhnnm_c64_eval", "mhnnm", 64, 12, 5, False, True),
    ("mhnnm_c256_train", "mhnnm", 256, 13, 4, True, False),
    ("egnn_equihnns_c64", "egnn_equihnns", 64, 21, 6, True, True),
    ("egnn_equihnns_c64_b", "egnn_equihnns", 64, 22, 10, True, True),
    ("egnn_equihnns_c256", "egnn_equihnns", 256, 23, 5, True, False),
    ("mhnn_c64", "mhnn", 64, 41, 6, True, True),
    ("mhnns_c64", "mhnns", 64, 42, 6, True, True),
    ("egnn_equihnn_c64", "egnn_equihnn", 64, 43, 6, True, True),
    ("egnn_equihnnm_c64", "egnn_equihnnm", 64, 44, 6, True, True),
    ("faformer_equihnns_c64", "faformer_equihnns", 64, 51, 6, False, True),
    ("faformer_equihnns_c64_b", "faformer_equihnns", 64, 52, 3, False, True),
    ("faformer_equihnns_c256", "faformer_equihnns", 256, 53, 2, False, False),
    ("equiformer_equihnns_c64", "equiformer_equihnns", 64, 31, 6, True, True),
    ("equiformer_equihnns_c64_b", "equiformer_equihnns", 64, 32, 3, True, True),
    ("equiformer_equihnns_c256", "equiformer_equihnns", 256, 33, 2, True, False),
]


def main(only=None):
    torch.set_num_threads(8)
    methods = {c[1] for c in CASES if only is None or c[0] in only}
    mods = []
    if methods & {"mhnnm", "mhnn", "mhnns"}:
        mods.append("mhnn")
    if methods & {"egnn_equihnns", "egnn_equihnn", "egnn_equihnnm"}:
        mods.append("equihnn_egnn")
    if "equiformer_equihnns" in methods:
        mods.append("equihnn_equiformer")
    if "faformer_equihnns" in methods:
        mods.append("equihnn_fa_former")
    registry = import_reference(tuple(mods))
    for name, method, hidden, seed, n_mols, train_mode, store in CASES:
        if only is not None and name not in only:
            continue
        case = run_case(registry, method, hidden, seed, n_mols, train_mode, store)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **case)
        print(f"{name}: N={case['in_x'].shape[0]} M={case['in_edge_attr'].shape[0]} "
              f"nnz={case['in_edge_index0'].shape[0]} out[:3]={case['out'][:3]} "
              f"loss={float(case['loss']):.6f} -> {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main(set(sys.argv[1:]) or None)
