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
        if name.endswith("basis:(1,1)") or leaf == "beta":
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


# ------------------------------------------------------------------------------------------
# the cases: ONE table shared by the generator (make_golden.py) and the tests.  A case's inputs are
# a pure function of its row (this repo's own synthetic generator, no reference involved), so
# tests/test_golden_inputs.py re-derives them and compares them with what the .npz files hold.
# ------------------------------------------------------------------------------------------
# name: (method, hidden, seed, n_mols, train_mode, store_grads, options)
#   options: flavour ("qm9" | "pcqm"); last_conj (the last molecule is forced to carry a conjugated
#   hyperedge: the reference's mhnn / egnn_equihnn crash otherwise, mhnn.py:72-73 -- the six first-written
#   fixture families predate that rule and keep their frozen generator); big (atoms of one extra-large
#   molecule, PCQM4Mv2 / Molecule3D go up to ~60 heavy+H atoms); geometry ("degenerate": coincident atoms,
#   an edge along exactly -y, edges within 1e-3 rad of -y: equiformer/basis.py:169-191); dropout0 (FAFormer's
#   0.1 dropouts, active in training mode whatever --dropout says, forced to 0 so that a TRAIN-mode fixture
#   is deterministic); depth (Equiformer depth, for the type-1 path).
CASE_TABLE = {
    "mhnnm_c64_train": ("mhnnm", 64, 11, 6, True, True, dict(last_conj=False)),
    "mhnnm_c64_eval": ("mhnnm", 64, 12, 5, False, True, dict(last_conj=False)),
    "mhnnm_c256_train": ("mhnnm", 256, 13, 4, True, False, dict(last_conj=False)),
    "egnn_equihnns_c64": ("egnn_equihnns", 64, 21, 6, True, True, dict(last_conj=False)),
    "egnn_equihnns_c64_b": ("egnn_equihnns", 64, 22, 10, True, True, dict(last_conj=False)),
    "egnn_equihnns_c256": ("egnn_equihnns", 256, 23, 5, True, False, dict(last_conj=False)),
    "mhnn_c64": ("mhnn", 64, 41, 6, True, True, {}),
    "mhnns_c64": ("mhnns", 64, 42, 6, True, True, {}),
    "egnn_equihnn_c64": ("egnn_equihnn", 64, 43, 6, True, True, {}),
    "egnn_equihnnm_c64": ("egnn_equihnnm", 64, 44, 6, True, True, {}),
    "faformer_equihnns_c64": ("faformer_equihnns", 64, 51, 6, False, True, {}),
    "faformer_equihnns_c64_b": ("faformer_equihnns", 64, 52, 3, False, True, {}),
    "faformer_equihnns_c256": ("faformer_equihnns", 256, 53, 2, False, False, {}),
    "equiformer_equihnns_c64": ("equiformer_equihnns", 64, 31, 6, True, True, dict(last_conj=False)),
    "equiformer_equihnns_c64_b": ("equiformer_equihnns", 64, 32, 3, True, True, dict(last_conj=False)),
    "equiformer_equihnns_c256": ("equiformer_equihnns", 256, 33, 2, True, False, dict(last_conj=False)),
    # round 2: the PCQM4Mv2-like / Molecule3D-like workloads of BASELINE configs 4 and 5
    "egnn_equihnns_pcqm_c64": ("egnn_equihnns", 64, 61, 6, True, True, dict(flavour="pcqm", big=44)),
    "egnn_equihnns_pcqm_c256": ("egnn_equihnns", 256, 62, 4, True, False, dict(flavour="pcqm", big=52)),
    "faformer_equihnns_pcqm_c64": ("faformer_equihnns", 64, 63, 6, False, True, dict(flavour="pcqm", big=44)),
    "faformer_equihnns_pcqm_c256": ("faformer_equihnns", 256, 64, 3, False, False, dict(flavour="pcqm", big=41)),
    "mhnnm_pcqm_c64_train": ("mhnnm", 64, 65, 6, True, True, dict(flavour="pcqm", big=44)),
    "equiformer_equihnns_pcqm_c64": ("equiformer_equihnns", 64, 66, 4, True, True, dict(flavour="pcqm", big=41)),
    # FAFormer in TRAINING mode with its dropouts forced to p = 0
    "faformer_equihnns_c64_train_p0": ("faformer_equihnns", 64, 67, 5, True, True, dict(dropout0=True)),
    # a cloud of <= 16 atoms (one small molecule + the one-atom molecule: 10 atoms): fa_former_layer.py:664,697 takes min(16, N)
    # neighbours, equiformer_layer.py:1317-1323 min(16, N - 1)
    "faformer_equihnns_tiny": ("faformer_equihnns", 64, 71, 1, False, True, dict(last_conj=False)),
    "equiformer_equihnns_tiny": ("equiformer_equihnns", 64, 72, 1, True, True, dict(last_conj=False)),
    # degenerate edge geometry for the Equiformer's D construction
    "equiformer_equihnns_c64_degenerate": ("equiformer_equihnns", 64, 68, 5, True, True, dict(geometry="degenerate")),
    # round 5: configurations the reference accepts but its scripts never set (`args`: overrides of golden_args) -- BatchNorm
    # inside the MLPs (mlp.py:29-44: --normalization bn), zero-layer MLPs (conv.py:33-34,45-46,57-58,69-70,128-130,142-143:
    # slices / identity) and one-layer MLPs (a single Linear: mlp.py:30-36)
    "egnn_equihnns_c64_bn": ("egnn_equihnns", 64, 73, 6, True, True, dict(last_conj=False, args=dict(normalization="bn"))),
    "mhnnm_c64_bn_train": ("mhnnm", 64, 74, 6, True, True, dict(last_conj=False, args=dict(normalization="bn"))),
    "egnn_equihnns_c64_mlp0": ("egnn_equihnns", 64, 75, 6, True, True,
                               dict(last_conj=False, args=dict(MLP1_num_layers=0, MLP2_num_layers=0))),
    "mhnnm_c64_mlp0_13": ("mhnnm", 64, 76, 6, True, True, dict(last_conj=False, args=dict(MLP1_num_layers=0, MLP3_num_layers=0))),
    "mhnnm_c64_mlp0_24": ("mhnnm", 64, 77, 6, True, True, dict(last_conj=False, args=dict(MLP2_num_layers=0, MLP4_num_layers=0))),
    "egnn_equihnns_c64_mlp1": ("egnn_equihnns", 64, 78, 6, True, True,
                               dict(last_conj=False, args=dict(MLP1_num_layers=1, MLP2_num_layers=1, MLP3_num_layers=1))),
}


# Training trajectories: the reference's own step -- forward, nn.MSELoss, backward,
# Adam(model.parameters(), lr, weight_decay) (main.py:36,49-63,137-140) -- over `steps` different batches.
# name: (method, hidden, seed, n_mols, steps, lr); batch t is make_batch(seed + t, n_mols).
TRAJECTORY_TABLE = {
    "trajectory_egnn_equihnns_c64": ("egnn_equihnns", 64, 81, 6, 3, 1e-3),
    "trajectory_mhnnm_c64": ("mhnnm", 64, 85, 6, 3, 1e-3),
    # round 5: seed 107 keeps every ReLU input of all three steps >= 2.4e-5 rms from its kink on the reference (the widest of
    # seeds 101-116, `make_golden.py --scan trajectory_egnn_equihnns_c64 101 ... 116`); the DEFAULT (panel) path is held to the
    # tight bounds on it -- seed 81 above, whose margin is 1e-5, stays as the documented kink case
    "trajectory_egnn_equihnns_c64_b": ("egnn_equihnns", 64, 107, 6, 3, 1e-3),
}


# The reference's Equiformer LAYER class at a depth its wrapper never uses (full type-1 outputs, the
# (0->1) / (1->1) attention pairs, the (1,1) basis contraction): feats [N, C] and coordinates in, type-0 and type-1
# features out, plus the gradients of a fixed linear functional of both outputs.
# name: (hidden, depth, seed, atoms)
LAYER_TABLE = {
    "equiformer_layer_depth2_c32": (32, 2, 91, 60),
    "equiformer_layer_depth1_c32": (32, 1, 92, 44),
    "equiformer_layer_depth3_c64": (64, 3, 93, 70),
}


# Gradients pinned to the REFERENCE ITSELF in float64 (round 5): the reference's own model class, `.double()`, on the case's
# batch with float64 coordinates / targets -- the rounding-free value of the reference's algorithm, so the HIP path is held
# to 5e-5 of the largest gradient entry against the reference (not against this repo's oracle) and the 1e-2 fp32 tolerance
# of the hidden-256 / FAFormer cases is retired.  A float32 evaluation reproduces a gradient only while no ReLU input lies
# within its rounding distance (~1e-6 rms) of zero, so the seeds are the first ones from `seed0` on whose closest ReLU
# input (measured ON THE REFERENCE in float64, make_golden._ReluMargin) is at least F64_MIN_MARGIN rms away from the kink
# (`make_golden.py --scan-f64` prints the scan); the chosen seed is part of the row.
# name: (method, hidden, seed, n_mols, train_mode, options)
F64_MIN_MARGIN = 1e-5
F64_TABLE = {
    "egnn_equihnns_c256_f64": ("egnn_equihnns", 256, 2330, 5, True, dict(last_conj=False)),
    "mhnnm_c256_train_f64": ("mhnnm", 256, 1309, 4, True, dict(last_conj=False)),
    "equiformer_equihnns_c256_f64": ("equiformer_equihnns", 256, 3301, 2, True, dict(last_conj=False)),
    "faformer_equihnns_c256_f64": ("faformer_equihnns", 256, 5304, 2, False, {}),
    "egnn_equihnns_pcqm_c256_f64": ("egnn_equihnns", 256, 6209, 4, True, dict(flavour="pcqm", big=52)),
    "faformer_equihnns_pcqm_c256_f64": ("faformer_equihnns", 256, 6417, 3, False, dict(flavour="pcqm", big=41)),
    "faformer_equihnns_c64_f64": ("faformer_equihnns", 64, 5103, 6, False, {}),
}
F64_FULL_LIMIT = 20000      # gradients of at most this many entries are stored whole, larger ones as a sample
F64_SAMPLE = 4096


def f64_spec(name: str) -> dict:
    method, hidden, seed, n_mols, train, opt = F64_TABLE[name]
    spec = dict(name=name, method=method, hidden=hidden, seed=seed, n_mols=n_mols, train=train, store_grads=True,
                flavour="qm9", last_conj=True, big=None, geometry=None, dropout0=False, depth=1, args={})
    spec.update(opt)
    return spec


def f64_sample_indices(numel: int) -> np.ndarray:
    """Flat indices of the stored sample of a large gradient: F64_SAMPLE entries evenly spread over the tensor."""
    return np.unique(np.linspace(0, numel - 1, F64_SAMPLE).astype(np.int64))


def layer_inputs(name: str):
    """(feats [N, C], coors [N, 3], w0 [N, C], w1 [N, C, 3]) of a layer case: a molecule-like cloud (chain growth, so
    every atom has in-radius neighbours; a far-away group has none) and the weights of the scalar functional."""
    hidden, depth, seed, n = LAYER_TABLE[name]
    g = np.random.default_rng(seed)
    pos = np.zeros((n, 3))
    for i in range(1, n):
        u = g.standard_normal(3)
        pos[i] = pos[int(g.integers(0, i))] + (1.4 + 0.1 * g.uniform(-1, 1)) * u / np.linalg.norm(u)
    pos[-3:] += 40.0                                             # three atoms beyond everyone's radius
    t = lambda a: torch.from_numpy(a.astype(np.float32))
    return (t(g.standard_normal((n, hidden))), t(pos), t(g.standard_normal((n, hidden))),
            t(g.standard_normal((n, hidden, 3))))


def trajectory_batches(name: str):
    method, hidden, seed, n_mols, steps, lr = TRAJECTORY_TABLE[name]
    return [make_batch(seed + t, n_mols) for t in range(steps)]


def case_spec(name: str) -> dict:
    method, hidden, seed, n_mols, train, store, opt = CASE_TABLE[name]
    spec = dict(name=name, method=method, hidden=hidden, seed=seed, n_mols=n_mols, train=train, store_grads=store,
                flavour="qm9", last_conj=True, big=None, geometry=None, dropout0=False, depth=1, args={})
    spec.update(opt)
    return spec


def _kth_neighbour_ties(pos: np.ndarray, k: int = 16) -> int:
    """Rows whose k-th and (k+1)-th nearest other points are exactly equidistant in fp32 (torch.topk's pick among
    them is implementation-defined, so a fixture must not contain such rows)."""
    n = pos.shape[0]
    bad = 0
    for i in range(n):
        d = pos[i] - pos
        dd = ((d[:, 0] * d[:, 0]).astype(np.float32) + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
        dd = np.sqrt((dd + (d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)).astype(np.float32)
        dd[i] = np.inf
        srt = np.sort(dd)
        bad += int(n > k + 1 and srt[k - 1] == srt[k])
    return bad


def degenerate_positions(pos: np.ndarray, batch: np.ndarray) -> np.ndarray:
    """Edit a batch's coordinates so that the Equiformer's neighbour graph contains the cases
    rot_x_to_y_direction treats specially (equiformer/basis.py:169-191): two coincident atoms (rel_pos = 0),
    one edge along exactly -y (and its reverse, exactly +y), and edges 2e-4 .. 9e-4 rad away from -y, inside
    the |x_hat + y_hat|^2 < 1e-6 clamp.  Deterministic; atoms of molecules 0 and 1 are moved.  The coincident pair is
    the first one (scanning molecule 0) that leaves NO atom with the two copies tied at its 16th-neighbour boundary."""
    first = lambda b: int(np.flatnonzero(batch == b)[0])
    a, c = first(0), first(1)
    for shift in range(6):
        out = pos.copy()
        out[a + shift + 1] = out[a + shift]                                                  # coincident pair
        out[a + (shift + 3) % 8] = out[a + (shift + 2) % 8] + np.array([0.0, -1.25, 0.0], np.float32)   # exactly -y
        out[c + 1] = out[c] + np.array([3.0e-4, -1.3, 2.0e-4], np.float32)                   # s = |x_hat + y_hat|^2 ~ 7.7e-8
        out[c + 3] = out[c + 2] + np.array([-6.0e-4, -1.2, 4.0e-4], np.float32)              # s ~ 3.6e-7 (largest deviation)
        out[c + 5] = out[c + 4] + np.array([9.0e-4, -1.1, -6.0e-4], np.float32)              # s ~ 9.7e-7 (just inside)
        out = out.astype(np.float32)
        if _kth_neighbour_ties(out) == 0:
            return out
    raise RuntimeError("degenerate_positions: no tie-free placement found")


def make_batch(spec_or_seed, n_mols=None, flavour="qm9", with_isolated=True, last_conj=True, big=None,
               geometry=None):
    """The synthetic batch of a case: seeded molecules (equihgnn_amd.batch.synth_molecule), molecule 0 with a
    conjugated hyperedge of order >= 3, an isolated one-atom molecule in the middle, optionally one large
    molecule and the degenerate-geometry edit."""
    from equihgnn_amd.batch import collate, synth_molecule

    if isinstance(spec_or_seed, dict):
        sp = spec_or_seed
        seed, n_mols, flavour, last_conj, big, geometry = (sp["seed"], sp["n_mols"], sp["flavour"], sp["last_conj"],
                                                           sp["big"], sp["geometry"])
    else:
        seed = spec_or_seed
    rng = np.random.default_rng(seed)
    mols = [synth_molecule(rng, flavour) for _ in range(n_mols)]
    mols[0] = synth_molecule(rng, flavour, n_atoms=9, force_conj=True)  # conj hyperedge, order>=3
    if last_conj:
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
    if big:
        mols[1] = synth_molecule(rng, flavour, n_atoms=int(big), force_conj=True)
    out = collate(mols)
    if geometry == "degenerate":
        out.pos = torch.from_numpy(degenerate_positions(out.pos.numpy(), out.batch.numpy()))
    elif geometry is not None:
        raise ValueError(geometry)
    return out


def zero_dropouts(model: torch.nn.Module) -> None:
    """Force every dropout probability inside a model to 0 (nn.Dropout modules in the reference and the oracle,
    the ``p`` / ``attn_drop`` attributes of this repo's FAFormer)."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        else:
            for attr in ("p", "attn_drop"):
                if isinstance(getattr(m, attr, None), float):
                    setattr(m, attr, 0.0)


def assert_close(got, ref, tol=1e-5, what=""):
    """The north_star tolerance, element by element: |got - ref| <= tol * max(1, |ref|) -- ABSOLUTE 1e-5 wherever
    |ref| <= 1, the same relative budget above."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if got.size == 0:
        return
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    worst = int(np.argmax(err))
    assert float(err.reshape(-1)[worst]) <= tol, (
        f"{what}: |got - ref| / max(1, |ref|) = {err.reshape(-1)[worst]:.3e} > {tol:g} at flat index {worst} "
        f"(got {got.reshape(-1)[worst]:.8g}, ref {ref.reshape(-1)[worst]:.8g})")


def load_case(name: str) -> dict:
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        case = {k: z[k] for k in z.files}
    case["meta_name"] = np.array(name)      # (not stored: the file name is the case name)
    return case


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
