"""Molecular-hypergraph batch container, collate rule and synthetic generator.

The hot path consumes a PyG ``Batch`` of ``HData`` objects
(reference: equihgnn/data/utils.py:150-178).  PyG is not part of this image, so
this module carries the few fields the path reads, with the reference's batching
rule (``HData.__inc__``: ``edge_index0`` is offset by the number of nodes and
``edge_index1`` by ``n_e``, data/utils.py:172-178).  A real PyG ``Batch`` exposes
the same attribute names, so the model classes accept either.

Field layout (all row-major, on one device):
    x           [N, 9]  int64   ogb atom features
    pos         [N, 3]  float32 coordinates
    edge_index0 [nnz]   int64   node id of every incidence
    edge_index1 [nnz]   int64   hyperedge id of every incidence
    edge_attr   [M, 1]  int64   bond type 0-4, 5 = conjugated group
    n_e         [B]     int64   hyperedges per molecule
    e_order     [M]     int64   nodes per hyperedge
    batch       [N]     int64   molecule id per node (sorted)
    y           [B]     float32 regression target

The synthetic generator follows SURVEY.md §8(d): random spanning-tree molecules
with a few ring-closing bonds, at most one conjugated hyperedge, incidences in
the reference's order (data/utils.py:123-145: two incidences per bond, bond
hyperedges first, conjugated-group incidences after, in atom order).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, fields
from typing import List, Optional, Sequence

import numpy as np
import torch

# ogb 1.3.6 full_atom_feature_dims (ogb/utils/features.py), used by AtomEncoder.
ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)
NUM_BOND_TYPES = 6  # 0-4 bond types + 5 = conjugated hyperedge (data/utils.py:129-145)

# (mean atoms, std, max atoms, P(conjugated hyperedge)) per dataset flavour, SURVEY §8(d)
FLAVOURS = {
    "qm9": (18.0, 3.0, 29, 0.3),
    "pcqm": (30.0, 7.0, 60, 0.8),
}


@dataclass
class HBatch:
    x: torch.Tensor
    pos: torch.Tensor
    edge_index0: torch.Tensor
    edge_index1: torch.Tensor
    edge_attr: torch.Tensor
    n_e: torch.Tensor
    e_order: torch.Tensor
    batch: torch.Tensor
    y: torch.Tensor
    # host-side sizes, kept so the GPU path never needs a device sync to learn them
    num_nodes: int = 0
    num_hyperedges: int = 0
    num_graphs: int = 0

    def to(self, device, non_blocking: bool = False) -> "HBatch":
        flat = getattr(self, "_flat", None)
        if flat is not None:   # packed: ONE transfer, then the same views over the new buffer
            return self._from_flat(flat.to(device, non_blocking=non_blocking), self._layout)
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        out = HBatch(**kw)
        if hasattr(self, "num_real_graphs"):
            out.num_real_graphs = self.num_real_graphs
        return out

    def packed(self) -> "HBatch":
        """The same batch with every tensor field a view into ONE flat byte buffer (256-byte aligned
        slots): host-to-device staging and the refresh of a captured graph's static inputs become a
        single copy instead of one per field."""
        names = [f.name for f in fields(self) if torch.is_tensor(getattr(self, f.name))]
        layout, total = [], 0
        for n in names:
            t = getattr(self, n)
            layout.append((n, total, tuple(t.shape), t.dtype))
            total += (t.numel() * t.element_size() + 255) // 256 * 256
        flat = torch.empty(max(total, 256), dtype=torch.uint8, device=self.x.device)
        out = self._from_flat(flat, tuple(layout))
        for n in names:
            getattr(out, n).copy_(getattr(self, n))
        return out

    @classmethod
    def empty_packed(cls, n_nodes: int, n_hyperedges: int, n_inc: int, n_graphs: int, pin: bool = False) -> "HBatch":
        """An uninitialised packed host batch of the given extents (one flat buffer, optionally pinned): the staging
        buffer a loader thread collates into and the trainer's static inputs are refreshed from with ONE copy."""
        spec = (("x", (n_nodes, 9), torch.int64), ("pos", (n_nodes, 3), torch.float32),
                ("edge_index0", (n_inc,), torch.int64), ("edge_index1", (n_inc,), torch.int64),
                ("edge_attr", (n_hyperedges, 1), torch.int64), ("n_e", (n_graphs,), torch.int64),
                ("e_order", (n_hyperedges,), torch.int64), ("batch", (n_nodes,), torch.int64),
                ("y", (n_graphs,), torch.float32))
        layout, total = [], 0
        for name, shape, dtype in spec:
            layout.append((name, total, tuple(shape), dtype))
            nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            total += (nbytes + 255) // 256 * 256
        flat = torch.empty(max(total, 256), dtype=torch.uint8)
        if pin:
            flat = flat.pin_memory()
        proto = cls(**{name: torch.empty(0) for name, _, _ in spec}, num_nodes=n_nodes, num_hyperedges=n_hyperedges,
                    num_graphs=n_graphs)
        return proto._from_flat(flat, tuple(layout))

    def _from_flat(self, flat, layout) -> "HBatch":
        kw = {f.name: getattr(self, f.name) for f in fields(self) if not torch.is_tensor(getattr(self, f.name))}
        for n, off, shape, dtype in layout:
            nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            kw[n] = flat[off:off + nbytes].view(dtype).view(shape)
        out = HBatch(**kw)
        out._flat, out._layout = flat, layout
        if hasattr(self, "num_real_graphs"):
            out.num_real_graphs = self.num_real_graphs
        return out

    def pin_memory(self) -> "HBatch":
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v.pin_memory() if torch.is_tensor(v) else v
        return HBatch(**kw)

    @property
    def nnz(self) -> int:
        return int(self.edge_index0.shape[0])


@dataclass
class HMol:
    """One molecule before batching (the HData fields the path reads)."""

    x: np.ndarray            # [n, 9] int64
    pos: np.ndarray          # [n, 3] float32
    edge_index0: np.ndarray  # [nnz] int64, local node ids
    edge_index1: np.ndarray  # [nnz] int64, local hyperedge ids
    edge_attr: np.ndarray    # [m, 1] int64
    e_order: np.ndarray      # [m] int64
    y: float


def collate(mols: Sequence[HMol]) -> HBatch:
    """PyG ``Batch.from_data_list`` restricted to HData's fields.

    Offsets follow ``HData.__inc__`` (data/utils.py:172-178): node ids shift by the
    running node count, hyperedge ids by the running ``n_e``.
    """
    xs, ps, v, e, ea, eo, bt, ys, ne = [], [], [], [], [], [], [], [], []
    n_off = 0
    m_off = 0
    for b, mol in enumerate(mols):
        n = mol.x.shape[0]
        m = mol.edge_attr.shape[0]
        xs.append(mol.x)
        ps.append(mol.pos)
        v.append(mol.edge_index0 + n_off)
        e.append(mol.edge_index1 + m_off)
        ea.append(mol.edge_attr)
        eo.append(mol.e_order)
        bt.append(np.full((n,), b, dtype=np.int64))
        ys.append(mol.y)
        ne.append(m)
        n_off += n
        m_off += m
    return HBatch(
        x=torch.from_numpy(np.concatenate(xs, 0).astype(np.int64)),
        pos=torch.from_numpy(np.concatenate(ps, 0).astype(np.float32)),
        edge_index0=torch.from_numpy(np.concatenate(v, 0).astype(np.int64)),
        edge_index1=torch.from_numpy(np.concatenate(e, 0).astype(np.int64)),
        edge_attr=torch.from_numpy(np.concatenate(ea, 0).astype(np.int64)),
        n_e=torch.tensor(ne, dtype=torch.int64),
        e_order=torch.from_numpy(np.concatenate(eo, 0).astype(np.int64)),
        batch=torch.from_numpy(np.concatenate(bt, 0)),
        y=torch.tensor(ys, dtype=torch.float32),
        num_nodes=n_off,
        num_hyperedges=m_off,
        num_graphs=len(mols),
    )


class MolStore:
    """A dataset of molecules as a structure of arrays -- the concatenated fields plus per-molecule offsets -- so
    that a batch is assembled by array operations only (numpy gathers driven by ``cumsum`` offsets): no Python loop
    over molecules, which at 256-1024 molecules per batch and > 100 k molecules/s per GPU would be the bottleneck of
    the whole training step (the reference leaves this to PyG's ``Batch.from_data_list``, a per-molecule loop).
    Field semantics as HBatch / HData (data/utils.py:150-178)."""

    def __init__(self, mols: Sequence[HMol]):
        n = np.array([m.x.shape[0] for m in mols], dtype=np.int64)
        h = np.array([m.edge_attr.shape[0] for m in mols], dtype=np.int64)
        z = np.array([m.edge_index0.shape[0] for m in mols], dtype=np.int64)
        off = lambda c: np.concatenate(([0], np.cumsum(c)))
        self.n_nodes, self.n_he, self.n_inc = n, h, z
        self.node_off, self.he_off, self.inc_off = off(n), off(h), off(z)
        cat = lambda xs, dt, shape: (np.concatenate(xs, 0).astype(dt) if len(xs) else np.zeros(shape, dt))
        self.x = cat([m.x for m in mols], np.int64, (0, 9))
        self.pos = cat([m.pos for m in mols], np.float32, (0, 3))
        self.v = cat([m.edge_index0 for m in mols], np.int64, (0,))          # local node ids
        self.e = cat([m.edge_index1 for m in mols], np.int64, (0,))          # local hyperedge ids
        self.edge_attr = cat([m.edge_attr for m in mols], np.int64, (0, 1))
        self.e_order = cat([m.e_order for m in mols], np.int64, (0,))
        self.y = np.array([m.y for m in mols], dtype=np.float32)

    def __len__(self) -> int:
        return int(self.n_nodes.shape[0])

    def extents(self, idx) -> tuple:
        """(nodes, hyperedges, incidences) of the batch made of molecules ``idx``."""
        idx = np.asarray(idx, dtype=np.int64)
        return int(self.n_nodes[idx].sum()), int(self.n_he[idx].sum()), int(self.n_inc[idx].sum())

    @staticmethod
    def _ranges(starts, counts):
        """Concatenation of arange(starts[i], starts[i] + counts[i]) and the owner i of every element."""
        total = int(counts.sum())
        owner = np.repeat(np.arange(counts.shape[0], dtype=np.int64), counts)
        first = np.concatenate(([0], np.cumsum(counts)[:-1]))
        return starts[owner] + (np.arange(total, dtype=np.int64) - first[owner]), owner

    def collate(self, idx, pad_to: Optional[tuple] = None, out: Optional["HBatch"] = None) -> "HBatch":
        """The batch of molecules ``idx`` (HData.__inc__ offsets, data/utils.py:172-178), optionally padded to the
        static extents ``pad_to`` = (nodes, hyperedges, incidences) exactly as ``pad_batch`` does (one dummy molecule
        owns the padding, padded incidences are null), optionally written into the tensors of ``out`` (a packed,
        pinned staging batch of those extents).

        Assembled by the library's host-side ``hb_collate`` (csrc/collate.hip: a molecule's rows are contiguous in the
        store, so a batch is a few memcpy's per molecule): ~25 us for a 256-molecule QM9 batch, against ~1.2 ms for the
        numpy gathers of ``collate_numpy`` (kept as the restatement the tests compare it with, bit for bit)."""
        from . import hip
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        B = idx.shape[0]
        if B and (int(idx.min()) < 0 or int(idx.max()) >= len(self)):
            raise IndexError("collate: molecule index outside the store")
        n_tot, h_tot, z_tot = self.extents(idx) if (pad_to is None or out is None) else (None, None, None)
        if pad_to is None:
            PN, PM, PZ, PB = n_tot, h_tot, z_tot, B
        else:
            PN, PM, PZ = (int(v) for v in pad_to)
            PB = B + 1
        if out is None:
            t = lambda shape, dt: torch.empty(shape, dtype=dt)
            out = HBatch(x=t((PN, 9), torch.int64), pos=t((PN, 3), torch.float32), edge_index0=t((PZ,), torch.int64),
                         edge_index1=t((PZ,), torch.int64), edge_attr=t((PM, 1), torch.int64), n_e=t((PB,), torch.int64),
                         e_order=t((PM,), torch.int64), batch=t((PN,), torch.int64), y=t((PB,), torch.float32))
        a = hip.HbCollate()
        a.B, a.n_mols, a.idx = B, len(self), idx.ctypes.data
        for name, ptr in self._checked_pointers().items():
            setattr(a, name, ptr)
        a.PN, a.PM, a.PZ, a.padded = PN, PM, PZ, 0 if pad_to is None else 1
        shapes = dict(x=(PN, 9), pos=(PN, 3), edge_index0=(PZ,), edge_index1=(PZ,), edge_attr=(PM, 1), n_e=(PB,), e_order=(PM,),
                      batch=(PN,), y=(PB,))
        for name, shp in shapes.items():
            ten = getattr(out, name)
            want = torch.float32 if name in ("pos", "y") else torch.int64
            if tuple(ten.shape) != shp or ten.dtype != want or not ten.is_contiguous() or ten.device.type != "cpu":
                raise ValueError(f"collate: out.{name} must be a contiguous CPU {want} tensor of shape {shp}")
            setattr(a, "out_" + name, ten.data_ptr())
        counts = np.zeros(3, dtype=np.int64)
        a.out_counts = counts.ctypes.data
        rc = hip.lib().hb_collate(ctypes.byref(a))
        if rc == -3 and pad_to is not None:       # EQH_ERR_RANGE: the extents do not fit (indices were checked above)
            raise ValueError("collate: pad_to must exceed the batch (nodes and hyperedges strictly)")
        hip.check(rc, "hb_collate")
        if pad_to is not None:
            out.num_real_graphs = B
        out.num_nodes, out.num_hyperedges, out.num_graphs = PN, PM, PB
        return out

    _ARRAYS = ("node_off", "he_off", "inc_off", "x", "pos", "v", "e", "edge_attr", "e_order", "y")

    def _checked_pointers(self) -> dict:
        """Addresses of the store's arrays for ``hb_collate``, which memcpy's whole rows by them: a store whose arrays have
        another dtype, width or length (a processed file with a different feature layout re-seated into a MolStore by hand)
        must fail HERE, not read the wrong rows or run past the end.  Checked once per set of array objects."""
        key = tuple(id(getattr(self, n)) for n in self._ARRAYS)
        cached = getattr(self, "_ptr_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        n_mols = len(self)
        for name in ("node_off", "he_off", "inc_off"):
            arr = getattr(self, name)
            if arr.dtype != np.int64 or arr.shape != (n_mols + 1,) or not arr.flags.c_contiguous:
                raise ValueError(f"MolStore.{name} must be a C-contiguous int64 array of {n_mols + 1} offsets")
        N, M, Z = int(self.node_off[-1]), int(self.he_off[-1]), int(self.inc_off[-1])
        want = dict(x=(np.int64, (N, 9)), pos=(np.float32, (N, 3)), v=(np.int64, (Z,)), e=(np.int64, (Z,)),
                    edge_attr=(np.int64, (M, 1)), e_order=(np.int64, (M,)), y=(np.float32, (n_mols,)))
        ptrs = {}
        for name in self._ARRAYS:
            arr = getattr(self, name)
            if name in want:
                dt, shape = want[name]
                if arr.dtype != dt or tuple(arr.shape) != shape or not arr.flags.c_contiguous:
                    raise ValueError(f"MolStore.{name} must be a C-contiguous {np.dtype(dt).name} array of shape {shape}, got "
                                     f"{arr.dtype} {tuple(arr.shape)}{'' if arr.flags.c_contiguous else ' (strided)'}")
            ptrs[name] = arr.ctypes.data if arr.size else None
        self._ptr_cache = (key, ptrs)
        return ptrs

    def collate_numpy(self, idx, pad_to: Optional[tuple] = None, out: Optional["HBatch"] = None) -> "HBatch":
        """``collate`` by numpy gathers driven by cumsum offsets (rounds 2-4's implementation; the restatement
        tests/test_fit.py compares the native assembly with)."""
        idx = np.asarray(idx, dtype=np.int64)
        B = idx.shape[0]
        n, h, z = self.n_nodes[idx], self.n_he[idx], self.n_inc[idx]
        src_n, own_n = self._ranges(self.node_off[idx], n)
        src_h, _ = self._ranges(self.he_off[idx], h)
        src_z, own_z = self._ranges(self.inc_off[idx], z)
        N, M, Z = src_n.shape[0], src_h.shape[0], src_z.shape[0]
        node_base = np.concatenate(([0], np.cumsum(n)[:-1]))
        he_base = np.concatenate(([0], np.cumsum(h)[:-1]))
        if pad_to is None:
            PN, PM, PZ, PB = N, M, Z, B
        else:
            PN, PM, PZ = (int(v) for v in pad_to)
            PB = B + 1
            if PN <= N or PM <= M or PZ < Z:
                raise ValueError("collate: pad_to must exceed the batch (nodes and hyperedges strictly)")
        if out is None:
            t = lambda shape, dt: torch.empty(shape, dtype=dt)
            out = HBatch(x=t((PN, 9), torch.int64), pos=t((PN, 3), torch.float32), edge_index0=t((PZ,), torch.int64),
                         edge_index1=t((PZ,), torch.int64), edge_attr=t((PM, 1), torch.int64), n_e=t((PB,), torch.int64),
                         e_order=t((PM,), torch.int64), batch=t((PN,), torch.int64), y=t((PB,), torch.float32))
        a = lambda name: getattr(out, name).numpy()
        a("x")[:N] = self.x[src_n]
        a("pos")[:N] = self.pos[src_n]
        a("batch")[:N] = own_n
        a("edge_index0")[:Z] = self.v[src_z] + node_base[own_z]
        a("edge_index1")[:Z] = self.e[src_z] + he_base[own_z]
        a("edge_attr")[:M] = self.edge_attr[src_h]
        a("e_order")[:M] = self.e_order[src_h]
        a("n_e")[:B] = h
        a("y")[:B] = self.y[idx]
        if pad_to is not None:
            pn = PN - N
            a("x")[N:] = 0
            far = a("pos")[N:]
            far[:] = 0.0
            far[:, 0] = 1.0e4 + 10.0 * np.arange(pn, dtype=np.float32)      # as pad_batch: 10 A apart, 10^4 A away
            a("batch")[N:] = B
            a("edge_index0")[Z:] = -1
            a("edge_index1")[Z:] = -1
            a("edge_attr")[M:] = 0
            a("e_order")[M:] = 0
            a("n_e")[B] = PM - M
            a("y")[B] = 0.0
            out.num_real_graphs = B
        out.num_nodes, out.num_hyperedges, out.num_graphs = PN, PM, PB
        return out


def bucket_sizes(n_nodes: int, n_hyperedges: int, n_inc: int, quantum: int = 128):
    """Static shapes for hipGraph replay: round each extent up to a multiple of ``quantum`` with
    at least one spare slot (the padding molecule needs a node and a hyperedge of its own).  The quantum
    trades padded rows (every kernel and GEMM of the step works on them: 128 is <= 2.8 % of a 256-molecule
    QM9 batch) against the number of distinct shape buckets, each of which is captured once (~0.3 s)."""
    up = lambda v: -(-(v + 1) // quantum) * quantum
    return up(n_nodes), up(n_hyperedges), up(n_inc)


def pad_batch(b: HBatch, n_nodes: int, n_hyperedges: int, n_inc: int) -> HBatch:
    """Pad a batch to fixed extents with ONE extra dummy molecule (graph id B) that owns every
    padded node and hyperedge.  Padded atoms sit 10 A apart on a line 10^4 A away, so no real atom
    ever selects one as a neighbour (each real atom has >= 16 real candidates) and the 5 A radius mask
    drops them; padded rows never mix with real rows in any aggregation, and the loss is taken over the
    first B outputs only — real-molecule outputs and all parameter gradients are unchanged for models
    without batch statistics (LayerNorm models: egnn_equihnns, equiformer_equihnns).

    Padded INCIDENCES are null: both coordinates are -1, which the CSR builders drop (out-of-range keys),
    so no aggregation ever visits them and the padded nodes / hyperedges have no incidences at all.
    (Spreading them over the few padded rows instead gave those rows 20-40 incidences each against 2-3
    for real rows, and the wavefront that owned them ran 10x longer than the rest of the launch: 112 us
    for a backward kernel that needs 10.)  `HyperIndex` keeps them at -1 in its int32 gather copies, which the row-gather
    kernel reads as zero rows (zero gradient for null incidences on the unfused per-incidence path too)."""
    N, M, nnz, B = b.x.shape[0], b.edge_attr.shape[0], b.edge_index0.shape[0], b.y.shape[0]
    if n_nodes <= N or n_hyperedges <= M or n_inc < nnz:
        raise ValueError("pad_batch: target extents must exceed the batch (nodes and hyperedges strictly)")
    dev = b.x.device
    pn, pm, pz = n_nodes - N, n_hyperedges - M, n_inc - nnz
    far = torch.zeros((pn, 3), dtype=b.pos.dtype, device=dev)
    far[:, 0] = 1.0e4 + 10.0 * torch.arange(pn, device=dev, dtype=b.pos.dtype)
    cat = torch.cat
    zl = lambda n, like: torch.zeros((n, *like.shape[1:]), dtype=like.dtype, device=dev)
    iv = torch.full((pz,), -1, device=dev)
    ie = torch.full((pz,), -1, device=dev)
    out = HBatch(
        x=cat((b.x, zl(pn, b.x))), pos=cat((b.pos, far)),
        edge_index0=cat((b.edge_index0, iv.to(b.edge_index0.dtype))),
        edge_index1=cat((b.edge_index1, ie.to(b.edge_index1.dtype))),
        edge_attr=cat((b.edge_attr, zl(pm, b.edge_attr))),
        n_e=cat((b.n_e, torch.tensor([pm], dtype=b.n_e.dtype, device=dev))),
        e_order=cat((b.e_order, zl(pm, b.e_order))),
        batch=cat((b.batch, torch.full((pn,), B, dtype=b.batch.dtype, device=dev))),
        y=cat((b.y, zl(1, b.y))),
        num_nodes=n_nodes, num_hyperedges=n_hyperedges, num_graphs=B + 1)
    out.num_real_graphs = B
    return out


def synth_molecule(rng: np.random.Generator, flavour: str = "qm9",
                   n_atoms: Optional[int] = None, force_conj: Optional[bool] = None) -> HMol:
    mu, sigma, n_max, p_conj = FLAVOURS[flavour]
    if n_atoms is None:
        n = int(np.clip(np.rint(rng.normal(mu, sigma)), 3, n_max))
    else:
        n = int(n_atoms)
    x = np.stack([rng.integers(0, d, size=n) for d in ATOM_FEATURE_DIMS], axis=1).astype(np.int64)

    # bonds: random spanning tree + a few ring-closing bonds
    bonds = []
    parent = np.zeros(n, dtype=np.int64)
    for a in range(1, n):
        p = int(rng.integers(0, a))
        parent[a] = p
        bonds.append((p, a))
    have = set(bonds)
    for _ in range(int(np.rint(0.08 * n))):
        a, b = sorted(int(t) for t in rng.choice(n, size=2, replace=False))
        if (a, b) not in have:
            have.add((a, b))
            bonds.append((a, b))
    nb = len(bonds)
    v = np.empty(2 * nb, dtype=np.int64)
    e = np.empty(2 * nb, dtype=np.int64)
    for i, (a, b) in enumerate(bonds):
        v[2 * i], v[2 * i + 1] = a, b
        e[2 * i] = e[2 * i + 1] = i
    edge_attr = rng.integers(0, 4, size=(nb, 1)).astype(np.int64)
    e_order = np.full((nb,), 2, dtype=np.int64)

    conj = (rng.random() < p_conj) if force_conj is None else force_conj
    if conj and n >= 3:
        k = int(rng.integers(3, min(8, n) + 1))
        v = np.concatenate([v, np.arange(k, dtype=np.int64)])
        e = np.concatenate([e, np.full((k,), nb, dtype=np.int64)])
        edge_attr = np.concatenate([edge_attr, np.full((1, 1), 5, dtype=np.int64)], 0)
        e_order = np.concatenate([e_order, np.array([k], dtype=np.int64)])

    # coordinates: chain growth along the spanning tree, then centred
    pos = np.zeros((n, 3), dtype=np.float64)
    for a in range(1, n):
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        # bond length 1.4 A +- 0.1: a constant length would make the squared distances of all
        # bonded pairs EXACTLY equal in fp32, and torch.topk's choice among exact ties at the
        # k-th neighbour is implementation-defined (see tests: test_knn_ties_prefer_lower_index)
        pos[a] = pos[parent[a]] + (1.4 + 0.1 * rng.uniform(-1.0, 1.0)) * u
    pos -= pos.mean(0, keepdims=True)
    return HMol(x=x, pos=pos.astype(np.float32), edge_index0=v, edge_index1=e,
                edge_attr=edge_attr, e_order=e_order, y=float(rng.normal()))


def synth_batch(batch_size: int, seed: int, flavour: str = "qm9") -> HBatch:
    """Seeded synthetic batch.  Pure host code, deterministic per seed."""
    rng = np.random.default_rng(seed)
    return collate([synth_molecule(rng, flavour) for _ in range(batch_size)])


def shard_indices(n_items: int, rank: int, world_size: int, seed: int, epoch: int = 0,
                  shuffle: bool = True) -> List[int]:
    """DistributedSampler semantics (the Lightning default the reference relies on,
    main.py:271-283): a shared shuffled permutation, padded to a multiple of the world
    size, strided by rank."""
    return shard_permutation(n_items, world_size, seed, epoch, shuffle)[rank::world_size].tolist()


def shard_permutation(n_items: int, world_size: int, seed: int, epoch: int = 0, shuffle: bool = True) -> np.ndarray:
    """The epoch's shared permutation, padded to a multiple of the world size: rank r's shard is ``[r::world_size]``
    (every rank can therefore see what every other rank will draw -- ``fit.BucketedLoader.plan``)."""
    perm = np.random.default_rng(seed + epoch).permutation(n_items) if shuffle else np.arange(n_items)
    total = -(-n_items // world_size) * world_size
    reps = -(-total // max(n_items, 1))
    return np.concatenate([perm] * reps)[:total] if total > n_items else perm
