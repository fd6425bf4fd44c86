"""Reader for the reference's processed dataset files.

The reference's datasets (data/qm9.py:133,171-173; pcqm4.py, molecule3d.py, opv3d.py alike) are torch_geometric
``InMemoryDataset``s: ``processed/<name>.pt`` holds ``torch.save((data, slices))`` where ``data`` is ONE ``HData``
(data/utils.py:150-178) with the fields of all molecules concatenated (node / hyperedge ids LOCAL to their molecule:
``InMemoryDataset.collate`` does not increment) and ``slices`` maps every field to its per-molecule offsets.  That is
already a structure of arrays, i.e. what ``batch.MolStore`` is built from -- no per-molecule objects are needed.

torch_geometric is not a dependency of this package.  Its classes in the pickle stream (``HData``, ``Data``,
``GlobalStorage`` ...) are replaced by an inert stand-in while loading, and the tensor fields are then looked up by
name in whatever container the installed torch_geometric version nested them in."""
from __future__ import annotations

import pickle
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .batch import MolStore

FIELDS = ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "e_order", "y", "n_e")
_FOREIGN = ("torch_geometric", "equihgnn")


class _Bag:
    """Stand-in for any pickled object of a foreign class: keeps the constructor arguments and the state."""

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs, self.state = args, kwargs, None

    def __setstate__(self, state):
        self.__dict__.setdefault("args", ())
        self.__dict__.setdefault("kwargs", {})
        self.__dict__["state"] = state


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in _FOREIGN:
            return _Bag
        return super().find_class(module, name)


class _PickleModule:
    """What torch.load(pickle_module=...) needs: Unpickler (persistent_load is installed by torch) and load."""
    __name__ = "equihgnn_amd.reader"
    Unpickler = _Unpickler

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()


def _find_fields(obj, depth=0) -> Optional[Dict[str, torch.Tensor]]:
    """The first mapping (at any nesting depth of stand-ins / dicts / sequences) that holds the HData tensors."""
    if depth > 8:
        return None
    if isinstance(obj, dict):
        if "edge_index0" in obj and "x" in obj and torch.is_tensor(obj["x"]):
            return obj
        children = list(obj.values())
    elif isinstance(obj, _Bag):
        children = [obj.state, obj.args, obj.kwargs] + [v for k, v in obj.__dict__.items() if k not in ("state", "args", "kwargs")]
    elif isinstance(obj, (list, tuple)):
        children = list(obj)
    else:
        return None
    for c in children:
        hit = _find_fields(c, depth + 1)
        if hit is not None:
            return hit
    return None


def load_processed(path) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(fields, slices) of a reference ``processed/*.pt`` file: the concatenated tensors of all molecules and the
    per-molecule offsets of each."""
    obj = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    if not (isinstance(obj, (tuple, list)) and len(obj) >= 2):
        raise ValueError(f"{path}: expected the (data, slices) pair of an InMemoryDataset")
    fields = _find_fields(obj[0])
    slices = obj[1] if isinstance(obj[1], dict) else _find_fields(obj[1])
    if fields is None or not isinstance(slices, dict):
        raise ValueError(f"{path}: no HData fields (x, edge_index0, ...) found")
    missing = [k for k in ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "y") if k not in fields or k not in slices]
    if missing:
        raise ValueError(f"{path}: fields missing from the file: {missing}")
    return dict(fields), dict(slices)


def store_from_slices(fields: Dict[str, torch.Tensor], slices: Dict[str, torch.Tensor], target: int = 0) -> MolStore:
    """A MolStore over the molecules of (fields, slices); ``target`` selects the column of y (OneTarget, data/utils.py:181-189).
    ``e_order`` (hyperedge orders, data/utils.py:57-61) is recomputed from edge_index1 when the file has none."""
    npy = lambda t, dt: np.ascontiguousarray(t.detach().cpu().numpy()).astype(dt, copy=False)
    sl = lambda k: npy(slices[k], np.int64)
    st = MolStore.__new__(MolStore)
    st.node_off, st.inc_off, st.he_off = sl("x"), sl("edge_index0"), sl("edge_attr")
    st.n_nodes, st.n_inc, st.n_he = np.diff(st.node_off), np.diff(st.inc_off), np.diff(st.he_off)
    n_mol = st.n_nodes.shape[0]
    if not (np.array_equal(sl("pos"), st.node_off) and np.array_equal(sl("edge_index1"), st.inc_off)):
        raise ValueError("slices of pos / edge_index1 disagree with x / edge_index0")
    if fields["x"].dim() != 2 or fields["x"].shape[-1] != 9:
        raise ValueError(f"x must hold the 9 ogb atom features per node (data/utils.py:150-178), got shape {tuple(fields['x'].shape)}")
    st.x = npy(fields["x"], np.int64).reshape(-1, 9)
    st.pos = npy(fields["pos"], np.float32).reshape(-1, 3)
    st.v, st.e = npy(fields["edge_index0"], np.int64).reshape(-1), npy(fields["edge_index1"], np.int64).reshape(-1)
    st.edge_attr = np.ascontiguousarray(npy(fields["edge_attr"], np.int64).reshape(st.he_off[-1], -1)[:, :1])
    if "n_e" in fields and not np.array_equal(npy(fields["n_e"], np.int64).reshape(-1), st.n_he):
        raise ValueError("n_e disagrees with the number of edge_attr rows per molecule")
    if "e_order" in fields:
        st.e_order = npy(fields["e_order"], np.int64).reshape(-1)
    else:       # count of incidences per (molecule, local hyperedge)
        owner = np.repeat(np.arange(n_mol, dtype=np.int64), st.n_inc)
        st.e_order = np.bincount(st.he_off[owner] + st.e, minlength=int(st.he_off[-1])).astype(np.int64)
    y = npy(fields["y"], np.float32).reshape(n_mol, -1)
    st.y = np.ascontiguousarray(y[:, target])
    if st.x.shape[0] != st.node_off[-1] or st.v.shape[0] != st.inc_off[-1] or st.e_order.shape[0] != st.he_off[-1]:
        raise ValueError("field lengths disagree with their slices")
    if st.v.size and (int(st.v.max()) >= int(st.n_nodes.max()) or int(st.e.max()) >= int(st.n_he.max())):
        raise ValueError("edge_index0 / edge_index1 are not local to their molecules")
    return st


def read_processed(path, target: int = 0) -> MolStore:
    """MolStore of a reference ``processed/*.pt`` dataset file."""
    return store_from_slices(*load_processed(path), target=target)
