"""Reader for the reference's processed InMemoryDataset files (equihgnn_amd/reader.py): a file written
here with stand-ins that carry torch_geometric's / the reference's class names (neither is installed) must load without
them, and batches collated from it must equal batches collated from the original molecules."""
import sys
import types

import numpy as np
import torch

from equihgnn_amd.batch import MolStore, synth_molecule
from equihgnn_amd.reader import load_processed, read_processed, store_from_slices


def _write_processed(path, mols, with_e_order=True):
    """torch.save((data, slices)) as InMemoryDataset.collate lays it out: concatenated fields with LOCAL ids, offsets per
    field; the containers are instances of classes named like PyG's (Data -> _store: GlobalStorage -> _mapping)."""
    mods = {}
    for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.storage", "equihgnn", "equihgnn.data",
                 "equihgnn.data.utils"):
        mods[name] = types.ModuleType(name)

    class GlobalStorage:
        pass

    class HData:
        pass

    GlobalStorage.__module__, GlobalStorage.__qualname__ = "torch_geometric.data.storage", "GlobalStorage"
    HData.__module__, HData.__qualname__ = "equihgnn.data.utils", "HData"
    mods["torch_geometric.data.storage"].GlobalStorage = GlobalStorage
    mods["equihgnn.data.utils"].HData = HData
    cat = lambda xs: torch.from_numpy(np.concatenate(xs, 0))
    off = lambda c: torch.from_numpy(np.concatenate(([0], np.cumsum(c))).astype(np.int64))
    fields = {"x": cat([m.x for m in mols]), "pos": cat([m.pos for m in mols]),
              "edge_index0": cat([m.edge_index0 for m in mols]), "edge_index1": cat([m.edge_index1 for m in mols]),
              "edge_attr": cat([m.edge_attr for m in mols]),
              "y": torch.tensor([[0.5, m.y, -1.0] for m in mols], dtype=torch.float32),      # three targets; column 1 is ours
              "n_e": torch.tensor([m.edge_attr.shape[0] for m in mols]), "idx": torch.arange(len(mols))}
    if with_e_order:
        fields["e_order"] = cat([m.e_order for m in mols])
    n = [m.x.shape[0] for m in mols]
    z = [m.edge_index0.shape[0] for m in mols]
    h = [m.edge_attr.shape[0] for m in mols]
    one = off(np.ones(len(mols), dtype=np.int64))
    slices = {"x": off(n), "pos": off(n), "edge_index0": off(z), "edge_index1": off(z), "edge_attr": off(h),
              "e_order": off(h), "y": one, "n_e": one, "idx": one}
    store, data = GlobalStorage(), HData()
    store.__dict__["_mapping"] = fields
    data.__dict__["_store"] = store
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        torch.save((data, slices), path)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_processed_file_reads_without_torch_geometric(tmp_path):
    rng = np.random.default_rng(5)
    mols = [synth_molecule(rng, "qm9") for _ in range(23)]
    path = tmp_path / "3dhg_data.pt"
    _write_processed(path, mols)
    assert "torch_geometric" not in sys.modules and "equihgnn.data.utils" not in sys.modules
    got = read_processed(path, target=1)
    ref = MolStore(mols)
    assert len(got) == 23
    idx = np.array([3, 0, 22, 7, 7, 15])
    a, b = got.collate(idx), ref.collate(idx)
    for f in ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "n_e", "e_order", "batch", "y"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    pa, pb = got.collate(idx, pad_to=(256, 256, 512)), ref.collate(idx, pad_to=(256, 256, 512))
    assert torch.equal(pa.edge_index0, pb.edge_index0) and torch.equal(pa.batch, pb.batch)


def test_e_order_is_recomputed_and_inconsistent_files_are_refused(tmp_path):
    rng = np.random.default_rng(6)
    mols = [synth_molecule(rng, "pcqm") for _ in range(9)]
    path = tmp_path / "data.pt"
    _write_processed(path, mols, with_e_order=False)
    fields, slices = load_processed(path)
    assert "e_order" not in fields
    st = store_from_slices(fields, slices, target=1)
    assert np.array_equal(st.e_order, MolStore(mols).e_order)
    bad = dict(fields)
    bad["edge_index0"] = fields["edge_index0"] + 100          # not local to the molecule any more
    try:
        store_from_slices(bad, slices)
    except ValueError as e:
        assert "local" in str(e)
    else:
        raise AssertionError("a global-id file was accepted")
