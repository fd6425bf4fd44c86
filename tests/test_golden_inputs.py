"""The committed golden files against the case table (tests/golden/common.py): every file's INPUTS are a pure
function of its row -- this repo's seeded synthetic generator, no reference involved -- so they are re-derived here
and compared bit for bit.  (That the stored OUTPUTS are what the reference computes for those inputs is checked in
the build container by `python tests/golden/make_golden.py --check`, which re-runs the reference on every case and
compares all arrays; the GPU box has no reference.)"""
import os

import numpy as np
import pytest

from common import CASE_TABLE, F64_TABLE, GOLDEN_DIR, case_spec, f64_spec, load_case, make_batch

FIELDS = ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "n_e", "e_order", "batch", "y")


def test_every_case_has_a_file_and_every_file_a_case():
    files = {f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz")}
    extra = {"equiformer_D"} | {f for f in files if f.startswith(("trajectory_", "equiformer_layer_"))}   # not model cases
    assert files - extra == set(CASE_TABLE) | set(F64_TABLE)


@pytest.mark.parametrize("name", list(CASE_TABLE) + list(F64_TABLE))
def test_inputs_rederive_bit_for_bit(name):
    case, spec = load_case(name), (f64_spec(name) if name in F64_TABLE else case_spec(name))
    b = make_batch(spec)
    for k in FIELDS:
        got = getattr(b, k).numpy()
        assert got.dtype == case["in_" + k].dtype and got.shape == case["in_" + k].shape, k
        assert np.array_equal(got, case["in_" + k]), k
    assert str(case["meta_method"]) == spec["method"] and int(case["meta_hidden"]) == spec["hidden"]
    assert int(case["meta_seed"]) == spec["seed"] and bool(int(case["meta_train"])) == spec["train"]


def test_workload_flavours_are_covered():
    """BASELINE configs 4 / 5 run on PCQM4Mv2- / Molecule3D-like molecules: both methods have fixtures on that
    flavour, with a molecule of more than 40 atoms."""
    for method in ("egnn_equihnns", "faformer_equihnns"):
        names = [n for n in CASE_TABLE if case_spec(n)["method"] == method and case_spec(n)["flavour"] == "pcqm"]
        assert names, method
        for n in names:
            counts = np.bincount(load_case(n)["in_batch"])
            assert counts.max() >= 40, (n, counts.max())


def test_degenerate_case_contains_the_special_edges():
    case = load_case("equiformer_equihnns_c64_degenerate")
    pos = case["in_pos"].astype(np.float64)
    rel = pos[:, None, :] - pos[None, :, :]
    d = np.linalg.norm(rel, axis=-1)
    iu = np.triu_indices(len(pos), 1)
    assert (d[iu] == 0).sum() >= 1                                        # coincident atoms
    with np.errstate(invalid="ignore", divide="ignore"):
        xhat = rel / d[..., None]
    s = ((xhat + np.array([0.0, 1.0, 0.0])) ** 2).sum(-1)
    close = d < 2.0
    assert ((s == 0) & close).sum() >= 1                                  # an edge along exactly -y
    assert ((s > 0) & (s < 1e-6) & close).sum() >= 3                      # inside the clamp of basis.py:187-190
