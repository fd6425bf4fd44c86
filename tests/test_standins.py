"""Property tests of the stand-ins for the reference's absent third-party packages
(tests/golden/make_golden.py: torch_scatter.scatter, torch_geometric global_add_pool / to_dense_batch, ogb AtomEncoder,
einx.get_at).  The reference's own arithmetic is what the golden vectors pin; these stand-ins only have to honour the
documented semantics of the packages they replace, which is what is checked here, on CPU, without
importing the reference."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from common import GOLDEN_DIR

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN_DIR, "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)


def _rand(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32))


@pytest.mark.parametrize("lead", [(), (1,)])
@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_scatter_against_a_python_loop(lead, reduce):
    g = np.random.default_rng(0)
    nnz, rows, c = 57, 13, 5
    idx = torch.from_numpy(g.integers(0, rows - 2, nnz))          # rows 11, 12 stay empty
    src = _rand((*lead, nnz, c), 1)
    out = mg._standin_scatter(src, idx, dim=-2, dim_size=rows, reduce=reduce)
    want = torch.zeros(*lead, rows, c, dtype=torch.float64)
    cnt = torch.zeros(rows)
    for p in range(nnz):
        want[..., idx[p], :] += src[..., p, :].double()
        cnt[idx[p]] += 1
    if reduce == "mean":
        want = want / cnt.clamp(min=1)[:, None]
    np.testing.assert_allclose(out.numpy(), want.numpy(), atol=1e-6)
    assert float(out[..., 11:, :].abs().max()) == 0.0                                   # empty rows are zero


def test_scatter_properties():
    g = np.random.default_rng(2)
    idx = torch.from_numpy(g.integers(0, 9, 40))
    src = _rand((40, 4), 3)
    s = mg._standin_scatter(src, idx, dim=-2, reduce="sum")
    assert s.shape[0] == int(idx.max()) + 1                                             # dim_size=None -> max+1
    np.testing.assert_allclose(s.sum(0).numpy(), src.sum(0).numpy(), atol=1e-5)         # sum of scatter = sum of src
    const = torch.full((40, 4), 2.5)
    m = mg._standin_scatter(const, idx, dim=-2, dim_size=12, reduce="mean")
    present = torch.bincount(idx, minlength=12) > 0
    assert torch.all(m[present] == 2.5) and torch.all(m[~present] == 0)                 # mean of a constant
    perm = torch.from_numpy(g.permutation(40))                                          # permutation of the incidences
    np.testing.assert_allclose(mg._standin_scatter(src[perm], idx[perm], dim=-2, dim_size=12, reduce="mean").numpy(),
                               mg._standin_scatter(src, idx, dim=-2, dim_size=12, reduce="mean").numpy(), atol=1e-6)
    with pytest.raises(ValueError):
        mg._standin_scatter(src, idx, dim=-2, reduce="max")
    # differentiable in src: the mean's gradient is 1/count of the row
    src.requires_grad_(True)
    mg._standin_scatter(src, idx, dim=-2, dim_size=12, reduce="mean").sum().backward()
    cnt = torch.bincount(idx, minlength=12).float()
    np.testing.assert_allclose(src.grad[:, 0].numpy(), (1.0 / cnt[idx]).numpy(), rtol=1e-6)


def test_global_add_pool_reduces_dim_minus_2():
    batch = torch.tensor([0, 0, 1, 1, 1, 3])
    x = _rand((1, 6, 3), 4)
    out = mg._standin_global_add_pool(x, batch)
    assert out.shape == (1, 4, 3)
    np.testing.assert_allclose(out[0, 1].numpy(), x[0, 2:5].sum(0).numpy(), atol=1e-6)
    assert float(out[0, 2].abs().max()) == 0.0
    assert mg._standin_global_add_pool(x[0], batch, size=5).shape == (5, 3)


def test_atom_encoder_is_the_sum_of_nine_lookups():
    from equihgnn_amd.batch import ATOM_FEATURE_DIMS
    torch.manual_seed(0)
    enc = mg._StandinAtomEncoder(8)
    assert [e.weight.shape[0] for e in enc.atom_embedding_list] == list(ATOM_FEATURE_DIMS)
    g = np.random.default_rng(5)
    x = torch.from_numpy(np.stack([g.integers(0, d, 7) for d in ATOM_FEATURE_DIMS], 1))
    want = sum(enc.atom_embedding_list[f].weight[x[:, f]] for f in range(9))
    np.testing.assert_allclose(enc(x).detach().numpy(), want.detach().numpy(), atol=1e-6)
    for e in enc.atom_embedding_list:                 # xavier-uniform bound sqrt(6 / (rows + dim))
        assert float(e.weight.abs().max()) <= (6.0 / (e.weight.shape[0] + 8)) ** 0.5 + 1e-6


def test_get_at_patterns():
    t = _rand((1, 6, 4, 3), 6)
    idx = torch.tensor([[[1, 2], [0, 5], [3, 3], [2, 1], [0, 0], [4, 5]]])
    got = mg._standin_get_at("b [i] d m, b j k -> b j k d m", t, idx)
    assert got.shape == (1, 6, 2, 4, 3)
    for j in range(6):
        for k in range(2):
            assert torch.equal(got[0, j, k], t[0, idx[0, j, k]])
    pair = _rand((1, 6, 5, 3), 7)
    sel = torch.tensor([[[0, 4], [1, 1], [2, 3], [4, 0], [3, 3], [0, 1]]])
    got = mg._standin_get_at("b i [j] c, b i k -> b i k c", pair, sel)
    for i in range(6):
        for k in range(2):
            assert torch.equal(got[0, i, k], pair[0, i, sel[0, i, k]])
    got = mg._standin_get_at("b i [j], b i k -> b i k", pair[..., 0], sel)
    assert torch.equal(got, pair[..., 0].gather(2, sel))
    with pytest.raises(NotImplementedError):
        mg._standin_get_at("b [i], b j -> b j", t, idx)


def test_to_dense_batch():
    batch = torch.tensor([0, 0, 0, 1, 2, 2])
    x = _rand((6, 2), 8)
    dense, mask = mg._standin_to_dense_batch(x, batch)
    assert dense.shape == (3, 3, 2) and mask.sum() == 6
    assert torch.equal(dense[mask], x)                                # row-major order of the valid slots = input order
    assert float(dense[~mask].abs().max()) == 0.0
    d1, m1 = mg._standin_to_dense_batch(x, None, max_num_nodes=6)     # FAFormer's call: one "batch" of N tokens
    assert d1.shape == (1, 6, 2) and bool(m1.all())


def test_reconstructed_j_matrices():
    """J_l must be an orthogonal involution (it is the representation of an axis swap), SURVEY.md §8c."""
    for j in mg.reconstructed_J():
        eye = torch.eye(j.shape[0], dtype=j.dtype)
        assert torch.allclose(j @ j, eye, atol=1e-12) and torch.allclose(j @ j.T, eye, atol=1e-12)
