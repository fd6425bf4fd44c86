"""The LitModel-equivalent harness (equihgnn_amd/fit.py) on CPU with the oracle model: loss goes
down, the LR schedule / early stopping / best-checkpoint bookkeeping follow main.py:137-151,259-267,
the target normalisation reproduces utils/data_split.py:67-72, bootstrapped metrics behave."""
import numpy as np
import torch

import oracle  # noqa: F401
from equihgnn_amd.batch import synth_molecule
from equihgnn_amd.fit import (BootstrapMetrics, Fitter, MolLoader, normalize_targets_like_reference,
                              split_80_10_10)
from equihgnn_amd.registry import default_args
from oracle import ref_models as O


def _mols(n, seed):
    rng = np.random.default_rng(seed)
    mols = [synth_molecule(rng) for _ in range(n)]
    for m in mols:  # a learnable target: number of atoms (standardised below)
        m.y = float(m.x.shape[0])
    return mols


def test_split_and_reference_normalisation():
    tr, va, te = split_80_10_10(103, seed=0)
    assert (len(tr), len(va), len(te)) == (82, 10, 11) and sorted(tr + va + te) == list(range(103))
    y = torch.tensor([1.0, 2.0, 4.0, 9.0])
    out, std = normalize_targets_like_reference(y)
    m, s = y.mean(), y.std()
    ref = (((y - m) / s - m) / s - m) / s          # three passes over the SHARED dataset
    assert torch.allclose(out, ref) and abs(std - float(s)) < 1e-7


def test_bootstrap_metrics():
    bm = BootstrapMetrics(50, seed=1)
    g = torch.Generator().manual_seed(0)
    p, t = torch.randn(4000, generator=g), torch.randn(4000, generator=g)
    bm.update(p[:2000], t[:2000])
    bm.update(p[2000:], t[2000:])
    out = bm.compute()
    mae, mse = float((p - t).abs().mean()), float(((p - t) ** 2).mean())
    assert abs(out["mae_mean"] - mae) < 0.03 and abs(out["mse_mean"] - mse) < 0.08
    assert 0 < out["mae_std"] < 0.05 and 0 < out["mse_std"] < 0.1


def test_fit_loop_learns_and_keeps_best_checkpoint():
    torch.manual_seed(0)
    mols = _mols(60, 5)
    y, std = normalize_targets_like_reference(torch.tensor([m.y for m in mols]))
    for m, v in zip(mols, y.tolist()):
        m.y = v
    tr, va, te = split_80_10_10(len(mols), seed=1)
    pick = lambda ids: [mols[i] for i in ids]
    args = default_args(method="mhnns", MLP_hidden=32, output_hidden=16)
    model = O.MODELS["mhnns"](1, args)
    fitter = Fitter(model, lr=3e-3, std=std, patience_lr=1, patience_stop=3)
    res = fitter.fit(MolLoader(pick(tr), 16, True, seed=0), MolLoader(pick(va), 16, False), epochs=8)
    h = res.history
    assert h[-1]["train_loss"] < h[0]["train_loss"]
    assert res.best_epoch >= 0 and res.best_state is not None
    assert abs(res.best_val_mae - min(e["val_mae_mean"] for e in h)) < 1e-12
    assert all(e["lr"] <= 3e-3 + 1e-12 for e in h) and h[-1]["lr"] >= 3e-3 * 1e-5 - 1e-15
    metrics, table = fitter.test(MolLoader(pick(te), 16, False), res.best_state)
    assert table.shape == (len(te), 2) and np.isfinite(table).all()
    assert set(metrics) == {"test_mae_mean", "test_mae_std", "test_mse_mean", "test_mse_std"}


def test_early_stopping_and_plateau_schedule():
    """A frozen model cannot improve except through the bootstrap resampling noise of the monitored
    metric (which the reference has too): training stops once `patience_stop` consecutive epochs
    fail to beat the best value (EarlyStopping semantics)."""
    mols = _mols(24, 2)
    args = default_args(method="mhnns", MLP_hidden=16, output_hidden=8)
    model = O.MODELS["mhnns"](1, args)
    fitter = Fitter(model, lr=0.0, patience_lr=1, patience_stop=4)
    res = fitter.fit(MolLoader(mols[:16], 8, False), MolLoader(mols[16:], 8, False), epochs=50)
    assert res.stopped_early and len(res.history) < 50
    assert len(res.history) - 1 - res.best_epoch == 4   # exactly `patience_stop` bad epochs after the best


def test_packed_batch_is_a_view_of_one_buffer_and_survives_to():
    """HBatch.packed(): same values, every tensor field inside one flat buffer; .to() keeps the packing
    (one transfer) and the extra attributes."""
    from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
    raw = synth_batch(6, 11, "qm9")
    b = pad_batch(raw, *bucket_sizes(raw.num_nodes, raw.num_hyperedges, raw.nnz))
    p = b.packed()
    lo, hi = p._flat.data_ptr(), p._flat.data_ptr() + p._flat.numel()
    for name in ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "n_e", "e_order", "batch", "y"):
        t, u = getattr(b, name), getattr(p, name)
        assert torch.equal(t, u) and t.dtype == u.dtype
        assert lo <= u.data_ptr() < hi and (u.data_ptr() - lo) % 256 == 0
    q = p.to("cpu")
    assert q._layout == p._layout and torch.equal(q.pos, b.pos) and q.num_real_graphs == b.num_real_graphs
    assert (q.num_nodes, q.num_hyperedges, q.num_graphs) == (b.num_nodes, b.num_hyperedges, b.num_graphs)
