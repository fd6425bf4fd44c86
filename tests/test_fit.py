"""The LitModel-equivalent harness (equihgnn_amd/fit.py) on CPU with the oracle model: loss goes
down, the LR schedule / early stopping / best-checkpoint bookkeeping follow main.py:137-151,259-267,
the target normalisation reproduces utils/data_split.py:67-72, bootstrapped metrics behave."""
import numpy as np
import pytest
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


def test_vectorised_collate_equals_loop_collate():
    """MolStore.collate (array operations only) against the per-molecule collate and pad_batch, field by field,
    including a batch that contains the zero-hyperedge one-atom molecule."""
    from equihgnn_amd.batch import MolStore, bucket_sizes, collate, pad_batch
    mols = _mols(200, 9)
    lone = mols[17]
    lone.x, lone.pos = lone.x[:1], lone.pos[:1]
    lone.edge_index0, lone.edge_index1 = lone.edge_index0[:0], lone.edge_index1[:0]
    lone.edge_attr, lone.e_order = lone.edge_attr[:0], lone.e_order[:0]
    store = MolStore(mols)
    rng = np.random.default_rng(1)
    fields = ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "n_e", "e_order", "batch", "y")
    for trial in range(4):
        idx = rng.permutation(200)[: 33 + trial]
        if trial == 0:
            idx[5] = 17
        ref = collate([mols[i] for i in idx])
        got = store.collate(idx)
        for f in fields:
            assert torch.equal(getattr(ref, f), getattr(got, f)), f
        assert store.extents(idx) == (ref.num_nodes, ref.num_hyperedges, ref.nnz)
        tgt = bucket_sizes(ref.num_nodes, ref.num_hyperedges, ref.nnz, 64)
        refp, gotp = pad_batch(ref, *tgt), store.collate(idx, pad_to=tgt)
        for f in fields:
            assert torch.equal(getattr(refp, f), getattr(gotp, f)), f
        assert gotp.num_real_graphs == len(idx) and gotp.num_graphs == len(idx) + 1


def test_bucketed_loader_static_shapes_and_coverage():
    """A few static buckets per run (a ladder of `levels` shapes one quantum apart; exactly one with levels=1), every
    molecule exactly once per epoch, packed staging buffers, DistributedSampler sharding across two ranks."""
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    mols = _mols(230, 4)
    for i, m in enumerate(mols):
        m.y = float(i)
    store = MolStore(mols)
    seen = []
    for rank in range(2):
        ld = BucketedLoader(store, 32, True, seed=5, device=None, rank=rank, world=2, prefetch=2, levels=1 + 2 * rank)
        ld.lookahead = False                                    # (count exactly one epoch's collation below)
        shapes = set()
        for b in ld:
            nb = b.num_real_graphs
            seen.extend(int(v) for v in b.y[:nb].tolist())
            shapes.add((b.x.shape[0], b.edge_attr.shape[0], b.edge_index0.shape[0], b.y.shape[0]))
            assert getattr(b, "_flat", None) is not None and int(b.batch[-1]) == nb
        buckets = sorted({s[:3] for s in shapes})
        assert 1 <= len(buckets) <= ld.levels                 # rank 0: one bucket; rank 1: a ladder of up to three
        for lo, hi in zip(buckets, buckets[1:]):              # neighbouring ladder steps: (q, q, 2 q) apart
            assert (hi[0] - lo[0]) % ld.quantum == 0 and hi[2] - lo[2] == 2 * (hi[0] - lo[0]) and hi[1] - lo[1] == hi[0] - lo[0]
        assert ld.collated == 115 and ld.collate_seconds > 0
    assert sorted(seen) == list(range(230))


@pytest.mark.gpu
def test_fit_two_epochs_on_graphed_step():
    """fit() with the hipGraph-replayed step (GraphedTrainStep) fed by the bucketed prefetching loader: two epochs,
    a handful of captured graphs at most, the loss goes down, the scheduler's learning rate reaches the device, and
    evaluation on padded batches scores the real molecules only."""
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.trainer import GraphedTrainStep
    torch.manual_seed(0)
    mols = _mols(400, 6)
    y, std = normalize_targets_like_reference(torch.tensor([m.y for m in mols]))
    for m, v in zip(mols, y.tolist()):
        m.y = v
    tr, va, te = split_80_10_10(len(mols), seed=1)
    pick = lambda ids: MolStore([mols[i] for i in ids])
    args = default_args(method="egnn_equihnns", MLP_hidden=64, output_hidden=32)
    model = MODELS["egnn_equihnns"](1, args).to("cuda:0")
    fitter = Fitter(model, lr=2e-3, std=std, patience_lr=0, patience_stop=5, step_factory=GraphedTrainStep)
    train = BucketedLoader(pick(tr), 32, True, seed=0, device="cuda:0", levels=2)
    res = fitter.fit(train, BucketedLoader(pick(va), 32, False, device="cuda:0", levels=1), epochs=3)
    h = res.history
    assert len(h) == 3 and h[-1]["train_loss"] < h[0]["train_loss"] and np.isfinite(h[-1]["val_mae_mean"])
    assert len(fitter.step.slots) <= 4                          # a few static shapes (a two-step ladder), not one per batch
    assert train.collated >= 3 * 320                            # (+ what was collated ahead for a fourth epoch)
    fitter.step.opt.sync_lr()                                   # (the trainer does this before every replay)
    assert abs(float(fitter.step.opt.state[fitter.step.pflat]["lr"]) - fitter.step.opt.param_groups[0]["lr"]) < 1e-9
    assert fitter.step.opt.param_groups[0]["lr"] <= 2e-3
    metrics, table = fitter.test(BucketedLoader(pick(te), 32, False, device="cuda:0"), res.best_state)
    assert table.shape == (40, 2) and np.isfinite(metrics["test_mae_mean"])


def test_bucketed_loader_reused_buffers_do_not_carry_a_stale_index():
    """The ring hands the same HBatch objects out again (prefetch + 3 per shape, across epochs): whatever a consumer
    cached on a batch object -- HyperIndex.from_batch stores the CSRs as `_hyper_index` -- must be gone when the
    object comes back with other molecules in it."""
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    store = MolStore(_mols(16 * 12, 8))
    ld = BucketedLoader(store, 16, False, device=None, prefetch=2)
    seen = {}
    for epoch in range(2):
        for i, b in enumerate(ld):
            assert getattr(b, "_hyper_index", None) is None, (epoch, i)
            b._hyper_index = ("index of", epoch, i)          # what HyperIndex.from_batch would do
            seen[id(b)] = seen.get(id(b), 0) + 1
    assert max(seen.values()) > 1                            # buffers really were reused


def test_bucketed_loader_reraises_a_failure_of_the_prefetch_thread():
    """An exception in the collating thread must not look like a short epoch (ranks would run different step counts
    and the next all-reduce would hang): it is re-raised in the consumer."""
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    store = MolStore(_mols(64, 3))
    ld = BucketedLoader(store, 16, False, device=None, prefetch=2)
    real, calls = store.collate, []

    def flaky(idx, **kw):
        calls.append(len(idx))
        if len(calls) == 3:
            raise ValueError("boom")
        return real(idx, **kw)

    store.collate = flaky
    got = 0
    with pytest.raises(RuntimeError, match="prefetch thread failed") as info:
        for _ in ld:
            got += 1
    assert got == 2 and isinstance(info.value.__cause__, ValueError)


def test_bucketed_loader_releases_its_thread_when_the_consumer_leaves_early():
    import threading
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    store = MolStore(_mols(16 * 10, 3))
    ld = BucketedLoader(store, 16, False, device=None, prefetch=1)
    before = threading.active_count()
    for i, _ in enumerate(ld):
        if i == 1:
            break                                             # generator closed: its finally-block stops the producer
    assert threading.active_count() == before
    assert sum(1 for _ in ld) == 10                           # and the next epoch is complete


def test_bucketed_loader_runs_one_epoch_ahead_and_no_further():
    """The prefetch thread starts the NEXT epoch when it has queued the current one; an epoch that fits in the queue
    whole must not go on to start the one after it.  close() joins whatever was started ahead."""
    import threading
    import time
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    store = MolStore(_mols(32, 3))
    before = threading.active_count()
    ld = BucketedLoader(store, 16, True, seed=1, device=None, prefetch=3)        # 2 batches per epoch < queue depth
    assert sum(1 for _ in ld) == 2
    time.sleep(0.4)
    assert 32 <= ld.collated <= 64                           # this epoch + at most the one started ahead
    assert sum(b.num_real_graphs for b in ld) == 32          # which is the epoch the next pass consumes
    ld.close()
    assert threading.active_count() == before


def test_bucketed_loader_lookahead_does_not_refill_buffers_that_are_still_queued():
    """The epoch started ahead shares the queue of the one being consumed: with a queue of its own it would have
    refilled ring buffers (prefetch + 3 per shape) whose batches still waited, unconsumed, in the old queue -- the tail
    of every epoch silently replaced by molecules of the next one."""
    import time
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    mols = _mols(16 * 9, 3)                                   # 9 equal-shape batches per epoch > 6 ring buffers
    for i, m in enumerate(mols):
        m.y = float(i)
    ld = BucketedLoader(MolStore(mols), 16, True, seed=3, device=None, prefetch=3)
    for _ in range(3):
        seen = []
        for b in ld:
            time.sleep(0.02)                                  # a slow consumer: the producer runs as far ahead as it may
            seen.extend(int(v) for v in b.y[:b.num_real_graphs].tolist())
        assert sorted(seen) == list(range(len(mols)))
    ld.close()


@pytest.mark.gpu
def test_evaluation_on_reused_loader_buffers_matches_fresh_batches():
    """Fitter._evaluate / test on a BucketedLoader with MORE equal-shape batches than the ring holds,
    over two passes -- predictions equal those of freshly collated, unpadded batches of the same molecules."""
    from equihgnn_amd.batch import MolStore, collate
    from equihgnn_amd.fit import BucketedLoader
    from equihgnn_amd.models import MODELS
    torch.manual_seed(0)
    mols = _mols(16 * 9, 12)                                  # 9 equal-size batches > prefetch + 3 = 6 buffers
    args = default_args(method="egnn_equihnns", MLP_hidden=64, output_hidden=32)
    model = MODELS["egnn_equihnns"](1, args).to("cuda:0")
    for m in model.modules():                                  # (the 1e-3-initialised EGNN branch made live)
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.normal_(m.weight, std=m.in_features ** -0.5)
    model.eval()
    with torch.no_grad():
        want = torch.cat([model(collate(mols[i:i + 16]).to("cuda:0")) for i in range(0, len(mols), 16)]).cpu().numpy()
    fitter = Fitter(model, lr=0.0)
    ld = BucketedLoader(MolStore(mols), 16, False, device="cuda:0")
    for _ in range(2):
        _, table = fitter.test(ld)
        np.testing.assert_allclose(table[:, 0], want, rtol=1e-5, atol=1e-5)
        assert float(np.abs(want).std()) > 1e-3               # (predictions differ between molecules: the check bites)


def test_discarded_lookahead_epoch_does_not_skip_a_permutation():
    """an epoch that was planned AHEAD and then discarded (close() at the end of fit() / test(), a break) must
    not advance the shuffle epoch: the sequence of permutations a loader yields is the same with and without lookahead
    (DistributedSampler.set_epoch semantics)."""
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    mols = _mols(16 * 5, 6)
    for i, m in enumerate(mols):
        m.y = float(i)

    def epochs(lookahead, close_between):
        ld = BucketedLoader(MolStore(mols), 16, True, seed=11, device=None, prefetch=2)
        ld.lookahead = lookahead
        out = []
        for _ in range(3):
            out.append([int(v) for b in ld for v in b.y[:b.num_real_graphs].tolist()])
            if close_between:
                ld.close()                      # what Fitter.fit() / test() do when they are done with a loader
        ld.close()
        return out

    want = epochs(False, False)
    assert want[0] != want[1] != want[2]
    assert epochs(True, True) == want
    assert epochs(True, False) == want


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["egnn_equihnns", "mhnnm"])
def test_graphed_evaluation_matches_eager_evaluation(method):
    """Fitter._evaluate / test on padded bucketed batches through trainer.GraphedEvalStep (a forward-only hipGraph per shape
    bucket, main.py:65-87,89-132) give the predictions of the eager forward pass, before and after further training steps
    (the graphs read the parameters -- and BatchNorm's running statistics -- in place)."""
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.trainer import GraphedTrainStep
    torch.manual_seed(0)
    mols = _mols(16 * 10, 21)
    args = default_args(method=method, MLP_hidden=64, output_hidden=32)
    model = MODELS[method](1, args).to("cuda:0")
    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.normal_(m.weight, std=m.in_features ** -0.5)
    fitter = Fitter(model, lr=1e-3, step_factory=GraphedTrainStep)
    store = MolStore(mols)
    train = BucketedLoader(store, 16, True, seed=0, device="cuda:0")
    evald = BucketedLoader(store, 16, False, device="cuda:0")
    for round_ in range(2):
        for b in train:                                           # a few optimiser steps move parameters and buffers
            fitter.step.step(b)
        fitter.graph_eval = True
        m_graph, t_graph = fitter.test(evald)
        assert fitter.eval_step is not None and 1 <= len(fitter.eval_step.slots) <= 2
        fitter.graph_eval = False
        m_eager, t_eager = fitter.test(evald)
        np.testing.assert_allclose(t_graph, t_eager, rtol=1e-5, atol=1e-5)
        # (the bootstrap metric draws new resamples at every call: compare what it is computed from)
        assert np.isfinite(m_graph["test_mae_mean"]) and np.isfinite(m_eager["test_mae_mean"])
        assert float(np.abs(t_eager[:, 0]).std()) > 1e-3
    train.close()


@pytest.mark.gpu
def test_graphed_evaluation_before_the_first_training_step_is_recaptured():
    """An evaluation BEFORE the graphed trainer's bootstrap step (Fitter.test on a loaded checkpoint, an evaluation ahead of
    training) captures graphs that hold the parameters' pre-flatten addresses; the bootstrap then re-seats every p.data into
    the flat buffer.  GraphedEvalStep notices the moved addresses, drops the stale graphs and captures again: the next
    evaluation follows the trained parameters (it used to replay reads of the old storage and return the old predictions)."""
    from equihgnn_amd.batch import MolStore
    from equihgnn_amd.fit import BucketedLoader
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.trainer import GraphedTrainStep
    torch.manual_seed(0)
    mols = _mols(16 * 6, 33)
    args = default_args(method="egnn_equihnns", MLP_hidden=64, output_hidden=32)
    model = MODELS["egnn_equihnns"](1, args).to("cuda:0")
    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.normal_(m.weight, std=m.in_features ** -0.5)
    fitter = Fitter(model, lr=1e-2, step_factory=GraphedTrainStep)
    store = MolStore(mols)
    train = BucketedLoader(store, 16, True, seed=0, device="cuda:0")
    evald = BucketedLoader(store, 16, False, device="cuda:0")
    assert fitter.step.pflat is None                               # nothing has been flattened yet
    _, t_before = fitter.test(evald)                              # captured on the pre-flatten parameter addresses
    assert fitter.eval_step is not None and fitter.eval_step.recaptures == 0
    for b in train:
        fitter.step.step(b)                                       # bootstrap (re-seats p.data) + capture + replays
    assert fitter.step.pflat is not None
    _, t_graph = fitter.test(evald)
    assert fitter.eval_step.recaptures == 1
    fitter.graph_eval = False
    _, t_eager = fitter.test(evald)
    np.testing.assert_allclose(t_graph, t_eager, rtol=1e-5, atol=1e-5)
    assert float(np.abs(t_graph[:, 0] - t_before[:, 0]).max()) > 1e-3          # training at lr 1e-2 did move the predictions
    train.close()


def test_native_collate_equals_the_numpy_restatement_bit_for_bit():
    """MolStore.collate is the library's host-side hb_collate (csrc/collate.hip); MolStore.collate_numpy the array-operation
    restatement of rounds 2-4: same batch bit for bit, unpadded, padded, and into caller-owned staging tensors; extents that
    do not fit raise the same ValueError; a molecule index outside the store is an error, not a read past the arrays."""
    from equihgnn_amd.batch import HBatch, MolStore, bucket_sizes
    store = MolStore(_mols(97, 5))
    rng = np.random.default_rng(0)
    fields = ("x", "pos", "edge_index0", "edge_index1", "edge_attr", "n_e", "e_order", "batch", "y")
    for trial in range(4):
        idx = rng.permutation(len(store))[: int(rng.integers(1, 40))]
        a, b = store.collate(idx), store.collate_numpy(idx)
        for f in fields:
            assert torch.equal(getattr(a, f), getattr(b, f)), f
        tgt = bucket_sizes(*store.extents(idx), 64)
        a, b = store.collate(idx, pad_to=tgt), store.collate_numpy(idx, pad_to=tgt)
        for f in fields:
            assert torch.equal(getattr(a, f), getattr(b, f)), f
        assert a.num_real_graphs == b.num_real_graphs == len(idx) and a.num_graphs == len(idx) + 1
        out = HBatch.empty_packed(tgt[0], tgt[1], tgt[2], len(idx) + 1, pin=False)
        c = store.collate(idx, pad_to=tgt, out=out)
        assert c is out
        for f in fields:
            assert torch.equal(getattr(c, f), getattr(b, f)), f
    n, h, z = store.extents(idx)
    for bad in ((n, h + 1, z), (n + 1, h, z), (n + 1, h + 1, z - 1)):
        with pytest.raises(ValueError):
            store.collate(idx, pad_to=bad)
        with pytest.raises(ValueError):
            store.collate_numpy(idx, pad_to=bad)
    with pytest.raises(IndexError):
        store.collate(np.array([0, len(store)]))
