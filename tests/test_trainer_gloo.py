"""World-size-2 data-parallel step on CPU (gloo): the flat-gradient all-reduce of
equihgnn_amd.trainer reproduces what DDP does in the reference (main.py:281): every rank ends a
step with identical parameters, equal to single-process Adam on the rank-averaged gradient, and
parameters of dead branches keep grad=None on every rank.  The model driven here is the CPU oracle
(the trainer is model-agnostic; the HIP models need a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from equihgnn_amd.batch import shard_indices, synth_batch
from equihgnn_amd.registry import default_args
from equihgnn_amd.trainer import TrainStep

ARGS = dict(method="egnn_equihnns", MLP_hidden=32, output_hidden=16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(seed):
    from oracle import ref_models as O
    torch.manual_seed(seed)
    m = O.MODELS["egnn_equihnns"](1, default_args(**ARGS))
    for mod in m.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.normal_(mod.weight, std=mod.in_features ** -0.5)
    return m


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _make_model(seed=100 + rank)  # different init per rank: the broadcast must fix that
    tr = TrainStep(model, lr=1e-2)
    losses = []
    for step in range(3):
        data = synth_batch(6, 500 + 10 * step + rank)
        losses.append(float(tr.step(data)))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    dead = [n for n, p in model.named_parameters() if p.grad is None]
    torch.save({"sd": sd, "dead": dead, "losses": losses}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_matches_single_process_average(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), f"{k} differs across ranks"
    assert r0["dead"] == r1["dead"] and any("coors_mlp" in n for n in r0["dead"])

    # single-process emulation: same init as rank 0, Adam on the mean of the two ranks' gradients.
    # One thread, like the workers: Adam turns rounding-level differences of near-zero gradients
    # into O(lr) updates, so the arithmetic has to be the same, not merely close.
    torch.set_num_threads(1)
    model = _make_model(seed=100)
    live = None
    opt = None
    for step in range(3):
        grads = []
        for rank in range(world):
            for p in model.parameters():
                p.grad = None
            data = synth_batch(6, 500 + 10 * step + rank)
            torch.nn.functional.mse_loss(model(data), data.y).backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in model.parameters()])
        if live is None:
            live = [p for p, g in zip(model.parameters(), grads[0]) if g is not None]
            opt = torch.optim.Adam(live, lr=1e-2)
        for p, g0, g1 in zip(model.parameters(), *grads):
            p.grad = None if g0 is None else (g0 + g1) * 0.5
        opt.step()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(r0["sd"][k].numpy(), v.numpy(), atol=2e-6, rtol=1e-5, err_msg=k)


def test_shard_indices_partition_like_distributed_sampler():
    n, world = 103, 4
    parts = [shard_indices(n, r, world, seed=7, epoch=3) for r in range(world)]
    assert len({len(p) for p in parts}) == 1 and len(parts[0]) == -(-n // world)
    assert set().union(*map(set, parts)) == set(range(n))
    assert parts != [shard_indices(n, r, world, seed=7, epoch=4) for r in range(world)]
    assert shard_indices(10, 1, 2, seed=0, shuffle=False) == [1, 3, 5, 7, 9]


def test_shard_indices_repeat_a_store_smaller_than_the_world():
    """DistributedSampler pads by repeating the permutation as often as needed: no rank gets an empty shard."""
    parts = [shard_indices(3, r, 8, seed=1) for r in range(8)]
    assert all(len(p) == 1 for p in parts) and {p[0] for p in parts} == {0, 1, 2}


def _agree_worker(rank, world, port, out_dir):
    import types

    from equihgnn_amd.trainer import GraphedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    me = types.SimpleNamespace(gflat=torch.zeros(4))
    got = [GraphedTrainStep._agree_on_mode(me, mode) for mode in
           ("in_graph", "split", "in_graph" if rank == 0 else "split", "split" if rank == 0 else "in_graph")]
    torch.save(got, os.path.join(out_dir, f"agree{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_form_of_the_step(tmp_path):
    """GraphedTrainStep decides at capture time whether the gradient all-reduce is a node of the step's hipGraph; a rank that
    fell back to the split form alone would leave the others in a collective it never enters (the reference's DDP,
    main.py:271-283, has one collective sequence by construction).  "in_graph" survives only if EVERY rank reports it."""
    mp.spawn(_agree_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert torch.load(tmp_path / f"agree{rank}.pt") == ["in_graph", "split", "split", "split"]


def test_bucketed_loader_pads_every_rank_to_the_same_extents():
    """Ranks draw different molecules, hence different natural batch sizes; the loader of the graphed step must still hand
    every rank the SAME padded extents at every step (one captured graph per rank around the one all-reduce), computed
    without a collective: each rank evaluates the whole epoch's sampler."""
    from equihgnn_amd.batch import MolStore, synth_molecule
    from equihgnn_amd.fit import BucketedLoader
    rng = np.random.default_rng(11)
    store = MolStore([synth_molecule(rng) for _ in range(400)])
    world = 4
    for levels in (1, 3):
        plans = []
        for r in range(world):
            ld = BucketedLoader(store, 32, shuffle=True, seed=5, rank=r, world=world, levels=levels)
            ld.lookahead = False
            plans.append([ld.plan() for _ in range(3)])          # three epochs
            ld.close()
        for e in range(3):
            t0 = plans[0][e][1]
            for r in range(1, world):
                assert plans[r][e][1] == t0, (levels, e, r)
            # ... and the ranks really drew different batches with different natural sizes
            nat = [[int(store.n_nodes[b].sum()) for b in plans[r][e][0]] for r in range(world)]
            assert len({tuple(n) for n in nat}) == world
            # every rank's batches fit the agreed extents
            for r in range(world):
                for b, t in zip(plans[r][e][0], plans[r][e][1]):
                    assert store.n_nodes[b].sum() < t[0] and store.n_he[b].sum() < t[1] and store.n_inc[b].sum() <= t[2]


def test_with_next_pairs_every_batch_with_its_successor():
    """The one batch of look-ahead GraphedTrainStep.step(data, next_data) uses to build the next batch's index beside the
    current step: every element once, in order, the last one with None; works on one-shot iterators."""
    from equihgnn_amd.trainer import with_next
    assert list(with_next([1, 2, 3])) == [(1, 2), (2, 3), (3, None)]
    assert list(with_next(iter([7]))) == [(7, None)]
    assert list(with_next([])) == []


class _FakeClock:
    """Stands in for the device and the wall clock while GraphedTrainStep._calibrate is driven on the CPU: every step of the
    form that is 'running' costs a fixed number of milliseconds."""

    def __init__(self, cost):
        self.cost, self.now, self.form = cost, 0.0, "built_ahead"

    def step(self):
        self.now += self.cost[self.form] * 1e-3


def _calibration_stub(monkeypatch, cost):
    import time
    import types

    from equihgnn_amd.trainer import GraphedTrainStep
    clock = _FakeClock(cost)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(time, "perf_counter", lambda: clock.now)
    tr = GraphedTrainStep.__new__(GraphedTrainStep)
    tr.prefetch_policy, tr._prefetch_on, tr._cal, tr._alt_slots, tr.calibration = "auto", True, None, None, None
    tr.slots = {}

    def step(self, key="bucket"):
        """What GraphedTrainStep.step does around the calibration: the hook, a capture when the bucket has no graph, one step."""
        if self.prefetch_policy == "auto" and self.calibration is None:
            self._calibrate(key)
        if key not in self.slots:
            self.slots[key] = {"form": "built_ahead" if self._prefetch_on else "in_step"}
        clock.form = self.slots[key]["form"]
        clock.step()
    tr.step = types.MethodType(step, tr)
    return tr, clock


@pytest.mark.parametrize("cost,chosen", [({"built_ahead": 1.00, "in_step": 1.10}, "built_ahead"),
                                         ({"built_ahead": 1.10, "in_step": 1.00}, "in_step"),
                                         ({"built_ahead": 1.000, "in_step": 1.005}, "in_step")])      # inside the 1 % margin
def test_calibration_times_a_window_of_each_form_and_keeps_the_faster(monkeypatch, cost, chosen):
    """GraphedTrainStep._calibrate (host logic, no GPU): capture + CAL_WARM replays + CAL_STEPS timed steps built ahead, the same
    in the step graph (the first form's graphs kept aside), then the decision -- the built-ahead form has to win by 1 %."""
    from equihgnn_amd.trainer import GraphedTrainStep
    tr, clock = _calibration_stub(monkeypatch, cost)
    per_form = 1 + GraphedTrainStep.CAL_WARM + GraphedTrainStep.CAL_STEPS
    forms = []
    for i in range(2 * per_form + 3):
        tr.step()
        forms.append(clock.form)
        assert tr.calibrating == (i < 2 * per_form)
    assert forms[:per_form] == ["built_ahead"] * per_form and forms[per_form:2 * per_form] == ["in_step"] * per_form
    assert forms[2 * per_form:] == [chosen] * 3
    cal = tr.calibration
    assert cal["chosen"] == chosen and tr.index_prefetch == (chosen == "built_ahead") and tr._alt_slots is None
    assert cal["built_ahead_ms"] == pytest.approx(cost["built_ahead"], rel=1e-6) and cal["in_step_ms"] == pytest.approx(cost["in_step"], rel=1e-6)
    assert tr.slots["bucket"]["form"] == chosen


def test_calibration_restarts_its_window_when_the_bucket_changes_and_gives_up_when_buckets_keep_alternating(monkeypatch):
    from equihgnn_amd.trainer import GraphedTrainStep
    tr, clock = _calibration_stub(monkeypatch, {"built_ahead": 1.0, "in_step": 1.2})
    for _ in range(5):
        tr.step("a")
    tr.step("b")                                   # another bucket: the window starts again on it
    assert tr._cal["key"] == "b" and tr._cal["n"] == 0 and tr.calibrating
    per_form = 1 + GraphedTrainStep.CAL_WARM + GraphedTrainStep.CAL_STEPS
    for _ in range(2 * per_form + 1):
        tr.step("b")
    assert not tr.calibrating and tr.calibration["chosen"] == "built_ahead"
    # buckets that alternate faster than a window: after 16 restarts the running form stays, undecided windows are dropped
    tr2, _ = _calibration_stub(monkeypatch, {"built_ahead": 1.0, "in_step": 1.2})
    for i in range(60):
        tr2.step("ab"[i % 2])
        if not tr2.calibrating:
            break
    assert not tr2.calibrating and tr2.calibration["built_ahead_ms"] is None and tr2.calibration["chosen"] == "built_ahead"


def test_assigning_index_prefetch_pins_the_form():
    from equihgnn_amd.trainer import GraphedTrainStep
    tr = GraphedTrainStep.__new__(GraphedTrainStep)
    tr.prefetch_policy, tr._prefetch_on, tr._cal, tr._alt_slots, tr.calibration = "auto", True, {"form": "built_ahead"}, None, None
    assert tr.calibrating
    tr.index_prefetch = False
    assert not tr.calibrating and tr.prefetch_policy == "off" and tr._cal is None and not tr.index_prefetch
    tr.index_prefetch = True
    assert tr.prefetch_policy == "on" and tr.index_prefetch and not tr.calibrating
