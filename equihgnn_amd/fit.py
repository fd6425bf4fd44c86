"""Training harness equal to the reference's ``LitModel`` + ``pl.Trainer`` configuration
(main.py:21-151,259-293), without Lightning / torchmetrics / PyG (none are in this image):

* loss ``nn.MSELoss`` on the (normalised) target, Adam(lr, weight_decay)          main.py:36,49-63,137-140
* ``ReduceLROnPlateau(mode="min", factor=0.1, patience=10, min_lr=lr*1e-5)``
  stepped once per epoch on ``val_mae_mean``                                      main.py:141-151
* validation / test metrics = bootstrapped (50 resamples) MAE and MSE of
  ``out*std`` vs ``y*std``; ``*_mean`` and ``*_std`` are logged                   main.py:37-42,65-76,90-110
* ``ModelCheckpoint(save_top_k=1, monitor="val_mae_mean", mode="min")`` and
  ``EarlyStopping(monitor="val_mae_mean", patience=50)``                          main.py:259-267
* ``trainer.test(ckpt_path="best")`` gathers the predictions of all ranks          main.py:90-135,285-293
* data: 80/10/10 ``random_split`` and target normalisation                        utils/data_split.py:54-79
* data parallelism: per-rank batches with DistributedSampler semantics, one flat gradient all-reduce
  per step (``trainer.TrainStep``); metrics are reduced over ranks like ``sync_dist=True``.

The model is anything with the plugin contract ``model(data) -> Tensor[B]``: the HIP models on a
GPU, the CPU oracle in the tests.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

import contextlib
import queue
import threading
import weakref
import atexit

from .batch import HBatch, HMol, MolStore, bucket_sizes, collate, shard_indices, shard_permutation
from .trainer import GraphedTrainStep, TrainStep, _world, with_next


# ------------------------------------------------------------------------------------------------
# data split / normalisation
# ------------------------------------------------------------------------------------------------
def split_80_10_10(n: int, seed: int):
    """utils/data_split.py:54-65: random_split into int(0.8 n), int(0.1 n), rest."""
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
    n_tr, n_va = int(0.8 * n), int(0.1 * n)
    return perm[:n_tr], perm[n_tr:n_tr + n_va], perm[n_tr + n_va:]


def normalize_targets_like_reference(y: torch.Tensor):
    """utils/data_split.py:67-72.  The three subsets returned by random_split share ONE underlying
    dataset, and the reference normalises ``subset.dataset.y`` for each of them: the whole target
    column is standardised with the FULL-dataset mean/std, three times in a row.  Reproduced as is
    (returns the transformed targets and the std that main.py passes to LitModel)."""
    mean, std = y.mean(dim=0, keepdim=True), y.std(dim=0, keepdim=True)
    out = y
    for _ in range(3):
        out = (out - mean) / std
    return out, float(std.reshape(-1)[0])


# ------------------------------------------------------------------------------------------------
# metrics (torchmetrics BootStrapper(MeanAbsoluteError / MeanSquaredError, num_bootstraps=50))
# ------------------------------------------------------------------------------------------------
class BootstrapMetrics:
    """Poisson(1) resampling of every update, 50 copies of (sum |err|, sum err^2, count); ``compute``
    returns mean and std over the copies — the quantities LitModel logs as ``{mae,mse}_{mean,std}``."""

    def __init__(self, num_bootstraps: int = 50, seed: int = 0):
        self.nb = num_bootstraps
        self.gen = torch.Generator().manual_seed(seed)
        self.reset()

    def reset(self):
        self.abs = torch.zeros(self.nb, dtype=torch.float64)
        self.sq = torch.zeros(self.nb, dtype=torch.float64)
        self.cnt = torch.zeros(self.nb, dtype=torch.float64)

    def update(self, pred: torch.Tensor, target: torch.Tensor):
        err = (pred.detach().double().cpu() - target.detach().double().cpu()).reshape(-1)
        w = torch.poisson(torch.ones(self.nb, err.numel()), generator=self.gen).double()
        self.abs += w @ err.abs()
        self.sq += w @ (err * err)
        self.cnt += w.sum(1)

    def compute(self) -> Dict[str, float]:
        a, s, c = self.abs.clone(), self.sq.clone(), self.cnt.clone()
        if _world() > 1:  # sync_dist=True
            pack = torch.stack((a, s, c))
            dist.all_reduce(pack)
            a, s, c = pack
        mae, mse = a / c.clamp(min=1), s / c.clamp(min=1)
        return {"mae_mean": float(mae.mean()), "mae_std": float(mae.std()),
                "mse_mean": float(mse.mean()), "mse_std": float(mse.std())}


# ------------------------------------------------------------------------------------------------
# loaders
# ------------------------------------------------------------------------------------------------
class MolLoader:
    """torch_geometric DataLoader(batch_size, shuffle) over a list of HMol restricted to this rank
    (DistributedSampler semantics when more than one rank runs)."""

    def __init__(self, mols: Sequence[HMol], batch_size: int, shuffle: bool, seed: int = 0,
                 device=None, rank: int = 0, world: int = 1):
        self.mols, self.bs, self.shuffle, self.seed = mols, batch_size, shuffle, seed
        self.device, self.rank, self.world = device, rank, world
        self.epoch = 0

    def __iter__(self):
        idx = shard_indices(len(self.mols), self.rank, self.world, self.seed, self.epoch, self.shuffle)
        self.epoch += 1
        for i in range(0, len(idx), self.bs):
            b = collate([self.mols[j] for j in idx[i:i + self.bs]])
            yield b.to(self.device) if self.device is not None else b


_LIVE_LOADERS: "weakref.WeakSet" = weakref.WeakSet()


class BucketedLoader:
    """The loader of the graphed training step: per-rank batches with DistributedSampler semantics, assembled by array
    operations (``MolStore.collate``), padded to ONE static bucket per epoch -- the largest batch of the epoch, rounded
    up by ``batch.bucket_sizes`` -- written straight into pinned, packed staging buffers by a prefetch thread, and
    shipped to the device with one asynchronous copy each.  ``GraphedTrainStep`` then replays one captured hipGraph
    for (almost) every batch of the run; only a smaller last batch has a shape of its own.

    The reference does this with torch_geometric's DataLoader (main.py:227-229): a per-molecule Python collate on
    the training process's own thread."""

    def __init__(self, store: MolStore, batch_size: int, shuffle: bool, seed: int = 0, device=None, rank: int = 0,
                 world: int = 1, quantum: int = 64, prefetch: int = 3, pin: Optional[bool] = None, levels: int = 1):
        self.store, self.bs, self.shuffle, self.seed = store, batch_size, shuffle, seed
        self.device, self.rank, self.world, self.quantum, self.prefetch = device, rank, world, quantum, prefetch
        self.pin = (device is not None and torch.device(device).type == "cuda") if pin is None else pin
        # static shape buckets per run: a ladder below the largest batch (see plan).  Default ONE: measured on MI355X /
        # ROCm 7.2, replaying a DIFFERENT captured hipGraph than the previous step costs ~0.8 ms (2.28 against 1.49 ms
        # per step when consecutive batches alternate between three buckets), far more than the ~3 % of padded rows a
        # ladder saves; it pays only for a sampler that keeps equal-bucket batches together.
        self.levels = max(1, int(levels))
        self.lookahead = True           # start collating the next epoch before the current one is consumed (see _start_epoch)
        self._pending = None
        _LIVE_LOADERS.add(self)         # (closed at interpreter exit: see _close_live_loaders)
        self.epoch = 0
        self.collate_seconds = 0.0      # host time spent assembling batches (all epochs), for the bench line
        self.collated = 0               # molecules assembled
        self._ring = {}                 # (extents, molecules) -> pinned staging buffers, kept across epochs

    def plan(self):
        """This epoch's batches (index arrays) and, per batch, the static extents it is padded to: the top bucket holds
        the largest batch of the run, and ``levels - 1`` smaller buckets one quantum apart below it take the batches
        that fit -- a batch is padded by half a quantum on average instead of by (largest - mean) atoms, ~5 % of a
        256-molecule QM9 batch, all of it GPU work; each bucket is one captured hipGraph."""
        # Every rank holds the whole store and the sampler is a pure function of (seed, epoch, rank): each rank computes the
        # batch extents of ALL ranks and pads to their maximum, so the ranks agree on the static shape of every step without
        # a collective in the loader thread -- what bench.py does with an all-reduce(MAX) of the three sizes.  Ranks that
        # padded to their own maxima would each replay (and, when an epoch raises the envelope, capture) differently shaped
        # graphs around the one captured all-reduce; the 0.8 ms graph switch would then land on every rank's step whenever
        # ANY rank switches.
        perm = shard_permutation(len(self.store), self.world, self.seed, self.epoch, self.shuffle).astype(np.int64)
        shards = [perm[r::self.world] for r in range(self.world)]
        idx = shards[self.rank]
        self.epoch += 1
        batches = [idx[i:i + self.bs] for i in range(0, len(idx), self.bs)]
        # extents of every batch at once: prefix sums of the permuted per-molecule sizes (shards have equal lengths)
        cut = np.minimum(np.arange(0, len(idx) + self.bs, self.bs), len(idx))
        counts = (self.store.n_nodes, self.store.n_he, self.store.n_inc)
        sizes = [np.max([np.diff(np.concatenate(([0], np.cumsum(c[sh])))[cut]) for sh in shards], axis=0) for c in counts]
        ext = [sz.max() for sz in sizes]
        # ONE bucket for (almost) the whole run: the largest batch seen so far, a little above the first epoch's own
        # maximum (B x mean + 3.3 sigma sqrt(B) of the per-molecule sizes: the expected maximum of a few hundred batches)
        # so that later epochs -- other permutations, other maxima -- replay the graph captured in the first one; an
        # epoch whose largest batch still exceeds the envelope raises it (one more capture).  Every padded atom is
        # GPU work: the 4.5 sigma bound used before cost 3 % of the step.
        nb = min(self.bs, len(idx))
        stat = [c.mean() * nb + 3.3 * c.std() * np.sqrt(nb) for c in (self.store.n_nodes, self.store.n_he, self.store.n_inc)]
        env = getattr(self, "_envelope", None)
        ext = [max(int(e), int(np.ceil(s_)) if env is None else 0, 0 if env is None else env[i])
               for i, (e, s_) in enumerate(zip(ext, stat))]
        if nb == self.bs:
            self._envelope = ext
        top = bucket_sizes(ext[0], ext[1], ext[2], self.quantum)
        q = (self.quantum, self.quantum, 2 * self.quantum)
        ladder = [tuple(t - l * qq for t, qq in zip(top, q)) for l in range(self.levels)]      # ladder[0] = top
        # the rungs in use are those the FIRST epoch populated (each is then captured in that epoch); later epochs choose
        # among them only, so a rare small or large batch never triggers a capture (seconds, with GEMM tuning) mid-run
        known = getattr(self, "_rungs", None)
        if known is not None and known[0] != top:
            known = None                                      # the envelope grew: a new ladder
        tgts, used = [], {0}
        for i in range(len(batches)):
            need = [int(sz[i]) + 1 for sz in sizes]          # (one spare slot: the padding molecule's own node / hyperedge)
            lvl = 0
            for cand in range(1, len(ladder)):
                if known is not None and cand not in known[1]:
                    continue
                if all(n <= t and t > 0 for n, t in zip(need, ladder[cand])):
                    lvl = cand
                else:
                    break
            used.add(lvl)
            tgts.append(ladder[lvl])
        if known is None and nb == self.bs:
            self._rungs = (top, used)
        return batches, tgts

    def _start_epoch(self, q=None):
        """Plan the next epoch and start its prefetch thread; returns the epoch's (queue, thread, stop flag).  ``q``: the
        queue of the epoch this one is started AHEAD of.  Both epochs then share it, so the bound on batches in flight
        (queue depth + the one being staged + the three the consumer holds <= the ring of prefetch + 4 buffers) also holds
        across the epoch boundary -- with a queue of its own the new epoch would refill ring buffers that still wait,
        unconsumed, in the old one."""
        import time
        # planning advances the shuffle epoch (and may raise the envelope / fix the ladder): remember the state before it,
        # so that an epoch that was started AHEAD and is never consumed (close()) does not skip a permutation -- the
        # sequence of epochs must not depend on lookahead (DistributedSampler.set_epoch semantics)
        undo = (self.epoch, getattr(self, "_envelope", None), getattr(self, "_rungs", None))
        batches, tgts = self.plan()
        ring = self._ring               # (pinning host memory costs milliseconds per buffer: allocate once per shape)
        if q is None:
            q = queue.Queue(maxsize=self.prefetch)
        cuda = self.device is not None and torch.device(self.device).type == "cuda"
        side = None
        if cuda:    # host-to-device copies are issued by the prefetch thread on a stream of their own: they overlap the
            #         previous step's kernels and leave the training thread one stream-wait per batch
            side = self._side = getattr(self, "_side", None) or torch.cuda.Stream(self.device)

        def stage(tgt, n_mols):
            """a free staging pair (pinned host buffer, device buffer) for a batch of n_mols molecules (ring of
            prefetch + 4 per shape: the queue, the one being staged, and the THREE the consumer may hold -- the batch it steps
            on, the look-ahead batch behind it (trainer.with_next) and the one whose release is still to be recorded):
            [host, h2d_done, device, consumed]"""
            slot = ring.setdefault((tgt, n_mols), {"bufs": [], "next": 0})
            if len(slot["bufs"]) < self.prefetch + 4:
                host = HBatch.empty_packed(tgt[0], tgt[1], tgt[2], n_mols + 1, pin=self.pin)
                with torch.cuda.stream(side) if cuda else contextlib.nullcontext():
                    devb = host.to(self.device) if cuda else None
                slot["bufs"].append([host, None, devb, None])
                return slot["bufs"][-1]
            ent = slot["bufs"][slot["next"] % len(slot["bufs"])]
            slot["next"] += 1
            if ent[1] is not None:
                ent[1].synchronize()          # its previous host-to-device copy has finished (the host buffer is free)
            return ent

        stop = threading.Event()        # set when the consumer leaves early: the producer must not block on q.put
        adopted = threading.Event()     # set when a consumer starts on this epoch (__iter__)

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            # An exception here (pinned-memory OOM, a HIP error on the side stream, a ValueError from collate) must end
            # the epoch LOUDLY: it travels through the queue and is re-raised by the consumer.  Swallowed, the epoch
            # would just be shorter on this rank, and under data parallelism the next all-reduce would hang.
            try:
                for b, tgt in zip(batches, tgts):
                    if stop.is_set():
                        return
                    ent = stage(tgt, len(b))
                    t0 = time.perf_counter()
                    self.store.collate(b, pad_to=tgt, out=ent[0])
                    self.collate_seconds += time.perf_counter() - t0
                    self.collated += len(b)
                    if cuda:
                        with torch.cuda.stream(side):
                            if ent[3] is not None and not ent[3].query():
                                # the trainer has taken the device buffer's previous content (a host-side query first: the
                                # ring is prefetch + 4 deep, so the event has almost always fired and no wait enters the stream)
                                side.wait_event(ent[3])
                            ent[2]._flat.copy_(ent[0]._flat, non_blocking=True)
                            ent[2].num_real_graphs = getattr(ent[0], "num_real_graphs", None)
                            ev = torch.cuda.Event()
                            ev.record(side)
                            ent[1] = ev
                    if not put(ent):
                        return
                if put(None) and self.lookahead:
                    # the next epoch's first batches are collated and shipped while the trainer still works through the
                    # tail of this one: no pipeline drain (a few milliseconds of idle GPU) at the epoch boundary.  ONE
                    # epoch ahead only: an epoch that fits in the queue whole finishes before anybody consumes it, and
                    # must not go on to start the one after it (and so on, without end)
                    while not stop.is_set() and not adopted.wait(0.05):
                        pass
                    if not stop.is_set():
                        self._pending = self._start_epoch(q)
            except BaseException as exc:  # noqa: BLE001 -- handed to the consumer, which re-raises it
                put(exc)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        return {"q": q, "th": th, "stop": stop, "cuda": cuda, "adopted": adopted, "undo": undo}

    def close(self):
        """Stop a prefetch thread that was started ahead for an epoch that will not be consumed."""
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is not None:
            pend["stop"].set()
            pend["th"].join()
            if not pend["adopted"].is_set():       # planned, never consumed: the next epoch is this one again
                self.epoch, env, rungs = pend["undo"]
                for name, val in (("_envelope", env), ("_rungs", rungs)):
                    if val is None:
                        self.__dict__.pop(name, None)
                    else:
                        setattr(self, name, val)

    def __iter__(self):
        pend, self._pending = getattr(self, "_pending", None), None
        ep = pend if pend is not None else self._start_epoch()
        q, th, stop, cuda = ep["q"], ep["th"], ep["stop"], ep["cuda"]
        ep["adopted"].set()
        # two Python threads share the interpreter lock; with the default 5 ms switch interval the training thread (a few
        # hundred microseconds of work per 1.5 ms step) can wait milliseconds for the collating thread to yield it
        import sys
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 2e-4))
        prev = None

        def fresh(b):
            # The ring hands the SAME HBatch objects out again and again (prefetch + 4 per shape, kept across epochs),
            # and HyperIndex.from_batch caches the batch's CSRs / kNN lists on the object: a refilled buffer must not
            # carry the index of the molecules it held before.
            b._hyper_index = None
            b._generation = getattr(b, "_generation", 0) + 1     # (GraphedTrainStep's index prefetch: new contents, same object)
            return b

        done = False
        older = None
        try:
            while True:
                ent = q.get()
                # A consumer with one batch of look-ahead (trainer.with_next) asks for batch t+1 BEFORE it enqueues the step
                # on batch t: what it enqueued by now reads the batch handed out TWO gets ago, not the previous one.
                if older is not None and cuda:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    older[3] = ev
                older = prev
                if ent is None:
                    done = True
                    break
                if isinstance(ent, BaseException):
                    raise RuntimeError("BucketedLoader: the prefetch thread failed") from ent
                if self.device is None:
                    yield fresh(ent[0])
                    continue
                if not cuda:
                    yield ent[0].to(self.device)
                    continue
                if not ent[1].query():        # (its copy was issued `prefetch` batches ago: normally complete -- no stream wait)
                    torch.cuda.current_stream(self.device).wait_event(ent[1])
                prev = ent
                yield fresh(ent[2])
        finally:
            # also reached when the consumer abandons the generator (an exception in step(), a `break`): release the
            # producer, which may be blocked on a full queue holding ring entries the next __iter__ shares
            if not done:
                stop.set()
            if cuda:
                # the consumer left with the last yielded device buffers possibly still being read by kernels it
                # enqueued: record that point, so the next producer waits for it before refilling them
                for e_ in (older, prev):
                    if e_ is not None:
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(self.device))
                        e_[3] = ev
            th.join()                   # (a finished producer has also started the next epoch by now: see _start_epoch)
            if not done:
                self.close()            # (an epoch started ahead of an abandoned one is abandoned too)
            sys.setswitchinterval(old_interval)


def _close_live_loaders():
    """At interpreter exit: stop and join every prefetch thread that was started ahead.  A daemon thread killed inside a
    HIP call (an event record, a pinned copy) while the runtime is torn down aborts the process."""
    for ld in list(_LIVE_LOADERS):
        try:
            ld.close()
        except Exception:  # noqa: BLE001 -- nothing useful can be done at exit
            pass


atexit.register(_close_live_loaders)


def _drop_index(data):
    """A loader may hand the same batch OBJECT out with new contents: the per-batch index cached on it
    (HyperIndex.from_batch) belongs to the previous contents."""
    try:
        data._hyper_index = None
    except Exception:
        pass


def _real(out, data):
    """Predictions / targets of the real molecules of a (possibly padded) batch."""
    nb = getattr(data, "num_real_graphs", None)
    return (out, data.y) if not nb else (out[:nb], data.y[:nb])


# ------------------------------------------------------------------------------------------------
# the fit / test loop
# ------------------------------------------------------------------------------------------------
@dataclass
class FitResult:
    history: List[Dict[str, float]] = field(default_factory=list)
    best_epoch: int = -1
    best_val_mae: float = float("inf")
    stopped_early: bool = False
    best_state: Optional[dict] = None


class Fitter:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-4, weight_decay: float = 0.0,
                 std: Optional[float] = None, patience_lr: int = 10, patience_stop: int = 50,
                 step_factory: Callable = TrainStep, metric_seed: int = 0):
        self.model, self.lr, self.std = model, lr, std
        self.step = step_factory(model, lr=lr, weight_decay=weight_decay)
        self.metrics = BootstrapMetrics(50, metric_seed)
        self.patience_lr, self.patience_stop = patience_lr, patience_stop
        self.sched = None
        self.eval_step = None           # GraphedEvalStep, built at the first evaluation of a graphed fit (see _predict)
        self.graph_eval = True

    def _predict(self, data):
        """model(data) for an evaluation pass (eval mode, no_grad): a replayed forward-only hipGraph when the training step is
        the graphed one and the batch is a padded static-shape batch on the GPU (trainer.GraphedEvalStep), else eager."""
        from .trainer import GraphedEvalStep, GraphedTrainStep
        if (self.graph_eval and isinstance(self.step, GraphedTrainStep) and getattr(data, "num_real_graphs", None)
                and data.y.is_cuda):
            if self.eval_step is None:
                self.eval_step = GraphedEvalStep(self.model)
            return self.eval_step(data)
        _drop_index(data)
        return self.model(data)

    def _evaluate(self, loader) -> Dict[str, float]:
        self.model.eval()
        self.metrics.reset()
        scale = self.std if self.std else 1.0          # main.py:67-70: `if self.std:`
        with torch.no_grad():
            for data in loader:
                out, y = _real(self._predict(data), data)
                self.metrics.update(out * scale, y * scale)
        self.model.train()
        return self.metrics.compute()

    def fit(self, train_loader, valid_loader, epochs: int) -> FitResult:
        try:
            return self._fit(train_loader, valid_loader, epochs)
        finally:    # the loaders may have started an epoch ahead (BucketedLoader.lookahead) that nobody will consume
            for ld in (train_loader, valid_loader):
                if hasattr(ld, "close"):
                    ld.close()

    def _fit(self, train_loader, valid_loader, epochs: int) -> FitResult:
        res = FitResult()
        bad = 0
        for epoch in range(epochs):
            tot, n = 0.0, 0
            if isinstance(self.step, GraphedTrainStep):     # one batch of look-ahead: the next batch's index is built beside this step
                for data, nxt in with_next(train_loader):
                    tot += float(self.step.step(data, nxt))
                    n += 1
            else:
                for data in train_loader:
                    tot += float(self.step.step(data))
                    n += 1
            if self.sched is None and self.step.opt is not None:  # built lazily with the optimiser
                self.sched = torch.optim.lr_scheduler.ReduceLROnPlateau(
                    self.step.opt, mode="min", factor=0.1, patience=self.patience_lr, min_lr=self.lr * 1e-5)
            val = self._evaluate(valid_loader)
            lr_now = self.step.opt.param_groups[0]["lr"] if self.step.opt is not None else self.lr
            res.history.append({"epoch": epoch, "train_loss": tot / max(n, 1), "lr": float(lr_now),
                                **{"val_" + k: v for k, v in val.items()}})
            if self.sched is not None:
                self.sched.step(val["mae_mean"])
            if val["mae_mean"] < res.best_val_mae:     # ModelCheckpoint(save_top_k=1) + EarlyStopping
                res.best_val_mae, res.best_epoch, bad = val["mae_mean"], epoch, 0
                res.best_state = copy.deepcopy(self.model.state_dict())
            else:
                bad += 1
                if bad >= self.patience_stop:
                    res.stopped_early = True
                    break
        return res

    def test(self, test_loader, best_state: Optional[dict] = None):
        """trainer.test(ckpt_path="best"): metrics plus the gathered (pred, truth) table that
        main.py:122-132 logs as test_results.csv (unscaled, as there)."""
        if best_state is not None:
            self.model.load_state_dict(best_state)
        self.model.eval()
        self.metrics.reset()
        scale = self.std if self.std else 1.0
        preds, truth = [], []
        try:
            with torch.no_grad():
                for data in test_loader:
                    out, y = _real(self._predict(data), data)
                    self.metrics.update(out * scale, y * scale)
                    preds.append(out.detach().float().cpu())
                    truth.append(y.detach().float().cpu())
        finally:
            if hasattr(test_loader, "close"):       # (one pass only: drop the epoch a BucketedLoader started ahead)
                test_loader.close()
        p, t = torch.cat(preds), torch.cat(truth)
        if _world() > 1:  # self.all_gather(preds)
            gp = [None] * _world()
            gt = [None] * _world()
            dist.all_gather_object(gp, p)
            dist.all_gather_object(gt, t)
            p, t = torch.cat(gp), torch.cat(gt)
        self.model.train()
        return {"test_" + k: v for k, v in self.metrics.compute().items()}, np.stack((p.numpy(), t.numpy()), 1)
