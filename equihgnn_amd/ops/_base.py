"""Shared helpers of the operator modules: pointers / streams for the C ABI, the in-graph Timeline, scratch that must
outlive a deferred reduction.

Part of equihgnn_amd.ops (host-side operators over libequihgnn_hip.so; no CPU fallback).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import hip


_c_void_p = ctypes.c_void_p


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else _c_void_p(t.data_ptr())


def _stream(device) -> _c_void_p:
    return _c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise hip.HipLibraryError(
            f"{what}: tensor on {t.device}; equihgnn_amd runs on MI355X (HIP) devices only — "
            "there is no CPU fallback")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t.contiguous()


class Timeline:
    """Measurement aid for bench.py: while ``ops.TIMELINE`` is set, the launches of the aggregation kernels are
    bracketed by device-side time stamps (eqh_stamp, one-thread kernels that store the wall clock).  The stamps are
    ordinary launches on the current stream, so they are captured into a hipGraph with everything else and give the
    IN-GRAPH duration of each bracketed launch on every replay -- where HIP events on the launching stream see
    nothing.  ``entries``: (kernel name, algorithmic bytes or flops, slot before, slot after), in launch order."""

    def __init__(self, device, capacity: int = 8192):
        self.slots = torch.zeros(capacity, dtype=torch.int64, device=device)
        self.entries = []
        self.n = 0
        self.khz = int(hip.lib().eqh_wall_clock_khz())

    def stamp(self) -> int:
        i = self.n
        if i >= self.slots.numel():
            raise RuntimeError("Timeline: out of slots")
        self.n += 1
        hip.check(hip.lib().eqh_stamp(_c_void_p(self.slots.data_ptr() + 8 * i), _stream(self.slots.device)), "eqh_stamp")
        return i

    def reset(self):
        self.entries, self.n = [], 0

    def pair(self, name: str = "stamp_pair"):
        """Two stamps back to back: their distance is the launch slot every bracket includes once."""
        a = self.stamp()
        b = self.stamp()
        self.entries.append((name, 0, a, b))

    def read_us(self):
        """[(name, work, microseconds)] from the stamps of the last run (synchronises)."""
        t = self.slots[: self.n].cpu()
        return [(n, w, float(t[b] - t[a]) * 1e3 / self.khz) for n, w, a, b in self.entries]


TIMELINE: Optional[Timeline] = None


def timed(name: str, work, launch):
    """Run ``launch()``; under an active Timeline bracket it with stamps.  ``work``: algorithmic bytes (or flops)."""
    tl = TIMELINE
    if tl is None:
        return launch()
    a = tl.stamp()
    out = launch()
    b = tl.stamp()
    tl.entries.append((name, int(work() if callable(work) else work), a, b))
    return out


class StepSignal:
    """A device counter that a node of the training step's hipGraph bumps (eqh_signal_post) and another stream waits on
    (eqh_signal_wait): how GraphedTrainStep starts the NEXT batch's index build when the running step has passed its
    chip-filling front-end -- the place a model marks with ``ops.signal_point()`` -- instead of at the step's head.
    ``posted``: how many posts the replays so far have enqueued (the host's mirror of the counter's final value)."""

    def __init__(self, device):
        self.counter = torch.zeros(1, dtype=torch.int32, device=device)
        self.posted = 0
        self.armed = False

    def post(self):
        self.armed = False
        hip.check(hip.lib().eqh_signal_post(_ptr(self.counter), _stream(self.counter.device)), "eqh_signal_post")

    def wait(self, target: int, timeout_us: int = 200000):
        """Hold the CURRENT stream until the counter has reached ``target`` (at most timeout_us: a step of the largest workload,
        faformer_equihnns at batch 512, is 19 ms and the wait spans about one step; an expired wait only starts the index build
        early -- the staged buffers' reuse is ordered by an event, not by this signal)."""
        t = ((int(target) + 2 ** 31) % 2 ** 32) - 2 ** 31          # the device counter wraps as int32
        hip.check(hip.lib().eqh_signal_wait(_ptr(self.counter), t, int(timeout_us), _stream(self.counter.device)), "eqh_signal_wait")


class StreamEvent:
    """An event that orders two streams of one device and nothing else (eqh_event_*: no timing, no system-scope fence).  A
    ``torch.cuda.Event`` writes the L2 back and invalidates it when it is recorded; GraphedTrainStep records one at the head of
    every step with the index built ahead (and one on the prefetch stream), which cost such a step 50-60 us -- more than a
    short index build saves.  ``record`` / ``wait`` take the current stream unless given one; re-recording re-arms the event
    (a wait sees the latest record enqueued before it), so one object serves every step."""

    def __init__(self):
        h = ctypes.c_void_p()
        hip.check(hip.lib().eqh_event_create(ctypes.byref(h)), "eqh_event_create")
        self._h = h

    def record(self, stream=None):
        st = torch.cuda.current_stream() if stream is None else stream
        hip.check(hip.lib().eqh_event_record(self._h, ctypes.c_void_p(st.cuda_stream)), "eqh_event_record")

    def wait(self, stream=None):
        st = torch.cuda.current_stream() if stream is None else stream
        hip.check(hip.lib().eqh_event_wait(self._h, ctypes.c_void_p(st.cuda_stream)), "eqh_event_wait")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                hip.lib().eqh_event_destroy(h)
            except Exception:       # noqa: BLE001 -- interpreter shutdown
                pass


SIGNAL: Optional[StepSignal] = None


def signal_point(name: str = "front_end"):
    """Called where the chip stops being full -- "front_end": behind the EGNN edge kernel; "conv_<l>" / "conv_last": before the
    l-th / last application of the conv stack; "readout" / "after_readout": around the 17-workgroup read-out head; "tail": before
    the step's closing reductions.  Under an armed StepSignal that listens for ``name`` -- a
    GraphedTrainStep capturing with index prefetch -- the post lands here; otherwise nothing happens."""
    sg = SIGNAL
    if sg is None:
        return
    seen = getattr(sg, "seen", None)        # (a trainer's probe pass: which points does this model's step reach?)
    if seen is not None:
        seen.add(name)
    elif sg.armed and getattr(sg, "at", "front_end") == name:
        sg.post()


def signal_take(name: str):
    """signal_point(name) for a caller whose NEXT launch can post the signal itself (hg_conv_panel's F2 stage takes a counter):
    returns the armed signal's device counter -- the caller's launch must then bump it -- or None (nothing to do; under a
    trainer's probe pass the point is recorded as reached)."""
    sg = SIGNAL
    if sg is None:
        return None
    seen = getattr(sg, "seen", None)
    if seen is not None:
        seen.add(name)
        return None
    if sg.armed and getattr(sg, "at", "front_end") == name:
        sg.armed = False
        return sg.counter
    return None


def _row_view(t, what):
    """2-D fp32 device tensor usable as a matrix operand in place (unit inner stride, 16-byte aligned rows)."""
    if not (t.dim() == 2 and t.dtype == torch.float32 and t.is_cuda):
        raise TypeError(f"{what}: 2-D float32 device tensor expected")
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _as2d(t: torch.Tensor):
    """The Equiformer wrapper carries a leading 1-dim (equihnn_equiformer.py:82-85); every op
    here reduces along dim -2, so flatten the leading dims of size 1."""
    lead = t.shape[:-2]
    for s in lead:
        if s != 1:
            raise ValueError(f"leading dims must be 1, got {tuple(t.shape)}")
    return t.reshape(t.shape[-2], t.shape[-1]), lead


def _contiguous_run(ts):
    """True if the tensors sit back to back in one storage, in order (each contiguous)."""
    for a, b in zip(ts[:-1], ts[1:]):
        if not (a.is_contiguous() and b.is_contiguous() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                and b.storage_offset() == a.storage_offset() + a.numel()):
            return False
    return ts[0].is_contiguous()


def _stacked_view(ts):
    """[sum rows, C] view over tensors for which _contiguous_run holds (no copy)."""
    rows = sum(t.shape[0] for t in ts)
    return torch.as_strided(ts[0].detach(), (rows, ts[0].shape[1]), (ts[0].shape[1], 1), ts[0].storage_offset())


def _rows_ld(t):
    """(tensor, row stride in floats) for a 2-D fp32 gradient that may be a column block of a wider matrix
    (unit inner stride, 16-byte aligned rows): used as is; anything else is made contiguous first."""
    if (t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]
            and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
        return t, t.stride(0)
    t = _f32c(t)
    return t, t.shape[-1]


_DEFER = {"active": False, "keep": [], "wgrad": [], "colsum": [], "merged": [], "zslab": None, "scratch": None}


def _workspace(nbytes, device):
    """Scratch for one kernel call; while reductions are deferred it must outlive the call (the slabs
    it holds are read by defer_flush), so it is parked until then."""
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if _DEFER["active"]:
        _DEFER["keep"].append(ws)
    return ws


def _acc_target(param):
    """The persistent gradient accumulator of a parameter (set by the graphed trainer), or None."""
    return getattr(param, "_eqh_gbuf", None) if param is not None else None


def _hand_out(grads, targets):
    """Gradients of a parameter group computed into fresh tensors while only SOME of the group own a persistent
    accumulator: those are added to in place (and autograd gets None for them), the rest go to autograd."""
    out = []
    for g, t in zip(grads, targets):
        if t is not None:
            if (g.is_cuda and g.dtype == torch.float32 and t.dtype == torch.float32 and g.is_contiguous() and t.is_contiguous()
                    and g.numel() == t.numel()):
                # (not `t.add_(g)`: inside the step's deferral window the addition rides the one batched reduction launch --
                # six 5-us launches per mhnnm step for its BatchNorm gammas / betas otherwise)
                hip.check(hip.lib().eqh_accumulate(_ptr(g), _ptr(t), g.numel(), _stream(g.device)), "eqh_accumulate")
                if _DEFER["active"]:
                    _DEFER["keep"].append(g)
            else:
                t.add_(g.view_as(t))
            out.append(None)
        else:
            out.append(g)
    return out


def _note_acc(*params):
    """Remember 1-D parameters whose gradient the kernels can accumulate in place."""
    if torch.is_grad_enabled():
        for p in params:
            if p is not None and p.requires_grad and p.is_leaf and not hasattr(p, "_eqh_transient"):
                ACC_PARAMS[id(p)] = p


# parameters seen by ops.linear since the last reset (the trainer decides which of them get a
# persistent gradient accumulator, see trainer.GradBuffers)
LINEAR_PARAMS = {}
ACC_PARAMS = {}   # 1-D parameters (biases, LayerNorm gamma / beta) used through the fused kernels
