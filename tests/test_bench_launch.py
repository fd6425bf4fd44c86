"""bench.py's own rank launcher (`python bench.py --gpus N` with no torch.distributed.run around it) and the
world-size bookkeeping of the JSON line (reference behaviour: Trainer(devices="auto", strategy="ddp..."),
main.py:271-283)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--batch", "32", "--steps", "2", "--warmup", "1", "--pool", "2", "--no-roofline", "--no-cpu-baseline",
         "--no-pipeline", "--c4-steps", "0"]


def _bench(args, env_extra=None, timeout=600):
    env = dict(os.environ, EQH_NO_TUNABLEOP="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_launcher_refuses_more_ranks_than_devices():
    """No silent single-GPU run: asking for more GPUs than are visible exits non-zero before any GPU work."""
    n = torch.cuda.device_count()
    r = _bench(["--gpus", str(n + 3)] + SMALL)
    assert r.returncode == 2 and "device(s) are visible" in r.stderr
    assert not r.stdout.strip()


def test_mismatched_world_size_is_an_error():
    r = _bench(["--gpus", "1"] + SMALL, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_launcher_starts_two_ranks_gloo_on_one_device():
    """Two ranks sharing the one GPU of the test box (gloo: RCCL cannot form a communicator on a shared device):
    the launcher path, the per-rank batches, the flat-gradient all-reduce between the hipGraphs and the n_gpus
    field are the ones the 8-GPU run uses."""
    r = _bench(["--gpus", "2"] + SMALL, {"EQH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["config"]["parallelism"] == "dp2"
    assert "gloo" in line["config"]["collective"] and line["value"] > 0 and line["scaling"] == "weak"
    assert len(r.stdout.strip().splitlines()) == 1          # ONE JSON line on stdout (library banners go to stderr)


@pytest.mark.gpu
def test_two_ranks_with_the_roofline_measurement():
    """The in-graph roofline measurement runs on rank 0 alone while the other ranks wait at the final barrier: its trainer
    must be rank-local (GraphedTrainStep(collective=False)) -- a collective there has no partner and the job hangs."""
    args = [a for a in SMALL if a != "--no-roofline"] + ["--timeline-replays", "2"]
    r = _bench(["--gpus", "2"] + args, {"EQH_BACKEND": "gloo"}, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["roofline"]["launches_per_step"] == 3 and 0 < line["roofline"]["frac"] < 1


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_launcher_starts_two_ranks_rccl():
    r = _bench(["--gpus", "2"] + SMALL + ["--c4-steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "nccl" in line["config"]["collective"]
    assert line["strong_scaling_c4"]["n_gpus"] == 2 and line["strong_scaling_c4"]["scaling"] == "strong"
