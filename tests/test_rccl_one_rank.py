"""The multi-rank training step on the ONE GPU a test box has: a 1-rank "nccl" (RCCL) process group drives
GraphedTrainStep's collective path (parameter / buffer broadcast, flat-gradient all-reduce, 1 / world scaling) --
reference behaviour: Trainer(strategy="ddp_find_unused_parameters_true"), main.py:271-283.  With one rank every
collective is the identity, so the parameters after K steps must equal those of the rank-local trainer BIT FOR BIT,
whether the all-reduce is a node of the step's hipGraph ("in_graph") or runs eagerly between two graphs ("split")."""
import json
import os
import socket

import pytest
import torch

from golden.common import fill_state_dict

pytestmark = pytest.mark.gpu


def _worker(rank, port, out_path):
    import torch.distributed as dist
    from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
    from equihgnn_amd.models import MODELS
    from equihgnn_amd.registry import default_args
    from equihgnn_amd.trainer import GraphedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    report = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    raw = [synth_batch(8, 810 + i) for i in range(3)]
    ext = [bucket_sizes(b.num_nodes, b.num_hyperedges, b.nnz, 64) for b in raw]
    tgt = tuple(max(e[i] for e in ext) for i in range(3))
    batches = [pad_batch(b, *tgt).to(dev) for b in raw]
    for b in batches:
        b.num_real_graphs = 8
    for method in ("egnn_equihnns", "mhnnm"):      # the second has BatchNorm buffers (DDP's per-forward broadcast)
        finals = {}
        for label, kw in (("local", dict(collective=False)),
                          ("in_graph", dict(force_collective=True)),
                          ("split", dict(force_collective=True, graph_collective=False)),
                          ("refused", dict(force_collective=True))):
            m = MODELS[method](1, default_args(method=method, MLP_hidden=64, output_hidden=32))
            fill_state_dict(m, 5)
            m.to(dev).train()
            tr = GraphedTrainStep(m, lr=1e-3, **kw)
            if label == "refused":      # a backend that cannot record the collective: only THAT makes the step fall back
                def refuse():
                    raise RuntimeError("stand-in: collective not capturable")
                tr._captured_all_reduce = refuse
            losses = [float(tr.step(batches[i % 3])) for i in range(6)]
            torch.cuda.synchronize()
            finals[label] = ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, losses)
            report[f"{method}/{label}"] = {"mode": tr.collective_mode, "capture_error": tr.capture_error,
                                           "graphs": [s["opt"] is not None for s in tr.slots.values()],
                                           "flat_buffers": len(tr.bflat)}
            tr.close()
        for label in ("in_graph", "split", "refused"):
            sd, losses = finals[label]
            assert losses == finals["local"][1], (method, label, losses, finals["local"][1])
            for k, v in sd.items():
                assert torch.equal(v, finals["local"][0][k]), (method, label, k)
    # the trainer's calibration of the index form (two captures of the step, the second one in the middle of training) with the
    # collective INSIDE the graphs: the same steps as the rank-local trainer pinned to the in-step form
    n_long = 2 * (2 + GraphedTrainStep.CAL_WARM + GraphedTrainStep.CAL_STEPS) + 5
    finals = {}
    for label, kw in (("local", dict(collective=False)), ("in_graph", dict(force_collective=True))):
        m = MODELS["egnn_equihnns"](1, default_args(method="egnn_equihnns", MLP_hidden=64, output_hidden=32))
        fill_state_dict(m, 5)
        m.to(dev).train()
        tr = GraphedTrainStep(m, lr=1e-3, **kw)
        if label == "local":
            tr.index_prefetch = False
        losses = [float(tr.step(batches[i % 3], batches[(i + 1) % 3])) for i in range(n_long)]
        torch.cuda.synchronize()
        finals[label] = ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, losses)
        if label == "in_graph":
            report["calibration"] = {"mode": tr.collective_mode, "calibration": tr.calibration, "calibrating": tr.calibrating}
        tr.close()
    assert finals["in_graph"][1] == finals["local"][1]
    for k, v in finals["in_graph"][0].items():
        assert torch.equal(v, finals["local"][0][k]), ("calibration", k)
    # an error that is NOT a refused collective (a kernel's argument check, a bug) must surface, not turn into "split"
    m = MODELS["egnn_equihnns"](1, default_args(method="egnn_equihnns", MLP_hidden=64, output_hidden=32))
    fill_state_dict(m, 5)
    m.to(dev).train()
    tr = GraphedTrainStep(m, lr=1e-3, force_collective=True)
    tr.step(batches[0])
    real, calls = tr._loss_backward, []

    def broken(data):
        calls.append(torch.cuda.is_current_stream_capturing())
        if calls[-1]:
            raise ValueError("stand-in: bad argument inside the captured pass")
        return real(data)
    tr._loss_backward = broken
    try:
        tr.step(batches[0])
        report["kernel_error"] = "swallowed"
    except ValueError as exc:
        report["kernel_error"] = f"raised: {exc}"
    torch.cuda.synchronize()
    tr.close()
    json.dump(report, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_step_through_a_one_rank_rccl_group(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "report.json"
    mp.spawn(_worker, args=(port, str(out)), nprocs=1, join=True)       # RCCL lives and dies in a child process
    rep = json.load(open(out))
    print("one-rank RCCL report:", json.dumps(rep))
    assert rep["backend"] == "nccl" and rep["world"] == 1
    for method in ("egnn_equihnns", "mhnnm"):
        assert rep[f"{method}/local"]["mode"] == "none"
        assert rep[f"{method}/split"]["mode"] == "split" and all(rep[f"{method}/split"]["graphs"])
        ig = rep[f"{method}/in_graph"]
        # the collective is captured into the step's one graph; falling back to "split" is legal only with the reason recorded
        assert ig["mode"] == "in_graph" or (ig["mode"] == "split" and ig["capture_error"]), ig
        if ig["mode"] == "in_graph":
            assert not any(ig["graphs"])                       # no second graph
        rf = rep[f"{method}/refused"]
        assert rf["mode"] == "split" and "not capturable" in rf["capture_error"] and all(rf["graphs"]), rf
    assert rep["kernel_error"].startswith("raised"), rep["kernel_error"]
    cal = rep["calibration"]
    assert not cal["calibrating"] and cal["calibration"]["chosen"] in ("built_ahead", "in_step"), cal
    assert rep["mhnnm/in_graph"]["flat_buffers"] >= 1          # one broadcast per dtype, not one per buffer
