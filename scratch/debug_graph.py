import sys, os
sys.path.insert(0, os.getcwd())
import torch
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
from equihgnn_amd.models import MODELS
from equihgnn_amd.registry import default_args
from equihgnn_amd.trainer import GraphedTrainStep
dev = "cuda:0"
def stage(name, C, B, graph):
    torch.manual_seed(0)
    args = default_args(method="egnn_equihnns", MLP_hidden=C, output_hidden=C // 2)
    m = MODELS["egnn_equihnns"](1, args).to(dev)
    raw = [synth_batch(B, 2000 + i) for i in range(3)]
    ext = [bucket_sizes(b.num_nodes, b.num_hyperedges, b.nnz) for b in raw]
    tgt = tuple(max(e[i] for e in ext) for i in range(3))
    print(name, "target extents", tgt, [(b.num_nodes, b.num_hyperedges, b.nnz) for b in raw], flush=True)
    padded = [pad_batch(b, *tgt).to(dev) for b in raw]
    for b in padded: b.num_real_graphs = B
    if not graph:
        for b in padded:
            out = m(b); torch.nn.functional.mse_loss(out[:B], b.y[:B]).backward()
            torch.cuda.synchronize(); print(name, "eager padded step ok", float(out[0]), flush=True)
        return
    tr = GraphedTrainStep(m, lr=1e-4)
    for i in range(5):
        l = tr.step(padded[i % 3]); torch.cuda.synchronize(); print(name, "step", i, float(l), flush=True)
print("blas backend:", torch.backends.cuda.preferred_blas_library(), flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "B"
if which == "B": stage("B graph C256 B8", 256, 8, True)
if which == "C": stage("C graph C64 B256", 64, 256, True)
if which == "D": stage("D graph C256 B256", 256, 256, True)
