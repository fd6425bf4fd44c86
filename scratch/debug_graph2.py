import sys, os
sys.path.insert(0, os.getcwd())
import torch
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
from equihgnn_amd.models import MODELS
from equihgnn_amd.registry import default_args
from equihgnn_amd.trainer import GraphedTrainStep
dev = "cuda:0"
if len(sys.argv) > 2: torch.backends.cuda.preferred_blas_library(sys.argv[2])
print("blas", torch.backends.cuda.preferred_blas_library(), flush=True)
C, B = int(sys.argv[1]), 8
torch.manual_seed(0)
args = default_args(method="egnn_equihnns", MLP_hidden=C, output_hidden=C // 2)
m = MODELS["egnn_equihnns"](1, args).to(dev)
raw = [synth_batch(B, 2000 + i) for i in range(3)]
ext = [bucket_sizes(b.num_nodes, b.num_hyperedges, b.nnz) for b in raw]
tgt = tuple(max(e[i] for e in ext) for i in range(3))
padded = [pad_batch(b, *tgt).to(dev) for b in raw]
for b in padded: b.num_real_graphs = B
tr = GraphedTrainStep(m, lr=1e-4)
def sync(msg):
    torch.cuda.synchronize(); print(msg, flush=True)
tr.step(padded[0]); sync("bootstrap ok")
tr.step(padded[0]); sync("capture + replay#1 (same batch 0) ok")
slot = next(iter(tr.slots.values()))
slot["graph"].replay(); sync("fwd/bwd replay#2 same data ok")
tr.opt_graph.replay(); sync("opt replay#2 ok")
slot["graph"].replay(); sync("fwd/bwd replay#3 same data ok")
st = slot["static"]
for f in padded[1].__dataclass_fields__:
    v = getattr(padded[1], f)
    if torch.is_tensor(v): getattr(st, f).copy_(v)
sync("copied batch 1")
slot["graph"].replay(); sync("fwd/bwd replay#4 batch 1 ok")
tr.opt_graph.replay(); sync("opt replay ok")
print("loss", float(slot["loss"]))
