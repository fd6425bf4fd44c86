import sys, os
sys.path.insert(0, os.getcwd())
import torch
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
from equihgnn_amd.models import MODELS
from equihgnn_amd.registry import default_args
dev = "cuda:0"
C, B = 256, 8
torch.manual_seed(0)
args = default_args(method="egnn_equihnns", MLP_hidden=C, output_hidden=C // 2)
m = MODELS["egnn_equihnns"](1, args).to(dev)
raw = [synth_batch(B, 2000 + i) for i in range(3)]
ext = [bucket_sizes(b.num_nodes, b.num_hyperedges, b.nnz) for b in raw]
tgt = tuple(max(e[i] for e in ext) for i in range(3))
padded = [pad_batch(b, *tgt).to(dev) for b in raw]
for rep in range(2):
  for i, b in enumerate(padded):
    b._hyper_index = None
    out = m(b); torch.cuda.synchronize(); print("eager fwd ok batch", i, flush=True)
    torch.nn.functional.mse_loss(out[:B], b.y[:B]).backward(); torch.cuda.synchronize(); print("eager bwd ok batch", i, flush=True)
print("ALL EAGER OK")
