import sys, os
sys.path.insert(0, os.getcwd())
import torch
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch, HBatch
from equihgnn_amd.index import HyperIndex
dev = "cuda:0"
B = 8
raw = [synth_batch(B, 2000 + i) for i in range(3)]
ext = [bucket_sizes(b.num_nodes, b.num_hyperedges, b.nnz) for b in raw]
tgt = tuple(max(e[i] for e in ext) for i in range(3))
padded = [pad_batch(b, *tgt).to(dev) for b in raw]
def build(b):
    b._hyper_index = None
    ix = HyperIndex.from_batch(b)
    nbr, d2, csr_t = ix.knn(b.pos, 16, 0)
    return dict(be_rp=ix.by_e.rowptr, be_perm=ix.by_e.perm, be_col=ix.by_e.col, bv_rp=ix.by_v.rowptr, bv_perm=ix.by_v.perm,
                bv_col=ix.by_v.col, pool_rp=ix.pool.rowptr, pool_perm=ix.pool.perm, nbr=nbr, d2=d2, t_rp=csr_t.rowptr, t_perm=csr_t.perm)
eager = [{k: v.clone() for k, v in build(b).items()} for b in padded]
torch.cuda.synchronize(); print("eager index ok", flush=True)
static = HBatch(**{f: (getattr(padded[0], f).clone() if torch.is_tensor(getattr(padded[0], f)) else getattr(padded[0], f)) for f in padded[0].__dataclass_fields__})
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    build(static); build(static)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = build(static)
torch.cuda.synchronize(); print("captured", flush=True)
for rep in range(2):
  for i, b in enumerate(padded):
    for f in b.__dataclass_fields__:
        v = getattr(b, f)
        if torch.is_tensor(v): getattr(static, f).copy_(v)
    g.replay(); torch.cuda.synchronize()
    bad = [k for k in outs if not torch.equal(outs[k], eager[i][k])]
    print("replay batch", i, "mismatching:", bad, flush=True)
