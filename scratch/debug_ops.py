import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np, torch
from common import *
from equihgnn_amd import ops
from equihgnn_amd.index import HyperIndex
from oracle import ref_models as O
case = load_case("egnn_equihnns_c256")
d = batch_from_case(case)
ix = HyperIndex.from_batch(d.to("cuda"))
v, e = d.edge_index0, d.edge_index1
N, M = ix.N, ix.M
import numpy as np
def npcsr(key, other, n):
    order = np.argsort(key.numpy(), kind="stable"); rp = np.concatenate([[0], np.cumsum(np.bincount(key.numpy(), minlength=n))])
    return rp, order, other.numpy()[order]
for nm, csr, key, other, n in (("by_e", ix.by_e, e, v, M), ("by_v", ix.by_v, v, e, N)):
    rp, perm, col = npcsr(key, other, n)
    print(nm, "rowptr ok", np.array_equal(rp, csr.rowptr.cpu().numpy()), "perm ok", np.array_equal(perm, csr.perm.cpu().numpy()), "col ok", np.array_equal(col, csr.col.cpu().numpy()))
    bad = np.nonzero(col != csr.col.cpu().numpy())[0]
    print("   bad col positions", bad, "rows", np.searchsorted(rp, bad, side="right")-1)
for C in (64, 256):
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, C, generator=g); w = torch.randn(M, C, generator=g)
    Xc = X.clone().requires_grad_(True); (O.segment_reduce(Xc[v], e, M, "mean") * w).sum().backward()
    Xd = X.cuda().requires_grad_(True); (ops.reduce_gathered(Xd, ix.by_e, ix.by_v, "mean") * w.cuda()).sum().backward()
    err = (Xd.grad.cpu() - Xc.grad).abs().max(-1).values
    print("C", C, "reduce_gathered bwd bad rows", torch.nonzero(err > 1e-4).flatten().tolist())
    wn = torch.randn(len(v), C, generator=g)
    Xc = X.clone().requires_grad_(True); (Xc[v] * wn).sum().backward()
    Xd = X.cuda().requires_grad_(True); (ops.gather_rows(Xd, ix.v32, ix.by_v) * wn.cuda()).sum().backward()
    err = (Xd.grad.cpu() - Xc.grad).abs().max(-1).values
    print("C", C, "gather_rows(v) bwd bad rows", torch.nonzero(err > 1e-4).flatten().tolist())
    Q = torch.randn(M, C, generator=g)
    Qc = Q.clone().requires_grad_(True); (Qc[e] * wn).sum().backward()
    Qd = Q.cuda().requires_grad_(True); (ops.gather_rows(Qd, ix.e32, ix.by_e) * wn.cuda()).sum().backward()
    err = (Qd.grad.cpu() - Qc.grad).abs().max(-1).values
    print("C", C, "gather_rows(e) bwd bad rows", torch.nonzero(err > 1e-4).flatten().tolist())
    H = torch.randn(len(v), C, generator=g); wv = torch.randn(N, C, generator=g)
    Hc = H.clone().requires_grad_(True); (O.segment_reduce(Hc, v, N, "mean") * wv).sum().backward()
    Hd = H.cuda().requires_grad_(True); (ops.reduce_entries(Hd, ix.by_v, ix.v32, "mean") * wv.cuda()).sum().backward()
    err = (Hd.grad.cpu() - Hc.grad).abs().max(-1).values
    print("C", C, "reduce_entries bwd bad rows", torch.nonzero(err > 1e-4).flatten().tolist())
