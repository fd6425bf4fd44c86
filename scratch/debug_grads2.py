import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np, torch
from common import *
from equihgnn_amd import models
from oracle import ref_models as O
name = sys.argv[1] if len(sys.argv) > 1 else "egnn_equihnns_c256"
case = load_case(name)
method = str(case["meta_method"]); hid = int(case["meta_hidden"])
ref = O.MODELS[method](1, golden_args(method, hid)); fill_state_dict(ref, int(case["meta_seed"])); ref = ref.double()
d64 = batch_from_case(case); d64.pos = d64.pos.double(); d64.y = d64.y.double()
t64 = {}
out64 = ref(d64, taps=t64)
for v in t64.values(): v.retain_grad()
torch.nn.functional.mse_loss(out64, d64.y).backward()
mine = models.MODELS[method](1, golden_args(method, hid)); fill_state_dict(mine, int(case["meta_seed"])); mine.cuda()
d = batch_from_case(case).to("cuda")
tm = {}
out = mine(d, taps=tm)
for v in tm.values(): v.retain_grad()
torch.nn.functional.mse_loss(out, d.y).backward()
for k in t64:
    a, b = tm[k].grad.cpu().double(), t64[k].grad
    print("d/d tap %-14s rel err %.2e  (max %.2e)" % (k, (a-b).abs().max()/b.abs().max(), b.abs().max()))
g64 = dict(ref.named_parameters())
for n, p in mine.named_parameters():
    if g64[n].grad is None: continue
    a, b = p.grad.cpu().double(), g64[n].grad
    e = (a-b).abs().max()/b.abs().max()
    if e > 2e-5: print("param %-40s rel-own-max err %.2e (max %.2e)" % (n, e, b.abs().max()))
print("----")
for k in ("conv1", "conv0", "front_end"):
    a, b = tm[k].grad.cpu().double(), t64[k].grad
    err = (a-b).abs().max(-1).values
    bad = torch.nonzero(err > 1e-4 * b.abs().max()).flatten()
    print(k, "bad rows", bad.tolist()[:40], "of", a.shape[0])
vv = case["in_edge_index0"]; deg = np.bincount(vv, minlength=case["in_x"].shape[0])
print("deg of nodes", deg.tolist())
print("batch", case["in_batch"].tolist())
