import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np, torch
from common import fill_state_dict
from equihgnn_amd import models
from equihgnn_amd.batch import synth_batch
from equihgnn_amd.registry import default_args
from oracle import ref_models as O
for method, bs, seed in [("mhnnm", 32, 1000), ("egnn_equihnns", 64, 2000)]:
    args = default_args(method=method)
    ref = O.MODELS[method](1, args); fill_state_dict(ref, seed)
    data = synth_batch(bs, seed)
    def run(m, d):
        for p in m.parameters(): p.grad = None
        out = m(d); torch.nn.functional.mse_loss(out, d.y).backward()
        return out.detach().cpu(), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    o_cpu, g_cpu = run(ref, data)
    import copy
    ref64 = copy.deepcopy(ref).double(); d64 = synth_batch(bs, seed); d64.pos = d64.pos.double(); d64.y = d64.y.double()
    o64, g64 = run(ref64, d64)
    refg = copy.deepcopy(ref).float().cuda()
    o_g, g_g = run(refg, data.to("cuda"))
    mine = models.MODELS[method](1, args); mine.load_state_dict(ref.state_dict()); mine.cuda()
    o_m, g_m = run(mine, data.to("cuda"))
    gmax = max(g.abs().max().item() for g in g64.values())
    print(method, "out err vs f64: cpu %.2e oracle-gpu %.2e mine %.2e" % ((o_cpu-o64).abs().max(), (o_g-o64).abs().max(), (o_m-o64).abs().max()))
    for n in g64:
        e = [(g[n].double()-g64[n]).abs().max().item()/gmax for g in (g_cpu, g_g, g_m)]
        if max(e) > 2e-6: print("  %-40s cpu %.1e oracle-gpu %.1e mine %.1e" % (n, *e))
