"""Device time per aten op (with input shapes) of ONE eager forward+backward of a method, and the ops behind long reduce
kernels -- what is left for fusion after the hand-written kernels.  usage: python tools/op_profile.py METHOD BATCH FLAVOUR"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from equihgnn_amd.models import MODELS
from equihgnn_amd.batch import bucket_sizes, pad_batch, synth_batch
from equihgnn_amd.registry import default_args
from equihgnn_amd.trainer import GraphedTrainStep
method, B, fl = sys.argv[1], int(sys.argv[2]), sys.argv[3]
dev = "cuda:0"
ns = default_args(method=method, batch_size=B)
torch.manual_seed(0)
model = MODELS[method](1, ns).to(dev)
host = synth_batch(B, 2000, fl)
b = pad_batch(host, *bucket_sizes(host.num_nodes, host.num_hyperedges, host.nnz)).packed().to(dev)
b.num_real_graphs = B
tr = GraphedTrainStep(model, lr=1e-4)
tr.step(b)
tr._fwd_bwd(b); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr._fwd_bwd(b); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 0 and e.key.startswith("aten::"):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("aten self device time total us", tot)
for t, c, k, s in rows[:45]:
    print("%9.1f us %4d  %-32s %s" % (t, c, k, s))
print("---- long reduce kernels and their ops")
for ev in prof.events():
    for k in getattr(ev, "kernels", []) or []:
        if "reduce_kernel" in k.name and k.duration > 150:
            par = ev.cpu_parent
            chain = []
            while par is not None and len(chain) < 4:
                chain.append(par.name); par = par.cpu_parent
            print("%8.1f us  %s %s  <- %s" % (k.duration, ev.name, str(ev.input_shapes)[:80], " <- ".join(chain)))
