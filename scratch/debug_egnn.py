import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np, torch
from common import batch_from_case, fill_state_dict, golden_args, load_case
from equihgnn_amd import models, ops
from equihgnn_amd.index import HyperIndex
from oracle import ref_models as O
for name in ["egnn_equihnns_c64", "egnn_equihnns_c64_b"]:
    case = load_case(name)
    method = str(case["meta_method"]); hid = int(case["meta_hidden"])
    m = models.MODELS[method](1, golden_args(method, hid)); fill_state_dict(m, int(case["meta_seed"])); m.cuda()
    data = batch_from_case(case).to("cuda")
    taps = {}
    out = m(data, taps=taps)
    print(name, "out err", np.abs(out.detach().cpu().numpy() - case["out"]))
    for k, v in taps.items():
        ref = case["tap_" + k]
        err = np.abs(v.detach().cpu().numpy().reshape(ref.shape) - ref)
        rows = np.nonzero(err.max(-1) > 1e-4)[0]
        print(" tap", k, "max err", err.max(), "bad rows", rows[:20], "batch of bad rows", np.unique(case["in_batch"][rows]) if ref.shape[0]==case["in_batch"].shape[0] else "")
    ix = HyperIndex.from_batch(data)
    nbr, d2, _ = ix.knn(data.pos, 16, 0)
    a = np.sort(nbr.cpu().numpy(), -1); b = np.sort(case["knn_idx"], -1)
    bad = np.nonzero((a != b).any(-1))[0]
    print(" knn bad rows", bad)
    for r in bad[:5]:
        print("  row", r, "mine", nbr[r].cpu().numpy(), d2[r].cpu().numpy())
        print("       ref ", case["knn_idx"][r], case["knn_val"][r])
