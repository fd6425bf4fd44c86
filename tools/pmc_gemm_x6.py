#!/usr/bin/env python3
"""SQ counters of k_gemm_x6 on FAFormer's product ([245760 x 256] . [256 x 256]): matrix-pipe busy time, wave-parked / issue-stall /
active shares, LDS activity and bank conflicts, plus HBM bytes.  Driver of the rocprofv3 passes AND their summary:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d DIR_SQ -- python3 tools/pmc_gemm_x6.py run
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace ... -d DIR_G -- python3 tools/pmc_gemm_x6.py run
    rocprofv3 --pmc FETCH_SIZE ... -d DIR_F ...; rocprofv3 --pmc WRITE_SIZE ... -d DIR_W ...
    python3 tools/pmc_gemm_x6.py summarise DIR_SQ DIR_G DIR_F DIR_W OUT.json

(counters only with --kernel-trace; FETCH_SIZE and WRITE_SIZE in passes of their own: MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("fc x W^T", 245760, 256, 256, 0, 1), ("fc dY W", 245760, 256, 256, 0, 0)]


def run():
    import torch
    from equihgnn_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(0)
    for name, M, N, K, ta, tb in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device="cuda:0", generator=g)
        B = torch.randn((N, K) if tb else (K, N), device="cuda:0", generator=g)
        C = torch.empty(M, N, device="cuda:0")
        for _ in range(6):
            ops.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=C)
        torch.cuda.synchronize()


def collect(root):
    acc = {}
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "k_gemm_x6" not in row["Kernel_Name"]:
                continue
            key = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            d = acc.setdefault(key, {}).setdefault(row["Counter_Name"], [0, 0.0])
            d[0] += 1
            d[1] += float(row["Counter_Value"])
    return {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}


def summarise(dsq, dg, df, dw, out):
    sq, gr, fe, wr = collect(dsq), collect(dg), collect(df), collect(dw)
    res = {"what": "means per launch of k_gemm_x6 on [245760 x 256] . [256 x 256] (tile chosen by the cost model); SQ_WAVE_CYCLES / SQ_WAIT_* / "
                   "SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs, "
                   "GRBM_GUI_ACTIVE cycles summed over the 8 XCDs", "kernels": {}}
    for k, c in sq.items():
        gui = gr.get(k, {}).get("GRBM_GUI_ACTIVE")
        cyc = gui / 8 if gui else None
        d = {n: round(v) for n, v in c.items()}
        if cyc:
            d["kernel_cycles"] = round(cyc)
            d["mfma_busy_share_of_1024_simds"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
        w = c.get("SQ_WAVE_CYCLES") or 1.0
        d["wave_parked_share"] = round(c.get("SQ_WAIT_ANY", 0) / w, 4)
        d["issue_stall_share"] = round(c.get("SQ_WAIT_INST_ANY", 0) / w, 4)
        d["active_share"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share_of_lds_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        f, wv = fe.get(k, {}).get("FETCH_SIZE"), wr.get(k, {}).get("WRITE_SIZE")
        if f is not None and wv is not None:
            d["hbm_bytes_per_launch"] = int((2 * f + wv) * 1024)      # (gfx950: FETCH_SIZE tallies 64 B per 128-B request)
            d["algorithmic_bytes"] = 4 * (245760 * 256 * 2 + 256 * 256)
        res["kernels"][k] = d
        print(k, json.dumps(d))
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarise(*sys.argv[2:7])
