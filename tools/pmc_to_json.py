#!/usr/bin/env python3
"""profiles/rNN_pmc_scatter_workload.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) over `python3 bench.py --only-roofline`, plus that command's own JSON line (the
algorithmic bytes per launch).  bench.py reads the result (pmc_traffic) instead of carrying a constant.

    python tools/pmc_to_json.py FETCH_DIR WRITE_DIR ROOFLINE_JSON OUT_JSON --method egnn_equihnns --batch 256 --flavour qm9

hbm_bytes = 2 x FETCH_SIZE (gfx950: FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane read) + WRITE_SIZE."""
import argparse
import csv
import glob
import json

KERNELS = ("k_gather_ln_fwd", "k_gather_ln_bwd", "k_inc_fwd_col", "k_inc_fwd", "k_inc_bwd_both", "k_segment_reduce",
           "k_frame_hidden_fwd", "k_frame_hidden_bwd", "k_drop_mean_fwd", "k_drop_mean_bwd")


def per_kernel(root, counter):
    acc = {}
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = next((k for k in KERNELS if k + "<" in name or k + "(" in name), None)
            if key is not None:
                d = acc.setdefault(key, [0, 0.0])
                d[0] += 1
                d[1] += float(row["Counter_Value"])
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("roofline_json"); ap.add_argument("out")
    ap.add_argument("--method", default="egnn_equihnns"); ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--flavour", default="qm9")
    a = ap.parse_args()
    fetch, write = per_kernel(a.fetch_dir, "FETCH_SIZE"), per_kernel(a.write_dir, "WRITE_SIZE")
    line = [l for l in open(a.roofline_json) if l.startswith("{")][-1]
    roof = json.loads(line)["roofline"]["kernels"]
    out = {"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --only-roofline "
                      f"--method {a.method} --batch {a.batch} --flavour {a.flavour} (two separate passes), tools/pmc_to_json.py",
           "units": "FETCH_SIZE, WRITE_SIZE in KiB per launch (mean over the replayed launches); hbm_bytes = 2 x FETCH_SIZE "
                    "(gfx950 correction for 16 B/lane reads) + WRITE_SIZE",
           "workload": {"method": a.method, "batch": a.batch, "flavour": a.flavour}, "kernels": {}}
    tot_b = tot_n = 0
    for k, r in roof.items():
        if k not in fetch or k not in write:
            continue
        hbm = int((2 * fetch[k][1] + write[k][1]) * 1024)
        out["kernels"][k] = {"FETCH_SIZE_KiB": fetch[k][1], "WRITE_SIZE_KiB": write[k][1], "hbm_bytes_per_launch": hbm,
                             "algorithmic_bytes_per_launch": r["alg_bytes_per_launch"],
                             "ratio": round(hbm / r["alg_bytes_per_launch"], 3), "launches_per_step": r["launches_per_step"]}
        tot_b += hbm * r["launches_per_step"]
        tot_n += r["launches_per_step"]
    out["all_scatter_kernels"] = {"launches_per_step": tot_n, "hbm_bytes_per_launch": int(tot_b / max(tot_n, 1))}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out["all_scatter_kernels"]))


if __name__ == "__main__":
    main()
