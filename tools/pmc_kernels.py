#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md
prescribes; counters only with --kernel-trace), for the kernels whose names contain one of the given substrings:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -- python3 bench.py ...
    python tools/pmc_kernels.py DIR_F DIR_W OUT.json k_rowgemm k_bn_ ...

hbm_bytes = 2 x FETCH_SIZE (gfx950: FETCH_SIZE tallies 64 B per 128-B request of a 16-B-per-lane read) + WRITE_SIZE, KiB -> bytes,
mean over the launches of each kernel (template arguments kept: the shapes differ)."""
import csv
import glob
import json
import re
import sys


def per_kernel(root, counter, subs):
    acc = {}
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            if not any(s in name for s in subs):
                continue
            key = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
            d = acc.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += float(row["Counter_Value"])
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    subs = sys.argv[4:]
    fetch, write = per_kernel(fdir, "FETCH_SIZE", subs), per_kernel(wdir, "WRITE_SIZE", subs)
    res = {"units": "KiB per launch (mean); hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", "kernels": {}}
    for k in sorted(fetch):
        if k not in write:
            continue
        res["kernels"][k] = {"launches": fetch[k][0], "FETCH_SIZE_KiB": round(fetch[k][1], 1), "WRITE_SIZE_KiB": round(write[k][1], 1),
                             "hbm_bytes_per_launch": int((2 * fetch[k][1] + write[k][1]) * 1024)}
        print(f"{k:60s} n={fetch[k][0]:4d} fetch {fetch[k][1] / 1024:8.1f} MiB x2  write {write[k][1] / 1024:8.1f} MiB  -> "
              f"{(2 * fetch[k][1] + write[k][1]) / 1024:8.1f} MiB")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
