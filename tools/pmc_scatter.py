"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per launch for the scatter kernels.

usage: python tools/pmc_scatter.py COUNTER DIR   (prints `COUNTER kernel launches N mean_per_launch X` lines)
Run once per counter (separate rocprofv3 passes, MI355X_MICROARCH.md HBM section); FETCH_SIZE / WRITE_SIZE are in KiB."""
import csv
import glob
import sys

KERNELS = ("k_gather_ln_fwd", "k_gather_ln_bwd", "k_inc_fwd_col", "k_inc_fwd", "k_inc_bwd_both", "k_segment_reduce")


def main():
    counter, root = sys.argv[1], sys.argv[2]
    acc = {}
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                key = next((k for k in KERNELS if k + "<" in name or k + "(" in name), None)
                if key is None:
                    continue
                d = acc.setdefault(key, [0, 0.0])
                d[0] += 1
                d[1] += float(row["Counter_Value"])
    for k, (n, tot) in sorted(acc.items()):
        print(f"{counter} {k} launches {n} mean_per_launch {tot / n}")


if __name__ == "__main__":
    main()
