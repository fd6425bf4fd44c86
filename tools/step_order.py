#!/usr/bin/env python3
"""Kernel order of ONE replayed training step from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 bench.py --steps 6 --warmup 3 ...
    python3 tools/step_order.py DIR OUT.txt [--anchor k_csr_front]

Each line: start offset within the step (us), duration (us), idle gap before the kernel (us), grid threads / workgroup
size, kernel name.  The step is the span between the last two launches of the anchor kernel (the first kernel of a
step)."""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void at::native::", "at::", n)
    m = re.match(r"Cijk_(\w{4})_(\w{4}).*?(MT\d+x\d+x\d+)", n)
    if m:
        return f"GEMM {m.group(1)} {m.group(2)} {m.group(3)}"
    return n[:100]


def main():
    d, out = sys.argv[1], sys.argv[2]
    anchor = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--anchor" else "k_csr_front"
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 r.get("Grid_Size_X", "") + "/" + r.get("Workgroup_Size_X", "")) for r in rows)
    idx = [i for i, e in enumerate(ev) if anchor in e[2]]
    a, b = idx[-2], idx[-1]
    t0, busy = ev[a][0], 0
    with open(out, "w") as o:
        prev_end = t0
        for s, e, n, g in ev[a:b]:
            o.write(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {max(0, s - prev_end) / 1e3:6.1f} {g:>14s}  {short(n)}\n")
            busy += e - s
            prev_end = max(prev_end, e)
        span = ev[b][0] - t0
        o.write(f"# kernels {b - a}, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us\n")
    print(f"kernels in step: {b - a}, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main()
