#!/usr/bin/env python3
"""Per-kernel means over WORKING launches from a rocprofv3 kernel trace.

rocprofv3's own `--stats` averages every dispatch of a kernel.  An iteration graph of this library holds two LM
passes; when the first one ends the loop, the kernels of the second return at their first instruction (a few us), and
those early returns dilute the averages (k_chol_dataflow 211 us by --stats, 228 us per launch that
did the work).  This tool reads `*_kernel_trace.csv` (rocprofv3 --kernel-trace --output-format csv), drops for every
kernel the launches shorter than `--drop-below` (default 0.30) of that kernel's longest launch and reports the means of
the rest, next to the diluted all-launch means.

usage:  python tools/kernel_trace_stats.py <prof_kernel_trace.csv> [--out profiles/r04_kernel_stats_working.csv]
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short_name(name):
    """'void vmm::k_eval_both<true, double, false>(vmm::EvalArgs, ...)' -> 'k_eval_both<true, double, false>'"""
    n = name.strip()
    depth = 0
    for i, ch in enumerate(n):   # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    n = re.sub(r"^(void|int|double)\s+", "", n)
    return n.replace("vmm::", "")


def read_trace(path):
    per = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
                continue
            per[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return per


def stats(per, drop_below):
    out = []
    for name, d in per.items():
        mx = max(d)
        work = [x for x in d if x >= drop_below * mx]
        work.sort()
        out.append({
            "kernel": short_name(name),
            "launches": len(d),
            "working_launches": len(work),
            "mean_all_us": sum(d) / len(d) / 1e3,
            "mean_working_us": sum(work) / len(work) / 1e3,
            "median_working_us": work[len(work) // 2] / 1e3,
            "min_working_us": work[0] / 1e3,
            "max_us": mx / 1e3,
            "total_working_ms": sum(work) / 1e6,
        })
    out.sort(key=lambda r: -r["total_working_ms"])
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("trace")
    ap.add_argument("--drop-below", type=float, default=0.30,
                    help="a launch shorter than this fraction of the kernel's longest launch is an early return")
    ap.add_argument("--out", default=None, help="write the table as CSV here (default: stdout only)")
    ap.add_argument("--only", default=None, help="regular expression on the short kernel name")
    a = ap.parse_args()
    rows = stats(read_trace(a.trace), a.drop_below)
    if a.only:
        rows = [r for r in rows if re.search(a.only, r["kernel"])]
    cols = ["kernel", "launches", "working_launches", "mean_working_us", "median_working_us", "min_working_us", "max_us",
            "mean_all_us", "total_working_ms"]
    if a.out:
        with open(a.out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols)
            for r in rows:
                w.writerow([r[c] if isinstance(r[c], (str, int)) else "%.3f" % r[c] for c in cols])
    wname = max([len(r["kernel"]) for r in rows] + [6])
    print("%-*s %8s %8s %12s %12s %10s" % (wname, "kernel", "launches", "working", "mean_work_us", "mean_all_us", "total_ms"))
    for r in rows:
        print("%-*s %8d %8d %12.2f %12.2f %10.3f" % (wname, r["kernel"], r["launches"], r["working_launches"],
                                                  r["mean_working_us"], r["mean_all_us"], r["total_working_ms"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
