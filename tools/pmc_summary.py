#!/usr/bin/env python3
"""Summarises the rocprofv3 counter passes written by tools/pmc_traffic.sh.

    python tools/pmc_summary.py <outdir> [name prefix]

For every kernel (template arguments kept: the counting step is an instantiation of
the plain step kernel): number of dispatches and, per counter, the median over its
dispatches in dispatch order, skipping the first `SKIP_FIRST` of each kernel when it
has more than that (the first sort of random input and the warm-up launches move
more bytes than the steady state).  HBM bytes per launch follow the gfx950 rule:
FETCH_SIZE counts KiB with 128-byte reads tallied at 64 bytes, so
reads = FETCH_SIZE x 1024 x 2; writes = WRITE_SIZE x 1024.
"""
import collections
import csv
import glob
import statistics
import sys

SKIP_FIRST = 2


def short(name):
    name = name.replace("void ", "").replace("pcl::", "")
    return name.split("(")[0]


def main(out, prefixes=("step_", "reorder_")):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{out}/*/*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k in sorted(per):
        c = per[k]
        if not k.startswith(tuple(prefixes)):
            continue
        n = max(len(v) for v in c.values())
        print(f"{k}  ({n} dispatches per pass)")
        med = {}
        for name in sorted(c):
            vals = [v for _, v in sorted(c[name])]
            steady = vals[SKIP_FIRST:] if len(vals) > SKIP_FIRST else vals
            med[name] = statistics.median(steady)
            print(f"    {name:28s} steady median {med[name]:.4e}   min {min(vals):.4e}   max {max(vals):.4e}")
        if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
            rd, wr = med["FETCH_SIZE"] * 1024 * 2, med["WRITE_SIZE"] * 1024
            print(f"    HBM bytes per launch: read {rd / 1e9:.3f} GB + written {wr / 1e9:.3f} GB = {(rd + wr) / 1e9:.3f} GB"
                  f"  ({rd + wr:.0f} B)")


if __name__ == "__main__":
    main(sys.argv[1], tuple(sys.argv[2:]) or ("step_", "reorder_"))
