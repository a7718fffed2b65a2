#!/usr/bin/env python3
"""Summarises the rocprofv3 counter passes written by tools/pmc_traffic.sh.

    python tools/pmc_summary.py <outdir>

For every kernel: number of dispatches and, per counter, the median over its
dispatches (the first sort of random input and the warm-up launches are outliers
the median ignores).  HBM bytes per launch follow the gfx950 rule: FETCH_SIZE
counts KiB with 128-byte reads tallied at 64 bytes, so reads = FETCH_SIZE x 1024 x 2;
writes = WRITE_SIZE x 1024.
"""
import collections
import csv
import glob
import statistics
import sys


def short(name):
    name = name.replace("void ", "").replace("pcl::", "")
    return name.split("(")[0]


def main(out):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{out}/*/*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(per):
        c = per[k]
        if not (k.startswith("step_") or k.startswith("reorder_")):
            continue
        n = max(len(v) for v in c.values())
        print(f"{k}  ({n} dispatches per pass)")
        med = {name: statistics.median(v) for name, v in c.items()}
        for name in sorted(med):
            print(f"    {name:28s} median {med[name]:.4e}   min {min(c[name]):.4e}   max {max(c[name]):.4e}")
        if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
            rd, wr = med["FETCH_SIZE"] * 1024 * 2, med["WRITE_SIZE"] * 1024
            print(f"    HBM bytes per launch: read {rd / 1e9:.3f} GB + written {wr / 1e9:.3f} GB = {(rd + wr) / 1e9:.3f} GB")


if __name__ == "__main__":
    main(sys.argv[1])
