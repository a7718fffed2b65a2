#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: one line per dispatch."""
import collections
import csv
import glob
import sys

paths = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for path in paths:
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if filt and filt not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        d = by.setdefault(k, {"name": r["Kernel_Name"][:60], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"],
                              "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"],
                              "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, d in by.items():
        meta = f"#{k} grid={d['grid']} wg={d['wg']} vgpr={d['vgpr']} sgpr={d['sgpr']} {d['us']:.1f}us {d['name']}"
        ctr = " ".join(f"{a}={b:.4g}" for a, b in d.items() if a not in ("name", "grid", "wg", "vgpr", "sgpr", "us"))
        print(meta, "|", ctr)
