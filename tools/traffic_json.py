#!/usr/bin/env python3
"""Regenerates profiles/traffic.json (the stored counter figures bench.py quotes in roofline.traffic) from the
summaries tools/pmc_summary.py wrote:   python tools/traffic_json.py profiles/rNN
reads <dir>/pmc_traffic_headline.txt (2^28 particles, 1024^2 field) and <dir>/pmc_traffic_cfg2.txt (2^24, 256^2)."""
import json
import os
import re
import sys

NAMES = {"step_stream_kernel<false, true, false, false, 256>": "step_stream_kernel",
         "step_stream_kernel<false, true, false, true, 512>": "step_stream_kernel+count",
         "step_scatter_kernel<false>": "step_scatter_kernel"}


def parse(path):
    total, fetch, write, cur = {}, {}, {}, None
    for line in open(path):
        if line.strip() and not line.startswith(" "):
            cur = line.split("  (")[0].strip()
        m = re.search(r"FETCH_SIZE\s+steady median ([0-9.e+]+)", line)
        if m:
            fetch[cur] = float(m.group(1))
        m = re.search(r"WRITE_SIZE\s+steady median ([0-9.e+]+)", line)
        if m:
            write[cur] = float(m.group(1))
        m = re.search(r"HBM bytes per launch.*\((\d+) B\)", line)
        if m:
            total[cur] = int(m.group(1))
    return total, fetch, write


def main(directory):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    target = os.path.join(root, "profiles", "traffic.json")
    d = json.load(open(target))
    for key, name in (("268435456|1024x1024x1|20.0", "pmc_traffic_headline.txt"), ("16777216|256x256x1|20.0", "pmc_traffic_cfg2.txt")):
        path = os.path.join(directory, name)
        total, fetch, write = parse(path)
        e = d.setdefault(key, {})
        e["kernels"] = {NAMES[k]: v for k, v in total.items() if k in NAMES}
        e["hbm_bytes"] = e["kernels"]["step_stream_kernel"]
        e["fetch_size_kib"] = {NAMES[k]: round(v) for k, v in fetch.items() if k in NAMES}
        e["write_size_kib"] = {NAMES[k]: round(v) for k, v in write.items() if k in NAMES}
        e["source"] = os.path.relpath(path, root)
        print(key, e["kernels"])
    json.dump(d, open(target, "w"), indent=1)


main(sys.argv[1])
