#!/bin/bash
# A/B of two builds of the library on one box: abtmp/old.so and abtmp/new.so take turns as
# particles_amd/libparticles_hip.so (on the GPU box's scratch copy of the tree only).
#   bash tools/ab_libs.sh [bench args...]
set -u
for round in 1 2; do
  for which in old new; do
    cp abtmp/$which.so particles_amd/libparticles_hip.so
    timeout -k 10 120 python bench.py --steps 60 --warmup 12 --no-cpu-baseline --fused-steps 24 "$@" > /tmp/ab.json 2>/tmp/ab.err || { echo "$which failed"; tail -n 3 /tmp/ab.err; exit 1; }
    python - $which <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = {x["kernel"]: x["avg_ms"] for x in d["roofline"]["kernels"]}
f = lambda v: "   —  " if v is None else f"{v:.4f}"
print(f"{sys.argv[1]:4s} plain {f(k['step_stream_kernel'])} ms  count {f(k['step_stream_kernel+count'])}  scatter {f(k['step_scatter_kernel'])}  whole {d['ms_per_step']:.4f} ms/step  fused {(d.get('fused_stepping') or {}).get('ms_per_step')}", flush=True)
PY
  done
done
