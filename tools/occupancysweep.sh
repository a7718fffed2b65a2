#!/bin/bash
# Plain step kernel by block size and resident blocks per CU (tuning bits 24-30 of the variant word).
#   bash tools/occupancysweep.sh [bench args...]
set -u
run() {  # label, variant
  timeout -k 10 120 python bench.py --steps 48 --warmup 12 --no-cpu-baseline --fused-steps 0 --variant $2 "${EXTRA[@]}" > /tmp/occ.json 2>/tmp/occ.err || { echo "variant $2 failed"; tail -n 3 /tmp/occ.err; exit 1; }
  python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/occ.json"))
k = {x["kernel"]: x["avg_ms"] for x in d["roofline"]["kernels"]}
f = lambda v: "   —  " if v is None else f"{v:.4f}"
print(f"{sys.argv[1]:40s} plain {f(k['step_stream_kernel'])} ms  count {f(k['step_stream_kernel+count'])}  scatter {f(k['step_scatter_kernel'])}  whole {d['ms_per_step']:.4f} ms/step", flush=True)
PY
}
EXTRA=("$@")
# block codes 0..3 = 512, 256, 128, 64 threads (the library instantiates no other size; larger codes mean 64)
sizes=(512 256 128 64)
run "the library's own choice" 0
for spec in "0 1" "0 2" "0 3" "1 2" "1 3" "1 4" "1 5" "1 6" "2 4" "2 6" "2 8" "2 10" "2 12" "3 8" "3 12"; do
  set -- $spec
  run "block ${sizes[$1]}, $2 per CU = $(( ${sizes[$1]} / 64 * $2 )) waves" $(( ($1 << 28) | ($2 << 24) ))
done
run "the library's own choice" 0
