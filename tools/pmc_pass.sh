#!/bin/bash
# One rocprofv3 counter pass over a short benchmark run (on the GPU box, from the repository root):
#   bash tools/pmc_pass.sh <outdir> "<counters>" [bench args...]
# Prints one line per dispatch of the kernels that take part in the order refresh.
set -u
OUT=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d "$R/$OUT" -- python3 "$R/bench.py" --steps 26 --warmup 2 --no-cpu-baseline --fused-steps 0 "$@" > "$R/$OUT/run.log" 2>&1 || { echo "pass failed"; tail -5 "$R/$OUT/run.log"; exit 1; }
python3 "$R/tools/pmc_summary.py" "$R/$OUT" step_ | tail -6
