#!/bin/bash
# Collects HBM traffic counters for the benchmark's kernels, one rocprofv3 pass per counter group
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box from the repository root:
#   bash tools/pmc_traffic.sh <outdir> [bench args...]
# 74 steps at the automatic refresh interval of 24: the first sort plus three steady refreshes.
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d "$R/$OUT/$tag" -- python3 "$R/bench.py" --steps 74 --warmup 2 --no-cpu-baseline --fused-steps 0 "$@" > "$R/$OUT/$tag.log" 2>&1 || { echo "pass $tag failed"; exit 1; }
  echo "pass $tag done"
done
echo done
