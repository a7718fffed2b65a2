#!/bin/bash
# The round's evidence in one GPU call (from the repository root on the GPU box):
#   bash tools/profile_round.sh gpurun_out/r03
# 1. the driver's command plain, 2. the same command under rocprofv3 --kernel-trace --stats,
# 3. the counter passes for HBM traffic (headline workload, then configuration 2).
set -u
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
timeout -k 10 300 python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$R/$OUT/bench_driver_command.json" 2> "$R/$OUT/bench_driver_command.err" || { echo "bench failed"; exit 1; }
echo "bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kstats" -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$R/$OUT/bench_driver_command_under_rocprof.json" 2> "$R/$OUT/kstats.err" || { echo "kernel trace failed"; exit 1; }
echo "kernel trace done"
cd "$R"
bash tools/pmc_traffic.sh "$OUT/pmc_headline" || exit 1
python3 tools/pmc_summary.py "$OUT/pmc_headline" > "$OUT/pmc_traffic_headline.txt"
bash tools/pmc_traffic.sh "$OUT/pmc_cfg2" --particles-per-gpu 16777216 --field 256 256 1 || exit 1
python3 tools/pmc_summary.py "$OUT/pmc_cfg2" > "$OUT/pmc_traffic_cfg2.txt"
echo all done
