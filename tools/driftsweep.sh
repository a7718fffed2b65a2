#!/bin/bash
# Fixed refresh intervals and the automatic pacing against scenes of different speed.
#   [INTERVALS="0 3 6 12 24 48 -1"] [VELS="0 1 10 40 100"] bash tools/driftsweep.sh [bench args...]
set -e
mkdir -p gpurun_out
for vel in ${VELS:-0 1 10 40 100}; do
  for iv in ${INTERVALS:-0 3 6 12 24 48 -1}; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --fused-steps 0 --particles-per-gpu 67108864 --vel-extent $vel --reorder-interval $iv --steps 192 --warmup 96 "$@" > gpurun_out/drift.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/drift.json')); print('vel_extent %5s interval %3s: %.4f ms/step, %d refreshes in 192 steps'%('$vel', '$iv', d['ms_per_step'], d['config']['reorder_passes_in_timed_region']))"
  done
done
