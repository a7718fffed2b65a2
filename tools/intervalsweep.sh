#!/bin/bash
# Refresh-interval sweep (headline workload, or the bench arguments given): bash tools/intervalsweep.sh [bench args...]
set -e
for iv in ${INTERVALS:-6 8 10 12 14 16}; do
  steps=$((iv * 10)); warm=$((iv * 2))
  timeout -k 10 120 python bench.py --no-cpu-baseline --fused-steps 0 --reorder-interval $iv --steps $steps --warmup $warm "$@" > gpurun_out/iv_${iv}.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/iv_${iv}.json')); k=d['roofline']['launches_by_kind']; r=d['roofline']; print('interval', $iv, 'ms/step %.4f'%d['ms_per_step'], 'plain %.4f (%.1f %%)'%(k['step_stream_kernel']['avg_ms'], r['frac']*100), 'count %.3f'%(k['step_stream_kernel+count']['avg_ms'] or 0), 'scatter %.3f'%(k['step_scatter_kernel']['avg_ms'] or 0), 'region %.1f %%'%(r['whole_region']['frac']*100))"
done
