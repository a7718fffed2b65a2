#!/bin/bash
# Which point of the refresh interval the tile order should be aimed at (variant bits 12..15, tenths).
set -e
for aim in 3 4 5 6 7 8; do
  v=$((aim << 12))
  timeout -k 10 120 python bench.py --no-cpu-baseline --fused-steps 0 --variant $v --steps 128 --warmup 32 > gpurun_out/aim_${aim}.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/aim_${aim}.json')); k=d['roofline']['launches_by_kind']; r=d['roofline']; print('aim 0.%d'%$aim, 'ms/step %.4f'%d['ms_per_step'], 'plain %.4f (%.1f %%)'%(k['step_stream_kernel']['avg_ms'], r['frac']*100), 'count %.3f'%(k['step_stream_kernel+count']['avg_ms'] or 0), 'scatter %.3f'%(k['step_scatter_kernel']['avg_ms'] or 0), 'region %.1f %%'%(r['whole_region']['frac']*100))"
done
