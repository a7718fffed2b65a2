set -e
for v in 96 128 160 192 224; do
  for iv in ${INTERVALS:-12 24}; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --variant $v --reorder-interval $iv --fused-steps 0 --steps 96 --warmup 32 > gpurun_out/ts_${v}_${iv}.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/ts_${v}_${iv}.json')); k=d['roofline']['launches_by_kind']; print('log2cells', $v>>4, 'interval', $iv, 'ms/step %.4f'%d['ms_per_step'], 'plain %.3f'%k['step_stream_kernel']['avg_ms'], 'count %.3f'%(k['step_stream_kernel+count']['avg_ms'] or 0), 'scatter %.3f'%(k['step_scatter_kernel']['avg_ms'] or 0))"
  done
done
