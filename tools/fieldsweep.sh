#!/bin/bash
# Step-kernel time against field size with the particles in caller (random) order and the field
# fetched through the caches (LDS staging off): how much of the step is the field fetch.
set -e
for f in 4 16 32 64 128 256; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --fused-steps 0 --reorder-interval 0 --variant 2 --field $f $f 1 --steps 40 --warmup 5 > gpurun_out/fs_${f}.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/fs_${f}.json')); r=d['roofline']; print('field %4d^2 (%7.1f KiB): plain kernel %.4f ms = %.1f %% of 8 TB/s'%($f, $f*$f*16/1024, r['kernel_ms'], r['frac']*100))"
done
