#!/bin/bash
# The benchmark configurations that fit one GPU, one bench.py run each; JSON lines into <outdir>.
#   bash tools/run_configs.sh gpurun_out/configs
set -e
OUT=$1
mkdir -p "$OUT"
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
f = d.get("fused_stepping")
print(f"{sys.argv[2]:34s} {d['ms_per_step']:8.4f} ms/step {d['value']/1e9:8.2f} Gp-steps/s  kernel {r['kernel_ms']:7.4f} ms {r['frac']*100:5.1f} %  region {r['whole_region']['frac']*100:5.1f} %"
      + (f"  fused {f['ms_per_step']:.4f} ms/step" if f else ""), flush=True)
PY
}
# configuration 2 steps in 0.15 ms: 100 steps end before the GPU has reached its clocks (plain kernel 0.153 ms in a
# 100-step run, 0.143 ms from about 40 ms of work on), hence the longer run, shown beside the default length
run cfg2_16M_256_uniform      --particles-per-gpu 16777216  --field 256 256 1   --pos-extent 20 --steps 480 --warmup 96
run cfg2_16M_256_cluster      --particles-per-gpu 16777216  --field 256 256 1   --pos-extent 2 --steps 480 --warmup 96
run cfg2_16M_256_uniform_100_steps --particles-per-gpu 16777216  --field 256 256 1   --pos-extent 20
run cfg3_256M_1024_uniform    --particles-per-gpu 268435456 --field 1024 1024 1 --pos-extent 20
run cfg3_256M_1024_cluster    --particles-per-gpu 268435456 --field 1024 1024 1 --pos-extent 2
run cfg4_share_128M_diag      --particles-per-gpu 134217728 --field 1024 1024 1 --pos-extent 20 --diagnostics on
run cfg5_64M_1024_edits_diag  --particles-per-gpu 67108864  --field 1024 1024 1 --pos-extent 20 --diagnostics on --edits
run cfg5_64M_256_edits_diag   --particles-per-gpu 67108864  --field 256 256 1   --pos-extent 20 --diagnostics on --edits
run default_scene_1000        --particles-per-gpu 1000      --field 10 10 10    --pos-extent 20 --steps 1000 --warmup 100
