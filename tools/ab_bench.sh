#!/bin/bash
# same-box A/B of one environment switch: tools/ab_bench.sh VAR [extra bench args]  -> gpurun_out/ab_VAR_{0,1}.log (+ per-shape GEMM tables)
V=$1; shift
mkdir -p gpurun_out/ab
for val in 1 0 1 0; do
  env $V=$val AVAMD_PROBE_SHAPES=gpurun_out/ab/${V}_${val}_shapes.txt timeout -k 10 300 python bench.py --no-cpu-baseline --single-variant "$@" > gpurun_out/ab/${V}_${val}.log 2>&1 || exit 1
  python - "$V" "$val" <<'PY'
import json, sys
l = [x for x in open(f"gpurun_out/ab/{sys.argv[1]}_{sys.argv[2]}.log") if x.startswith("{")][-1]
d = json.loads(l)
r = d.get("roofline") or {}
print(sys.argv[1], "=", sys.argv[2], d["value"], "utt/s", d["ms_per_step"], "ms; roofline", r.get("achieved"), "TF/s", r.get("avg_launch_us"), "us", flush=True)
PY
done
