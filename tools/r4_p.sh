#!/bin/bash
# B0 fragments kept in registers (three-m-tile v7 instantiations): same-box A/B against a build with -DAV_V7_KEEPB0=0; then the bench A/B
mkdir -p gpurun_out/r4p
for rep in 1 2; do
  echo "== keep B0 (product build)"; timeout -k 10 300 python tools/epi_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4p/epi_keep_$rep.txt
  echo "== re-read B0 (-DAV_V7_KEEPB0=0)"; AVAMD_LIB=tools/_bin/libavhip_nokeep.so timeout -k 10 300 python tools/epi_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4p/epi_nokeep_$rep.txt
done
for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --single-variant > gpurun_out/r4p/bench_keep_$rep.log 2>&1 || exit 1
  AVAMD_LIB=tools/_bin/libavhip_nokeep.so timeout -k 10 300 python bench.py --no-cpu-baseline --single-variant > gpurun_out/r4p/bench_nokeep_$rep.log 2>&1 || exit 1
done
grep -h '^{' gpurun_out/r4p/bench_*.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print(d['value'], 'utt/s', d['ms_per_step'], 'ms; roofline', r.get('achieved'), 'TF/s')"
