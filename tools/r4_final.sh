#!/bin/bash
# end-of-round measurement batch on ONE box: the bench line, the kernel stats of a single-stream profiled run, the three PMC passes
t=${1:-r4z}
mkdir -p gpurun_out/$t
timeout -k 10 900 python bench.py > gpurun_out/$t/bench.log 2>&1 || { tail -5 gpurun_out/$t/bench.log; exit 1; }
grep '^{' gpurun_out/$t/bench.log | tail -1 > gpurun_out/$t/bench_line.json
timeout -k 10 600 bash tools/prof_bench.sh $t/prof --no-cpu-baseline --single-variant --no-side-stream --steps 10 --warmup 3 > gpurun_out/$t/prof.log 2>&1 || { tail -5 gpurun_out/$t/prof.log; exit 1; }
timeout -k 10 900 bash tools/prof_pmc.sh $t/pmc --steps 2 --warmup 1 --no-cpu-baseline --no-probe --single-variant --no-side-stream > gpurun_out/$t/pmc.log 2>&1 || { tail -5 gpurun_out/$t/pmc.log; exit 1; }
cut -c1-600 gpurun_out/$t/bench_line.json
