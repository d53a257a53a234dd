#!/bin/bash
# does the persistent GEMM (static tile walk, one workgroup per CU) suffer when RCCL kernels share the device?  --force-dp (one rank, real RCCL group) with
# the persistent form on / off, interleaved on one box
mkdir -p gpurun_out/r4dp
for rep in 1 2; do for v in 1 0; do for dp in "--force-dp" ""; do
  AVAMD_GEMM_V7=$v timeout -k 10 300 python bench.py $dp --single-variant --no-cpu-baseline --no-probe --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('V7=$v', '$dp'.ljust(10), d['value'], 'utt/s', d['ms_per_step'], 'ms')"
done; done; done
