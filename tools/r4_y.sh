#!/bin/bash
# native per-layer forward entry: configs[0] (launch-bound) and the headline configuration, AVAMD_W2V2_NATIVE=1/0 interleaved on one box
mkdir -p gpurun_out/r4n2
for rep in 1 2; do for v in 1 0; do
  AVAMD_W2V2_NATIVE=$v timeout -k 10 300 python bench.py --batch 2 --seconds 1 --no-cpu-baseline --no-probe --single-variant 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config 1 (2 x 1 s) native=$v', d['value'], 'utt/s', d['ms_per_step'], 'ms')"
done; done
for rep in 1 2; do for v in 1 0; do
  AVAMD_W2V2_NATIVE=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-probe --single-variant 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('batch 64 x 4 s native=$v', d['value'], 'utt/s', d['ms_per_step'], 'ms')"
done; done
