#!/bin/bash
# hardware queues per process (ROCm: GPU_MAX_HW_QUEUES, default 4) against the step's seven or more streams, with and without an RCCL communicator
mkdir -p gpurun_out/r4hwq
for rep in 1 2; do for q in 3 4 5 6; do for dp in "--force-dp"; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py $dp --single-variant --no-cpu-baseline --no-probe --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q', '$dp'.ljust(10), d['value'], 'utt/s', d['ms_per_step'], 'ms')"
done; done; done
