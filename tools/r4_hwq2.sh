#!/bin/bash
# which stream pair must NOT run concurrently?  8 hardware queues (every stream on its own queue) with one stream at a time folded away
mkdir -p gpurun_out/r4hwq
run() { env "$@" timeout -k 10 300 python bench.py --single-variant --no-cpu-baseline --no-probe --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*'.ljust(80), d['value'], 'utt/s', d['ms_per_step'], 'ms')"; }
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=8 AVAMD_VISUAL_STREAMS=1
run GPU_MAX_HW_QUEUES=8 AVAMD_PASS_STREAMS=0
run GPU_MAX_HW_QUEUES=8 AVAMD_ATTN_DROPBITS=0
run GPU_MAX_HW_QUEUES=8 AVAMD_VISUAL_STREAMS=1 AVAMD_PASS_STREAMS=0
run GPU_MAX_HW_QUEUES=4 AVAMD_VISUAL_STREAMS=1
run GPU_MAX_HW_QUEUES=4 AVAMD_PASS_STREAMS=0
