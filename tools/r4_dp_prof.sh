#!/bin/bash
# kernel stats of the one-rank RCCL leg: what do the collectives cost on the device?
mkdir -p gpurun_out/r4dpp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4dpp/prof -o run -- python3 bench.py --force-dp --single-variant --no-cpu-baseline --no-probe --steps 8 --warmup 3 > gpurun_out/r4dpp/bench.log 2>&1 || exit 1
f=$(find gpurun_out/r4dpp/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4dpp/kernel_stats.csv
g=$(find gpurun_out/r4dpp/prof -name "*kernel_trace.csv" | head -1)
python3 - "$g" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "nccl" in n.lower() or "rccl" in n.lower() or "AllReduce" in n:
        d[n[:90]].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
for k, v in d.items():
    v.sort()
    print(f"{k:92s} n {len(v):4d} median {v[len(v)//2][0]:9.1f} us max {v[-1][0]:9.1f} us grid {v[0][1]} wg {v[0][2]}")
PY
rm -rf gpurun_out/r4dpp/prof
