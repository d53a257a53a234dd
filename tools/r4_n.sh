#!/bin/bash
# kernel-level timing of the dW microbench (k-major 8-phase form vs 128 x 128 kernel)
mkdir -p gpurun_out/r4n
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  AVAMD_GEMM_KM8=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o tn -- python3 $GRAFT_REPO_ROOT/tools/tn_microbench.py > $GRAFT_REPO_ROOT/gpurun_out/r4n/run_$v.txt 2>&1 || exit 1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r4n/stats_$v.csv
  g=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$g" > $GRAFT_REPO_ROOT/gpurun_out/r4n/trace_$v.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# group by (kernel, grid) -> durations
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "gemm" in n or "sum_slices" in n:
        key = (n[:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Z", ""))
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in d.items():
    v.sort()
    print(f"{k[0]:72s} grid {k[1]:>8s} z {k[2]:>4s} n {len(v):4d} median {v[len(v)//2]:8.1f} us")
PY
done
cat $GRAFT_REPO_ROOT/gpurun_out/r4n/trace_1.txt; echo; cat $GRAFT_REPO_ROOT/gpurun_out/r4n/trace_0.txt
