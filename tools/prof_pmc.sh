#!/bin/bash
# usage (GPU box, repo root): tools/prof_pmc.sh <tag> [bench.py args...]
# three separate rocprofv3 counter passes (kernel trace only, per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
# of the same bench.py command -> gpurun_out/<tag>/pmc_{fetch,write,mfma}.csv
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pass in fetch write mfma; do
  case $pass in
    fetch) ctr="FETCH_SIZE";;
    write) ctr="WRITE_SIZE";;
    mfma) ctr="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
  esac
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/p_$pass" -o run -- python3 bench.py "$@" > "$out/pmc_$pass.log" 2>&1
  f=$(find "$out/p_$pass" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$out/pmc_$pass.csv" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]; c = r["Counter_Name"]
    tot[k][c] += float(r["Counter_Value"]); n[k][c] += 1
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Counter_Name", "Launches", "Sum", "Per_Launch"])
for k in tot:
    for c in tot[k]:
        w.writerow([k, c, n[k][c], tot[k][c], tot[k][c] / n[k][c]])
PY
  rm -rf "$out/p_$pass"
done
ls -la "$out"
