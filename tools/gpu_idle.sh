#!/bin/bash
# usage (GPU box, repo root): tools/gpu_idle.sh <tag> [bench.py args...]
# kernel trace of the bench with its streams as they run -> union of kernel intervals per step window: how much of the wall time has
# NO kernel running (host-bound / synchronisation gaps) and the average number of kernels in flight
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d "$out/prof" -o run -- python3 bench.py --no-cpu-baseline --single-variant --no-probe "$@" > "$out/bench.log" 2>&1
f=$(find "$out/prof" -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee "$out/idle.txt"
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - 600_000_000]          # the last 0.6 s: steady-state timed steps
t0, t1 = win[0][0], max(r[1] for r in win)
busy = 0; cur_s, cur_e = win[0][0], win[0][1]; last_name = win[0][2]
gaps = []
for s, e, _ in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e, last_name, _)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = _
busy += cur_e - cur_s
tot = t1 - t0
ksum = sum(e - s for s, e, _ in win)
print(f"window {tot / 1e6:.1f} ms, {len(win)} kernels; some kernel running {100.0 * busy / tot:.1f} % of the time; sum of kernel durations / wall = {ksum / tot:.2f}")
gaps.sort(reverse=True)
for g in gaps[:40] if False else sorted(gaps, reverse=True)[:24]:
    print(f"  gap {g[0] / 1e3:8.1f} us at +{(g[1] - t0) / 1e6:8.2f} ms  after {g[2][:70]}  before {g[3][:70]}")
print("largest idle gaps (us):", [round(g[0] / 1e3, 1) for g in gaps[:12]], " total idle", round((tot - busy) / 1e6, 2), "ms; gaps > 20 us:", sum(1 for g in gaps if g[0] > 20000))
PY
rm -rf "$out/prof"
