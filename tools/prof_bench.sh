#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [bench.py args...]
# rocprofv3 kernel trace + stats of one bench.py run; keeps only the per-kernel stats CSV under gpurun_out/<tag>/
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o run -- python3 bench.py "$@" > "$out/bench.log" 2>&1
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/prof"
grep '^{' "$out/bench.log" > "$out/bench_line.json" || true
head -30 "$out/kernel_stats.csv" | cut -c1-200
