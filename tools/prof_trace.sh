#!/bin/bash
# kernel trace + stats only (no PMC passes): bash tools/prof_trace.sh <tag> [bench args...]
set -u
TAG=${1:-trace}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep "^{" "$OUT/trace.log" | tail -n 1 >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
