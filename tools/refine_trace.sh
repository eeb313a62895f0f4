#!/bin/bash
# Kernel trace of refinement frames through VolumeRenderer::render on the cfg2 tree (tools/refine_frame_trace.py split|both):
# per-kernel microseconds per frame, which says what of a refinement frame is the march and what runs replicated on every rank.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out/refine_trace
for m in split both; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/refine_trace/trace_$m -- python3 tools/refine_frame_trace.py $m > gpurun_out/refine_trace/trace_$m.log 2>&1
  tail -1 gpurun_out/refine_trace/trace_$m.log
  python3 - $m <<'PY'
import csv, glob, sys
m = sys.argv[1]
f = glob.glob(f"gpurun_out/refine_trace/trace_{m}/*/*kernel_stats.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = 0.0
for r in rows[:24]:
    per = float(r["TotalDurationNs"]) / 14e3
    tot += per
    print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {per:8.1f} us/frame")
print(f"  sum of the 24 largest: {tot:.1f} us per frame")
PY
done
