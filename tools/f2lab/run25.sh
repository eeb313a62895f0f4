#!/bin/bash
# fresh trace of a configs[4] frame with the current build + its timeline
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp
rm -rf gpurun_out/refine_trace; mkdir -p gpurun_out/refine_trace
python3 tools/refine_frame_trace.py both > gpurun_out/refine_trace/plain_both.log 2>&1; tail -1 gpurun_out/refine_trace/plain_both.log
bash tools/refine_trace.sh > gpurun_out/refine_trace/summary.txt 2>&1
python3 tools/refine_timeline.py gpurun_out/refine_trace/trace_both 8 > gpurun_out/refine_trace/timeline_both.txt 2>&1
tail -3 gpurun_out/refine_trace/timeline_both.txt
