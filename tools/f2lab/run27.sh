#!/bin/bash
# item-based grid patch: the refresh / refinement tests, the frame's timeline, frame time
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
timeout 1500 python3 -m pytest $(grep -ln "accel_refresh\|refresh_accel\|Renderer" tests/test_*gpu*.py) -x -q > gpurun_out/r04/refresh_tests.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/r04/refresh_tests.log | tail -3
python3 tools/refine_frame_trace.py both 2>/dev/null | tail -1 | cut -c1-60
rm -rf gpurun_out/refine_trace; mkdir -p gpurun_out/refine_trace
bash tools/refine_trace.sh > gpurun_out/refine_trace/summary.txt 2>&1
python3 tools/refine_timeline.py gpurun_out/refine_trace/trace_both 8 > gpurun_out/refine_trace/timeline_both.txt 2>&1
grep -n "patch\|frame 8" gpurun_out/refine_trace/timeline_both.txt | cut -c1-200
