#!/bin/bash
# final check of a source state: stress of the fused kernel, the whole GPU suite, the profiles of the round, a refinement frame's timeline
cd "$(dirname "$0")/.."
bash tools/final_check.sh
export TMPDIR=/tmp; rm -rf gpurun_out/refine_trace; mkdir -p gpurun_out/refine_trace
bash tools/refine_trace.sh > gpurun_out/refine_trace/summary.txt 2>&1
python3 tools/refine_timeline.py gpurun_out/refine_trace/trace_both 8 > gpurun_out/refine_trace/timeline_both.txt 2>&1
tail -1 gpurun_out/refine_trace/timeline_both.txt
