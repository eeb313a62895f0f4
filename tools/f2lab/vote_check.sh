#!/bin/bash
# the knobs below exist in the test-hook build of the library only (csrc/mnv_knobs.h)
export MNV_LIB_PATH=${MNV_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so}
# the vote as a selection: tests, time per call on a real tracker frame (both paths), configs[4] frame, kernel list
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04; timeout 900 python3 -m pytest tests/test_refine_gpu.py tests/test_renderer_refine_gpu.py -x -q > gpurun_out/r04/vote_tests.log 2>&1; tail -4 gpurun_out/r04/vote_tests.log
python3 tools/f2lab/vote_time.py 2>/dev/null | tail -3
MNV_VOTE_FULL_SORT=1 python3 tools/f2lab/vote_time.py 2>/dev/null | tail -2
python3 tools/refine_frame_trace.py both 2>/dev/null | tail -1
MNV_VOTE_FULL_SORT=1 python3 tools/refine_frame_trace.py both 2>/dev/null | tail -1
export TMPDIR=/tmp; rm -rf gpurun_out/r04/vote_trace; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/vote_trace -- python3 tools/f2lab/vote_time.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04/vote_trace/*/*kernel_stats.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]: print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
