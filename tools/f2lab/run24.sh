#!/bin/bash
cd "$(dirname "$0")/../.."
python3 tools/f2lab/vote_time.py 2>/dev/null | tail -3
MNV_VOTE_SORT=1 python3 tools/f2lab/vote_time.py 2>/dev/null | tail -2
export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/vote_trace -- python3 tools/f2lab/vote_time.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04/vote_trace/*/*kernel_stats.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]: print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
