#!/bin/bash
# the default bench line as the driver runs it (python bench.py), its wall time, and the fields a reader looks at first
cd "$(dirname "$0")/.."; R=${ROUND:-r04}; mkdir -p gpurun_out/$R
SECONDS=0
python3 bench.py > gpurun_out/$R/bench_final.json 2> gpurun_out/$R/bench_final.err
echo "rc $? wall ${SECONDS} s"
python3 - "$R" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/{sys.argv[1]}/bench_final.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"])
print(d["cfg3"]["value"], d["cfg3"]["roofline"]["traffic"], d["cfg4_n1"]["value"], d["cfg4_n1"]["roofline"]["frac"], d["cfg5"]["guided_ms_per_frame"], d["cfg5"]["both_ms_per_frame"])
PY
