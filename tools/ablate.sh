#!/bin/bash
# the knobs below exist in the test-hook build of the library only (csrc/mnv_knobs.h)
export MNV_LIB_PATH=${MNV_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so}
# Ablations of the march kernel (diagnostics instantiation, counters off): MNV_ABLATE bits 1 = no colour evaluation, 2 = no dense samples, 4 = colour rows from 64 K cached rows (wrong colours)
run() { echo -n "$*: "; env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"; }
run MNV_ABLATE=8
run MNV_ABLATE=4
run MNV_ABLATE=1
run MNV_ABLATE=2
