#!/bin/bash
# the knobs below exist in the test-hook build of the library only (csrc/mnv_knobs.h)
export MNV_LIB_PATH=${MNV_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so}
# A/B of environment knobs of the march kernel on the GPU box; every run checks 2 poses against the oracle:  bash tools/ab_env.sh "A=1" "B=2 C=3" ...
run() { echo -n "[$*]: "; env $* python3 bench.py --steps 6 --warmup 2 --cpu-poses 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], 'bad pixels', d['parity']['pixels_not_bit_identical'])"; }
for e in "$@"; do run $e; done
