#!/bin/bash
# A/B compiler flags for the march kernel (rebuilds on the GPU box)
run() { echo -n "[$1]: "; touch mega-nerf-viewer_amd/csrc/mnv_march_accel.hip; make -C mega-nerf-viewer_amd -j8 EXTRA="$1" > /tmp/mk.log 2>&1 || { echo build failed; tail -3 /tmp/mk.log; return; }; python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"; }
run ""
run "-fno-slp-vectorize"
run "-mllvm -amdgpu-early-inline-all=true"
run "-mllvm -amdgpu-sroa=1 -mllvm -unroll-threshold=50"
run "-O2"
run "-mllvm -amdgpu-schedule-metric-bias=0"
run ""
