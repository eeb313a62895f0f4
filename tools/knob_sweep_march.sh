#!/bin/bash
# Launch-shape knobs of the march on the headline workload after a change of the kernel (test-hook build).  usage (via gpurun): bash tools/knob_sweep_march.sh
export MNV_LIB_PATH=$(cd "$(dirname "$0")/.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so
run() { echo -n "[$*]: "; env $* timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --frame-streams 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"; }
run MNV_BLOCKS_PER_CU=8
run MNV_BLOCKS_PER_CU=7
run MNV_BLOCKS_PER_CU=6
run MNV_LDS_LEVEL=2
run MNV_LDS_LEVEL=4
run MNV_TILE_WLOG=2
run MNV_TILE_WLOG=4
run MNV_GRID2_LEVEL=8
run MNV_BRICK_LEVELS=0
