#!/bin/bash
# the knobs below exist in the test-hook build of the library only (csrc/mnv_knobs.h)
export MNV_LIB_PATH=${MNV_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so}
# cfg3 is bound by real HBM traffic: does a coarser second lookup grid (fits the MALL, more node loads per deep step) help THERE?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
for L in 9 8 7; do
  for wl in cfg3 cfg4; do
    MNV_GRID2_LEVEL=$L timeout 600 python3 bench.py --workload $wl --laps 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --frame-streams 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('grid2 level $L', '$wl', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')"
  done
done
