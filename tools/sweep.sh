#!/bin/bash
# env sweeps of the march kernel on the GPU box: bash tools/sweep.sh
run() { echo -n "$*: "; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"; }
for L in 3 4; do for B in 4 6 8; do run MNV_LDS_LEVEL=$L MNV_BLOCKS_PER_CU=$B; done; done
run MNV_LDS_LEVEL=5 MNV_BLOCKS_PER_CU=1
for G in 0 7 8 9; do run MNV_GRID2_LEVEL=$G; done
