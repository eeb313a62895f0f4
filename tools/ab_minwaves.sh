#!/bin/bash
# A/B the march kernel at different __launch_bounds__ min-waves (rebuilds on the GPU box).
for W in 6 7 8; do
  touch mega-nerf-viewer_amd/csrc/mnv_march_accel.hip
  make -C mega-nerf-viewer_amd -j8 EXTRA="-DMNV_MIN_WAVES=$W" > /dev/null 2>&1
  for B in 6 7 8; do
    [ $B -gt $W ] && continue
    echo -n "MIN_WAVES=$W BPC=$B: "
    MNV_BLOCKS_PER_CU=$B python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
  done
done
