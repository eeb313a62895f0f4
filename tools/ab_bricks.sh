#!/bin/bash
# A/B of the inline cell words and brick records below the second lookup grid (AccelView::grid2i, ::recs): MNV_BRICK_LEVELS=0 builds the accel without them (3: inline cell words + records, 1: inline words only).
# The knob exists in the test-hook build of the library only (csrc/mnv_knobs.h).   usage (via gpurun): bash tools/ab_bricks.sh [workloads...]
export MNV_LIB_PATH=${MNV_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so}
for wl in ${@:-cfg3 cfg4}; do
  for B in 3 0; do
    MNV_BRICK_LEVELS=$B timeout 900 python3 bench.py --workload $wl $( [ $wl = cfg2 ] && echo --laps 4 || echo --laps 1 ) --steps 5 --warmup 2 --no-cpu-baseline --no-extras --frame-streams 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl bricks $B:', d['value'], 'Mrays/s', d['roofline']['avg_launch_ms'], 'ms per launch')"
  done
done
# L2 misses of the same launches
export TMPDIR=/tmp
for wl in ${@:-cfg3 cfg4}; do
  for B in 3 0; do
    OUT=$PWD/gpurun_out/ab_bricks/pmc_${wl}_$B; mkdir -p $PWD/gpurun_out/ab_bricks
    MNV_BRICK_LEVELS=$B timeout 300 rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum --output-format csv -d "$OUT" -- python3 bench.py --workload $wl $( [ $wl = cfg2 ] && echo --laps 4 || echo --laps 1 ) --steps 3 --warmup 1 --no-cpu-baseline --no-extras --frame-streams 0 > "$OUT.log" 2>&1
    python3 - "$OUT" "$wl bricks $B" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
agg = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "march" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
  done
done
