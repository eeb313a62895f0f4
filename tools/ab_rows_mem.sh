#!/bin/bash
# Measurement: colour rows in uncached / fine-grained device memory -- does the L2 then request 64 bytes instead of a 128-byte line per row?
# Test-hook build only.   usage (via gpurun): bash tools/ab_rows_mem.sh <tag> [workload]
set -u
TAG=${1:-rows_mem}; WL=${2:-cfg3}
export TMPDIR=/tmp
export MNV_LIB_PATH=$(cd "$(dirname "$0")/.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
ARGS="--workload $WL --laps 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --frame-streams 0"
for M in 0 1 2; do
  MNV_ROWS_MEM=$M timeout 300 python3 bench.py --workload $WL --laps 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --frame-streams 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows_mem $M:', d['value'], 'Mrays/s', d['roofline']['avg_launch_ms'], 'ms')"
  MNV_ROWS_MEM=$M timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc_rdreq_$M" -- python3 bench.py $ARGS > "$OUT/pmc_rdreq_$M.log" 2>&1
  MNV_ROWS_MEM=$M timeout 300 rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum --output-format csv -d "$OUT/pmc_miss_$M" -- python3 bench.py $ARGS > "$OUT/pmc_miss_$M.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_*/"))):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "march" in row.get("Kernel_Name", ""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(os.path.basename(d.rstrip("/")), {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
