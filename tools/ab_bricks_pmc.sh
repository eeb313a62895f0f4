#!/bin/bash
# Where does the time of the brick-record kernel go against the node-word kernel?  Issue, wait and memory-pipeline counters of both.
# usage (via gpurun): bash tools/ab_bricks_pmc.sh [workload]
export MNV_LIB_PATH=${MNV_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so}
export TMPDIR=/tmp
WL=${1:-cfg4}
ARGS="--workload $WL --laps 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --frame-streams 0"
pass() {  # name, counters...
  local name=$1; shift
  for B in 2 0; do
    OUT=$PWD/gpurun_out/ab_bricks_pmc/${WL}_${name}_$B; mkdir -p $PWD/gpurun_out/ab_bricks_pmc
    MNV_BRICK_LEVELS=$B timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 bench.py $ARGS > "$OUT.log" 2>&1
    python3 - "$OUT" "$WL bricks $B $name" <<'PY'
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
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES
pass waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
pass ta TA_BUSY_sum TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum
