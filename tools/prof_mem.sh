#!/bin/bash
# Vector-memory path counters of the march kernel (TA busy, TCP stalls, L1->L2 read latency); usage via gpurun: bash tools/prof_mem.sh <tag>
set -u
TAG=${1:-mem}; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --frame-streams 0"
pmc() { local name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > "$OUT/pmc_$name.log" 2>&1; }
pmc ta TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
pmc lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pmc stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pmc stall2 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
python3 tools/prof_summary.py "$OUT" 2>&1 | grep "^pmc_" 
