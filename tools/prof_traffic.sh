#!/bin/bash
# HBM-traffic passes of the march kernel for one workload (the 5 GB tree of cfg3 / cfg4 takes ~25 s to build: only the passes the
# roofline needs, each in its own run, never together with a trace): kernel stats, FETCH_SIZE, WRITE_SIZE, the exact request counters,
# L2 hit / miss, L1->L2 requests + latency, the L1's stall cycles.   usage (via gpurun): bash tools/prof_traffic.sh <tag> [bench args...]
set -u
TAG=${1:-traffic}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --frame-streams 0 $*"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1
pmc() { local name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > "$OUT/pmc_$name.log" 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pmc tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pmc lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
pmc stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep "^{" "$OUT/trace.log" | tail -n 1 | cut -c1-1500 >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
