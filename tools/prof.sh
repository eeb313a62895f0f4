#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run).
# Usage (via gpurun): bash tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --frame-streams 0 $*"   # --frame-streams 0: no secondary per-frame launches of the same kernel name in the profile
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1
pmc() { # name counters...
  local name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > "$OUT/pmc_$name.log" 2>&1
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM
pmc tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pmc wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
