#!/bin/bash
# VALU instruction mix of the march kernel (PMC), one launch of 16 frames
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-mix}
mkdir -p "$OUT"
ARGS="--steps 1 --warmup 1 --no-cpu-baseline"
pmc() { local name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > "$OUT/pmc_$name.log" 2>&1; }
pmc m1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
pmc m2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
pmc m3 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
python3 tools/prof_summary.py "$OUT" | grep pmc_
