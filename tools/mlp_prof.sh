export TMPDIR=/tmp
OUT=$PWD/gpurun_out/mlp_prof; mkdir -p $OUT
python3 tools/mlp_prof.py 2>&1 | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/mlp_prof.py > $OUT/trace.log 2>&1
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$n -- python3 tools/mlp_prof.py > $OUT/pmc_$n.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/mlp_prof/pmc_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mlp_forward' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
for f in glob.glob('gpurun_out/mlp_prof/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'mlp' in r['Name'] or 'sort' in r['Name'].lower() or 'count' in r['Name'].lower(): print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
