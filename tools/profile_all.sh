#!/bin/bash
# The committed profiles of a round: run on the GPU box (gpurun -- 'bash tools/profile_all.sh'), then tools/collect_profiles.py here.
# ROUND=r03 (default) names the outputs.
R=${ROUND:-r06}
bash tools/prof.sh ${R}_batch > /dev/null 2>&1
python3 tools/make_traffic_json.py gpurun_out/${R}_batch/summary.txt 64 > gpurun_out/${R}_batch/traffic.json
bash tools/prof_mem.sh ${R}_mem > gpurun_out/${R}_mem.txt 2>&1
bash tools/prof_trace.sh ${R}_per_frame --per-frame --frame-streams 3 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh ${R}_per_frame_1stream --per-frame --frame-streams 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh ${R}_ref_layout --kernel ref_layout --per-frame --frame-streams 2 --laps 1 --steps 3 --warmup 1 > /dev/null 2>&1
# the reference's every-frame call: the tracker instantiation, one 1080p frame per launch
mkdir -p gpurun_out/${R}_tracker
TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_tracker/trace -- python3 tools/tracker_frame_time.py > gpurun_out/${R}_tracker/trace.log 2>&1
python3 tools/prof_summary.py gpurun_out/${R}_tracker > gpurun_out/${R}_tracker/summary.txt 2>&1; grep "^pose\|^col" gpurun_out/${R}_tracker/trace.log >> gpurun_out/${R}_tracker/summary.txt
# cfg3 / cfg4 (the 7.2 M-chunk tree) and fog (long dense runs): kernel stats AND the HBM-traffic passes; profiles/${R}_traffic_<wl>.json feed bench.py's cfg3 / cfg4_n1 / fog rooflines
for wl in cfg3 cfg4 fog; do
  # --laps 1: 16 frames per launch, the launch shape of the default bench line's cfg3 / cfg4_n1 objects
  bash tools/prof_traffic.sh ${R}_${wl} --workload $wl --laps 1 > /dev/null 2>&1
  python3 tools/make_traffic_json.py gpurun_out/${R}_${wl}/summary.txt 16 "--workload $wl --laps 1" > gpurun_out/${R}_${wl}/traffic.json
  cp gpurun_out/${R}_${wl}/traffic.json profiles/${R}_traffic_${wl}.json
done
export TMPDIR=/tmp
mkdir -p gpurun_out/${R}_guided
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_guided/trace -- python3 tools/guided_bench.py 32 4 > gpurun_out/${R}_guided/trace.log 2>&1
# counters of the fused kernels (separate passes, never with a trace): matrix-pipe busy cycles, wavefront cycles (-> wavefronts per SIMD), instruction mix
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/${R}_guided/pmc_$n -- python3 tools/guided_bench.py 32 4 > gpurun_out/${R}_guided/pmc_$n.log 2>&1
done
python3 tools/prof_summary.py gpurun_out/${R}_guided > gpurun_out/${R}_guided/summary.txt 2>&1; grep "^{\"max" gpurun_out/${R}_guided/trace.log >> gpurun_out/${R}_guided/summary.txt
python3 tools/guided_pmc_summary.py gpurun_out/${R}_guided >> gpurun_out/${R}_guided/summary.txt 2>&1
MNV_FUSED_DIAG=1 python3 tools/guided_bench.py 32 4 2> /dev/null | grep "^{" > gpurun_out/${R}_guided/phases.json; cat gpurun_out/${R}_guided/phases.json >> gpurun_out/${R}_guided/summary.txt
cp gpurun_out/${R}_batch/traffic.json profiles/${R}_traffic.json   # on the GPU box's copy: the bench line below then carries this run's traffic
python3 bench.py --steps 10 --warmup 2 > gpurun_out/${R}_bench_n1.json 2> /dev/null
cat gpurun_out/${R}_batch/traffic.json; tail -n 3 gpurun_out/${R}_mem.txt; cat gpurun_out/${R}_bench_n1.json | cut -c1-1200
