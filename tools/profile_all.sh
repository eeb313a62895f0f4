#!/bin/bash
# The committed profiles of a round: run on the GPU box (gpurun -- 'bash tools/profile_all.sh'), then tools/collect_profiles.py here.
bash tools/prof.sh r02_batch > /dev/null 2>&1
python3 tools/make_traffic_json.py gpurun_out/r02_batch/summary.txt 64 > gpurun_out/r02_batch/traffic.json
bash tools/prof_mem.sh r02_mem > gpurun_out/r02_mem.txt 2>&1
bash tools/prof_trace.sh r02_per_frame --per-frame --frame-streams 3 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh r02_per_frame_1stream --per-frame --frame-streams 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh r02_ref_layout --kernel ref_layout --per-frame --frame-streams 2 --laps 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh r02_cfg3 --workload cfg3 --frame-streams 0 --steps 4 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh r02_cfg4 --workload cfg4 --frame-streams 0 --steps 3 --warmup 1 > /dev/null 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_guided/trace -- python3 tools/guided_bench.py 32 4 > gpurun_out/r02_guided/trace.log 2>&1
python3 tools/prof_summary.py gpurun_out/r02_guided > gpurun_out/r02_guided/summary.txt 2>&1; grep "^{\"max" gpurun_out/r02_guided/trace.log >> gpurun_out/r02_guided/summary.txt
cp gpurun_out/r02_batch/traffic.json profiles/r02_traffic.json   # on the GPU box's copy: the bench line below then carries this run's traffic
python3 bench.py --steps 10 --warmup 2 > gpurun_out/r02_bench_n1.json 2> /dev/null
cat gpurun_out/r02_batch/traffic.json; tail -n 3 gpurun_out/r02_mem.txt; cat gpurun_out/r02_bench_n1.json | cut -c1-1200
