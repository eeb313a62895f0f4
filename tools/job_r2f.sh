bash tools/prof.sh r02_batch > /dev/null 2>&1
python3 tools/make_traffic_json.py gpurun_out/r02_batch/summary.txt 64 > gpurun_out/r02_batch/traffic.json
bash tools/prof_trace.sh r02_per_frame --per-frame --frame-streams 3 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh r02_per_frame_1stream --per-frame --frame-streams 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh r02_cfg3 --workload cfg3 --frame-streams 0 --steps 4 --warmup 1 > /dev/null 2>&1
bash tools/prof_trace.sh r02_cfg4 --workload cfg4 --frame-streams 0 --steps 3 --warmup 1 > /dev/null 2>&1
head -30 gpurun_out/r02_batch/summary.txt | cut -c1-250; cat gpurun_out/r02_batch/traffic.json; tail -n 5 gpurun_out/r02_per_frame/summary.txt | cut -c1-600
