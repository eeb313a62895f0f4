#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 1500 python3 tools/fused_stress.py 63 2 > gpurun_out/r04/final_stress_1008.txt 2>&1; tail -1 gpurun_out/r04/final_stress_1008.txt | cut -c1-300
timeout 900 python3 tools/fused_stress.py 20 2 track > gpurun_out/r04/final_stress_track_320.txt 2>&1; tail -1 gpurun_out/r04/final_stress_track_320.txt | cut -c1-300
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r04/pytest_gpu.txt
ROUND=r04 bash tools/profile_all.sh > gpurun_out/r04/profile_all.log 2>&1; tail -c 1500 gpurun_out/r04/profile_all.log
