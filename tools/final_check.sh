#!/bin/bash
# The gate of a source state of csrc/: 1008 + 320 stress frames of the producer / consumer kernel, the whole GPU suite, the profiles of the round
# (ROUND=rNN bash tools/profile_all.sh; afterwards: ROUND=rNN python3 tools/collect_profiles.py copies the summaries into profiles/).
cd "$(dirname "$0")/.."
R=${ROUND:-r04}; mkdir -p gpurun_out/$R
timeout 1500 python3 tools/fused_stress.py 63 2 > gpurun_out/$R/final_stress_1008.txt 2>&1; tail -1 gpurun_out/$R/final_stress_1008.txt | cut -c1-300
timeout 900 python3 tools/fused_stress.py 20 2 track > gpurun_out/$R/final_stress_track_320.txt 2>&1; tail -1 gpurun_out/$R/final_stress_track_320.txt | cut -c1-300
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/$R/pytest_gpu.txt 2>&1; tail -4 gpurun_out/$R/pytest_gpu.txt
ROUND=$R bash tools/profile_all.sh > gpurun_out/$R/profile_all.log 2>&1; tail -c 1500 gpurun_out/$R/profile_all.log
