#!/bin/bash
# long runs for the record: 10,080 frames of the producer / consumer kernel compared bit for bit with the four-step path, 3,200 with tracker rows
# and visit marks compared as well, 600 refinement frames with prunes (tree links + packed accel against the walking kernel after every frame)
cd "$(dirname "$0")/.."; R=${ROUND:-r04}; mkdir -p gpurun_out/$R
timeout 2400 python3 tools/fused_stress.py 630 2 > gpurun_out/$R/soak_stress_10080.txt 2>&1; tail -1 gpurun_out/$R/soak_stress_10080.txt | cut -c1-300
timeout 1800 python3 tools/fused_stress.py 200 2 track > gpurun_out/$R/soak_stress_track_3200.txt 2>&1; tail -1 gpurun_out/$R/soak_stress_track_3200.txt | cut -c1-300
timeout 1500 python3 tools/refine_soak.py 600 > gpurun_out/$R/soak_refine_600.txt 2>&1; tail -2 gpurun_out/$R/soak_refine_600.txt | cut -c1-400
