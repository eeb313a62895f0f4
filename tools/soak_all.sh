#!/bin/bash
# long runs for the record: 10,080 frames of the producer / consumer kernel compared bit for bit with the four-step path, 3,200 with tracker rows
# and visit marks compared as well, 600 refinement frames with prunes (tree links + packed accel against the walking kernel after every frame)
cd "$(dirname "$0")/.."; R=${ROUND:-r06}; mkdir -p gpurun_out/$R
timeout 2400 python3 tools/fused_stress.py 630 2 > gpurun_out/$R/soak_stress_10080.txt 2>&1; tail -1 gpurun_out/$R/soak_stress_10080.txt | cut -c1-300
timeout 1800 python3 tools/fused_stress.py 200 2 track > gpurun_out/$R/soak_stress_track_3200.txt 2>&1; tail -1 gpurun_out/$R/soak_stress_track_3200.txt | cut -c1-300
timeout 1500 python3 tools/refine_soak.py 600 > gpurun_out/$R/soak_refine_600.txt 2>&1; tail -2 gpurun_out/$R/soak_refine_600.txt | cut -c1-400
# the same loop on the test-hook build with every refresh / prune verifying every patched lookup word (grid, grid2, grid2_vox, inline cell words, brick
# records) against a fresh derivation -- a wrong word fails the call; and on a tree with inline words from the start that grows records on the way
HOOKS=$PWD/mega-nerf-viewer_amd/testhooks/libmnv.so
MNV_LIB_PATH=$HOOKS MNV_REFRESH_DEBUG=2 timeout 1500 python3 tools/refine_soak.py 600 > gpurun_out/$R/soak_refine_600_verified.txt 2> gpurun_out/$R/soak_refine_600_verified.err; tail -1 gpurun_out/$R/soak_refine_600_verified.txt | cut -c1-400; grep -c "mnv refresh" gpurun_out/$R/soak_refine_600_verified.err; grep -c "mnv verify" gpurun_out/$R/soak_refine_600_verified.err
MNV_LIB_PATH=$HOOKS MNV_REFRESH_DEBUG=2 timeout 1500 python3 tools/refine_soak.py 300 4000 shell_d7_sh9 > gpurun_out/$R/soak_refine_shell_verified.txt 2> gpurun_out/$R/soak_refine_shell_verified.err; tail -1 gpurun_out/$R/soak_refine_shell_verified.txt | cut -c1-400; grep -c "mnv refresh" gpurun_out/$R/soak_refine_shell_verified.err; grep -c "mnv verify" gpurun_out/$R/soak_refine_shell_verified.err
