#!/bin/bash
# verification of the fix (library built without packed-FP32 instructions, layer-0 barrier removed):
#  1. the delay reproducer assembled from the NEW build: 16 poses x 63 launches = 1008 frames
#  2. the plain new build: 640 frames, + 320 frames with trackers / visit marks
#  3. the fused tests (oracle comparison included) and the guided-frame timing
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/f2lab
MNV_LIB_PATH=$PWD/variants/libmnv_newdelay.so timeout 1500 python3 tools/fused_stress.py 63 2 > gpurun_out/f2lab/verify_newdelay_1008.txt 2>&1; tail -1 gpurun_out/f2lab/verify_newdelay_1008.txt
MNV_LIB_PATH=$PWD/variants/libmnv_repro.so timeout 600 python3 tools/fused_stress.py 2 2 > gpurun_out/f2lab/verify_oldrepro_32.txt 2>&1; tail -1 gpurun_out/f2lab/verify_oldrepro_32.txt | cut -c1-300
timeout 1200 python3 tools/fused_stress.py 40 2 > gpurun_out/f2lab/verify_plain_640.txt 2>&1; tail -1 gpurun_out/f2lab/verify_plain_640.txt
timeout 1200 python3 tools/fused_stress.py 20 2 track > gpurun_out/f2lab/verify_plain_track_320.txt 2>&1; tail -1 gpurun_out/f2lab/verify_plain_track_320.txt
timeout 1500 python3 -m pytest tests/test_guided_fused_gpu.py -x -q -s -m gpu > gpurun_out/f2lab/verify_pytest_fused.txt 2>&1; tail -5 gpurun_out/f2lab/verify_pytest_fused.txt; grep "fused vs all-CPU" gpurun_out/f2lab/verify_pytest_fused.txt
timeout 600 python3 tools/guided_bench.py 32 4 > gpurun_out/f2lab/verify_guided_bench.txt 2>&1; tail -3 gpurun_out/f2lab/verify_guided_bench.txt
