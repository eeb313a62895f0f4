#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r04/pytest_gpu.txt
ROUND=r04 bash tools/profile_all.sh > gpurun_out/r04/profile_all.log 2>&1; tail -c 3000 gpurun_out/r04/profile_all.log
