#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 1500 python3 -m pytest tests/test_refine_gpu.py tests/test_renderer_refine_gpu.py tests/test_cli_gpu.py -x -q -m gpu > gpurun_out/r04/pytest_refine.txt 2>&1; tail -5 gpurun_out/r04/pytest_refine.txt
bash tools/refine_trace.sh > gpurun_out/r04/refine_trace.txt 2>&1; cat gpurun_out/r04/refine_trace.txt | cut -c1-200
