#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "general_branching or invalid_arguments or tree_cache" > gpurun_out/r04/pytest_n.txt 2>&1; tail -8 gpurun_out/r04/pytest_n.txt
