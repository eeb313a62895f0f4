#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 1500 python3 -m pytest tests/test_comm_gpu.py tests/test_bench_multirank_gpu.py tests/test_multigpu_gloo.py -x -q -m gpu > gpurun_out/r04/pytest_multi.txt 2>&1; tail -6 gpurun_out/r04/pytest_multi.txt
: > gpurun_out/r04/root_emulation.jsonl
timeout 600 python3 tools/root_emulation.py --world 8 --root-periods 0,8 2>/dev/null | grep "^{" >> gpurun_out/r04/root_emulation.jsonl
timeout 600 python3 tools/root_emulation.py --world 4 --root-periods 0,16 2>/dev/null | grep "^{" >> gpurun_out/r04/root_emulation.jsonl
timeout 600 python3 tools/root_emulation.py --world 2 --root-periods 0,32 2>/dev/null | grep "^{" >> gpurun_out/r04/root_emulation.jsonl
cat gpurun_out/r04/root_emulation.jsonl
timeout 600 python3 tools/cfg5_rank_emulation.py > gpurun_out/r04/cfg5_rank_emulation.txt 2>&1; grep "^pose" gpurun_out/r04/cfg5_rank_emulation.txt
