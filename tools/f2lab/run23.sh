#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 1800 python3 -m pytest tests/test_refine_gpu.py tests/test_renderer_refine_gpu.py tests/test_cli_gpu.py -x -q -m gpu > gpurun_out/r04/pytest_vote.txt 2>&1; tail -4 gpurun_out/r04/pytest_vote.txt
MNV_VOTE_SORT=1 timeout 900 python3 -m pytest tests/test_refine_gpu.py -x -q -m gpu -k selection > gpurun_out/r04/pytest_vote_sort.txt 2>&1; tail -2 gpurun_out/r04/pytest_vote_sort.txt
for e in "" "MNV_VOTE_SORT=1"; do
  env $e python3 tools/refine_bench.py 2>/dev/null | tail -3 | cut -c1-400
done
timeout 900 python3 bench.py > gpurun_out/r04/bench_n1_e.json 2> gpurun_out/r04/bench_n1_e.err; python3 -c "
import json;d=json.load(open('gpurun_out/r04/bench_n1_e.json'));print(d['value'], d['cfg5']['guided_ms_per_frame'], d['cfg5']['both_ms_per_frame'], d['cfg5']['pixels_not_bit_identical_vs_four_step'])"
