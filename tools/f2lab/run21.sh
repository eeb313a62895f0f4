#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 1800 python3 -m pytest tests/test_renderer_refine_gpu.py tests/test_guided_fused_gpu.py tests/test_cli_gpu.py tests/test_scale_gpu.py -x -q -m gpu -k "not many_frames" > gpurun_out/r04/pytest_renderer.txt 2>&1; tail -4 gpurun_out/r04/pytest_renderer.txt
timeout 900 python3 bench.py > gpurun_out/r04/bench_n1_d.json 2> gpurun_out/r04/bench_n1_d.err; python3 -c "
import json;d=json.load(open('gpurun_out/r04/bench_n1_d.json'));print(d['value'], d['cfg5']['guided_ms_per_frame'], d['cfg5']['both_ms_per_frame'], d['cfg5']['pixels_not_bit_identical_vs_four_step'])"
