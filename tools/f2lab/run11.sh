#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -w tools/probes/pk_opsel_probe.hip -o /tmp/pk_opsel_probe && timeout 600 /tmp/pk_opsel_probe > gpurun_out/r04/probe_pk_opsel.txt 2>&1; grep -v " 0 0 0 0, wrong high results 0 0 0 0" gpurun_out/r04/probe_pk_opsel.txt | head -30
timeout 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "tree_cache or drop_in" > gpurun_out/r04/pytest_cache.txt 2>&1; tail -15 gpurun_out/r04/pytest_cache.txt
timeout 900 python3 bench.py --no-extras > gpurun_out/r04/bench_n1_c.json 2> gpurun_out/r04/bench_n1_c.err; python3 -c "
import json;d=json.load(open('gpurun_out/r04/bench_n1_c.json'));print(d['value']);print(d.get('ref_layout'))"; tail -3 gpurun_out/r04/bench_n1_c.err
