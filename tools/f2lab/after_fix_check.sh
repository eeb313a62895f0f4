#!/bin/bash
# after the fix: stand-alone probe of the packed add, the whole GPU suite, the default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -w tools/probes/pk_opsel_probe.hip -o /tmp/pk_opsel_probe && timeout 600 /tmp/pk_opsel_probe > gpurun_out/r04/probe_pk_opsel.txt 2>&1; grep -v " 0 0 0 0, wrong high results 0 0 0 0" gpurun_out/r04/probe_pk_opsel.txt | head -20; wc -l gpurun_out/r04/probe_pk_opsel.txt
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r04/pytest_gpu.txt
timeout 900 python3 bench.py > gpurun_out/r04/bench_n1_a.json 2> gpurun_out/r04/bench_n1_a.err; cut -c1-900 gpurun_out/r04/bench_n1_a.json
