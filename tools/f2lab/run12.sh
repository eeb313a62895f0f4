#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -w tools/probes/pk_opsel_probe.hip -o /tmp/pk_opsel_probe && timeout 600 /tmp/pk_opsel_probe > gpurun_out/r04/probe_pk_opsel.txt 2>&1; grep -v " 0 0 0 0, wrong high results 0 0 0 0" gpurun_out/r04/probe_pk_opsel.txt | head -30
