#!/bin/bash
# first triangulation run: classification of the wrong pixels of the delay reproducer, with and without the diagnostics buffer; two edits of it
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/f2lab
for v in repro prewait b32bias; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so timeout 600 python3 tools/f2lab/classify.py 2 1 > gpurun_out/f2lab/classify_$v.txt 2>&1
  tail -1 gpurun_out/f2lab/classify_$v.txt
done
MNV_LIB_PATH=$PWD/variants/libmnv_repro.so F2_NO_DIAG=1 timeout 600 python3 tools/f2lab/classify.py 2 1 > gpurun_out/f2lab/classify_repro_nodiag.txt 2>&1
tail -1 gpurun_out/f2lab/classify_repro_nodiag.txt
