#!/bin/bash
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/f2lab
for m in A B C; do
  MNV_LIB_PATH=$PWD/variants/libmnv_repro.so timeout 900 python3 tools/f2lab/constnet.py $m 4 2 > gpurun_out/f2lab/constnet_$m.txt 2>&1
  tail -c 1500 gpurun_out/f2lab/constnet_$m.txt; echo
done
