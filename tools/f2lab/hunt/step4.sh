#!/bin/bash
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/f2lab
for g in 1 2 4; do
  MNV_LIB_PATH=$PWD/variants/libmnv_repro.so F2_MAXG=$g F2_TAG=_g$g timeout 900 python3 tools/f2lab/constnet.py B 4 2 > gpurun_out/f2lab/constnet_B_g$g.txt 2>&1
  head -c 700 gpurun_out/f2lab/constnet_B_g$g.txt | tail -c 600; echo
done
