#!/bin/bash
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/f2lab
MNV_LIB_PATH=$PWD/variants/libmnv_P20.so timeout 900 python3 tools/f2lab/probe.py 20 12 0.001953125 > gpurun_out/f2lab/probe_P20.txt 2>&1
MNV_LIB_PATH=$PWD/variants/libmnv_P8.so timeout 900 python3 tools/f2lab/probe.py 8 12 0.001953125 > gpurun_out/f2lab/probe_P8.txt 2>&1
MNV_LIB_PATH=$PWD/variants/libmnv_P0.so timeout 900 python3 tools/f2lab/probe.py 0 8 0.03125 > gpurun_out/f2lab/probe_P0.txt 2>&1
for v in W1 W2 R2; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so F2_TAG=_$v timeout 900 python3 tools/f2lab/constnet.py B 4 2 > gpurun_out/f2lab/constnet_B_$v.txt 2>&1
  head -c 900 gpurun_out/f2lab/constnet_B_$v.txt | tail -c 800; echo
done
head -c 1500 gpurun_out/f2lab/probe_P20.txt
