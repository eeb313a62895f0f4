#!/bin/bash
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/f2lab
for v in Q1 Q2 Q3 Q4; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so timeout 900 python3 tools/f2lab/probe2.py 16 6 1 > gpurun_out/f2lab/probe2_$v.txt 2>&1
  head -c 1800 gpurun_out/f2lab/probe2_$v.txt | tail -c 1700; echo
done
for v in A1 A2 A3 A4; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so F2_TAG=_$v timeout 900 python3 tools/f2lab/constnet.py C 4 2 > gpurun_out/f2lab/constnet_C_$v.txt 2>&1
  head -c 700 gpurun_out/f2lab/constnet_C_$v.txt | tail -c 640; echo
done
