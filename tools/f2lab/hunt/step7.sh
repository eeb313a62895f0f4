#!/bin/bash
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/f2lab
for v in S1 S2 S3 S4; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so F2_TAG=_$v timeout 900 python3 tools/f2lab/constnet.py C 4 2 > gpurun_out/f2lab/constnet_C_$v.txt 2>&1
  echo $v; head -c 900 gpurun_out/f2lab/constnet_C_$v.txt | tail -c 840; echo
done
