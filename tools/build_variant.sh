#!/bin/bash
# A/B build of the tuned kernel: compiles csrc/mnv_march_accel.hip with extra flags into gpurun_out/variants/libmnv_<tag>.so
# (all other objects are the regular build's).  Select it with MNV_LIB_PATH=<that file>.
# usage: tools/build_variant.sh <tag> '<extra hipcc flags>'
set -e
TAG=$1; EXTRA=$2
cd "$(dirname "$0")/../mega-nerf-viewer_amd"
OUT=../variants; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c csrc/mnv_march_accel.hip -o $OUT/accel_$TAG.o
OBJS=$(ls csrc/*.o host/*.o | grep -v "mnv_march_accel.o\|host/main.o")
/opt/rocm/bin/hipcc -shared -o $OUT/libmnv_$TAG.so $OBJS $OUT/accel_$TAG.o -lz -lpthread -ldl
echo built $OUT/libmnv_$TAG.so
