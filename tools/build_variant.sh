#!/bin/bash
# A/B build of the tuned kernel: compiles csrc/mnv_accel_march.hip (the march instantiations) with extra flags into gpurun_out/variants/libmnv_<tag>.so
# (all other objects are the regular build's).  Select it with MNV_LIB_PATH=<that file>.
# usage: tools/build_variant.sh <tag> '<extra hipcc flags>' [source file, default csrc/mnv_accel_march.hip]
set -e
TAG=$1; EXTRA=$2; SRC=${3:-csrc/mnv_accel_march.hip}; BASE=$(basename $SRC .hip)
cd "$(dirname "$0")/../mega-nerf-viewer_amd"
OUT=../variants; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -Xclang -target-feature -Xclang -packed-fp32-ops"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $SRC -o $OUT/${BASE}_$TAG.o
OBJS=$(ls csrc/*.o host/*.o | grep -v "$BASE.o\|host/main.o")
/opt/rocm/bin/hipcc -shared -o $OUT/libmnv_$TAG.so $OBJS $OUT/${BASE}_$TAG.o -lz -lpthread -ldl
echo built $OUT/libmnv_$TAG.so
