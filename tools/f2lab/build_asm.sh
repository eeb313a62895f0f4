#!/bin/bash
# Assemble an edited device listing (see tools/asm_variant.sh prepare) into variants/libmnv_<tag>.so, with the host listing that belongs to it.
# usage: tools/f2lab/build_asm.sh <tag> <edited device .s> <host .s>
set -e
LLVM=/opt/rocm/lib/llvm/bin; LAB=/tmp/asmlab
TAG=$1; SRC=$2; HOST=$3
cd "$(dirname "$0")/../../mega-nerf-viewer_amd"
W=$LAB/w_$TAG; mkdir -p $W
$LLVM/clang -cc1as -triple amdgcn-amd-amdhsa -filetype obj -main-file-name mnv_accel_fused.hip -target-cpu gfx950 -mrelocation-model pic -o $W/device.o $SRC
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols -plugin-opt=mcpu=gfx950 -o $W/device.out $W/device.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/device.out -output=$W/device.hipfb
sed "s#/tmp/asmlab/device.hipfb#$W/device.hipfb#" $HOST > $W/host.s
$LLVM/clang -cc1as -triple x86_64-unknown-linux-gnu -filetype obj -main-file-name mnv_accel_fused.hip -target-cpu x86-64 -mrelocation-model pic -o $W/accel.o $W/host.s
mkdir -p ../variants
OBJS=$(ls csrc/*.o host/*.o | grep -v "mnv_accel_fused.o\|host/main.o")
/opt/rocm/bin/hipcc -shared -o ../variants/libmnv_$TAG.so $OBJS $W/accel.o -lz -lpthread -ldl
echo built ../variants/libmnv_$TAG.so
