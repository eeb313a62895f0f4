#!/bin/bash
# Build libmnv with a HAND-EDITED device listing of csrc/mnv_accel_march.hip: the way to change one instruction of a kernel without the
# compiler re-scheduling everything else (used to bisect the rare wrong denominator of guided_fused2_kernel, LAB_NOTEBOOK.md).
#   tools/asm_variant.sh prepare '<extra hipcc flags>'   -> /tmp/asmlab/device.s (edit a copy of it), host listing kept beside it
#   tools/asm_variant.sh build <tag> <edited device .s>  -> variants/libmnv_<tag>.so   (select with MNV_LIB_PATH)
set -e
LLVM=/opt/rocm/lib/llvm/bin; LAB=/tmp/asmlab; mkdir -p $LAB
cd "$(dirname "$0")/../mega-nerf-viewer_amd"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -Xclang -target-feature -Xclang -packed-fp32-ops"
if [ "$1" = prepare ]; then
  /opt/rocm/bin/hipcc $FLAGS $2 --save-temps=obj -c csrc/mnv_accel_march.hip -o $LAB/accel.o 2> /dev/null
  cp $LAB/mnv_accel_march-hip-amdgcn-amd-amdhsa-gfx950.s $LAB/device.s
  # the host listing with the device binary taken from a file instead of the embedded string
  python3 - <<'PY'
import re
p = "/tmp/asmlab/mnv_accel_march-host-x86_64-unknown-linux-gnu.s"
s = open(p, encoding="latin-1").read()
s = re.sub(r'\t\.asciz\t"__CLANG_OFFLOAD_BUNDLE__.*?\n\t\.size\t(\.L__unnamed_\d+), \d+\n', lambda m: '\t.incbin "/tmp/asmlab/device.hipfb"\n', s, count=1, flags=re.S)
open("/tmp/asmlab/host.s", "w", encoding="latin-1").write(s)
PY
  grep -c incbin $LAB/host.s; echo "prepared $LAB/device.s"
  exit 0
fi
TAG=$2; SRC=$3
$LLVM/clang -cc1as -triple amdgcn-amd-amdhsa -filetype obj -main-file-name mnv_accel_march.hip -target-cpu gfx950 -mrelocation-model pic -o $LAB/device.o $SRC
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols -plugin-opt=mcpu=gfx950 -o $LAB/device.out $LAB/device.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$LAB/device.out -output=$LAB/device.hipfb
$LLVM/clang -cc1as -triple x86_64-unknown-linux-gnu -filetype obj -main-file-name mnv_accel_march.hip -target-cpu x86-64 -mrelocation-model pic -o $LAB/accel_$TAG.o $LAB/host.s
mkdir -p ../variants
OBJS=$(ls csrc/*.o host/*.o | grep -v "mnv_accel_march.o\|host/main.o")
/opt/rocm/bin/hipcc -shared -o ../variants/libmnv_$TAG.so $OBJS $LAB/accel_$TAG.o -lz -lpthread -ldl
echo built ../variants/libmnv_$TAG.so
