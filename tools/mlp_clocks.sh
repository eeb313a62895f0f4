# Where a wavefront of mlp_forward_kernel spends its time: builds csrc/mnv_mlp.hip with -DMNV_MLP_CLOCKS (s_memtime at the phase boundaries of
# wavefront 0 of every workgroup, summed; the shipped build has none of it) into variants/libmnv_clocks.so (variants/ is not tracked) and runs
# tools/mlp_prof.py on it.  Run from the repository root after `make` in mega-nerf-viewer_amd/:  bash tools/mlp_clocks.sh [w128|w64] [name=value ...]
set -e
cd mega-nerf-viewer_amd
mkdir -p ../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt \
    -Xclang -target-feature -Xclang -packed-fp32-ops -DMNV_MLP_CLOCKS -c csrc/mnv_mlp.hip -o ../variants/mlp_clocks.o 2> >(grep -v "packed-fp32-ops' is not a recognized feature" >&2)
OBJS=$(ls csrc/*.o host/*.o | grep -v "mnv_mlp.o\|mnv_comm.o\|mnv_knobs.o\|main.o")
/opt/rocm/bin/hipcc -shared -o ../variants/libmnv_clocks.so $OBJS ../variants/mlp_clocks.o testhooks/mnv_comm.o testhooks/mnv_knobs.o -lz -lpthread -ldl
cd ..
MNV_LIB_PATH=$PWD/variants/libmnv_clocks.so python3 tools/mlp_prof.py "$@" 2>&1 | grep -v amdgpu.ids | tail -8
