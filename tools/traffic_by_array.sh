#!/bin/bash
# Which array do the L2 misses (= requests to the fabric / HBM) of the march kernel belong to?  The counters cannot say, so every array
# in turn is read TWICE: a variant of the march (-DMNV_SHADOW_MASK=<bit>, the plain instantiation, nothing else changed) repeats each load
# of that array at the same index of a copy at other addresses -- frames stay right, and TCC_MISS grows by about what that array's
# loads miss (a little more: the copy competes for the caches).
#   step 1 (here, no GPU):  bash tools/traffic_by_array.sh build        -> variants/libmnv_shadow{0,8,16,32}.so (test-hook builds)
#   step 2 (via gpurun):    bash tools/traffic_by_array.sh run <tag> [bench args, default --workload cfg3 --laps 1]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-}" = build ]; then
  cd "$ROOT/mega-nerf-viewer_amd" && make -j8 > /dev/null || exit 1
  OUT=../variants; mkdir -p $OUT
  FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -Xclang -target-feature -Xclang -packed-fp32-ops"
  OBJS=$(ls csrc/*.o host/*.o | grep -v "csrc/mnv_accel_march.o\|csrc/mnv_comm.o\|csrc/mnv_knobs.o\|host/main.o")
  for M in 0 8 16 32; do
    /opt/rocm/bin/hipcc $FLAGS -DMNV_SHADOW_MASK=$M -c csrc/mnv_accel_march.hip -o $OUT/mnv_accel_march_shadow$M.o 2> /dev/null || exit 1
    /opt/rocm/bin/hipcc -shared -o $OUT/libmnv_shadow$M.so $OBJS $OUT/mnv_accel_march_shadow$M.o testhooks/mnv_comm.o testhooks/mnv_knobs.o -lz -lpthread -ldl || exit 1
    echo built $OUT/libmnv_shadow$M.so
  done
  exit 0
fi
shift; TAG=${1:-by_array}; shift || true
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --frame-streams 0 ${*:---workload cfg3 --laps 1}"
for M in 0 8 16 32; do
  export MNV_LIB_PATH=$ROOT/variants/libmnv_shadow$M.so MNV_SHADOW=$M MNV_BRICK_LEVELS=0
  timeout 300 rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/pmc_shadow$M" -- python3 bench.py $ARGS > "$OUT/pmc_shadow$M.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shadow$M" -- python3 bench.py $ARGS > "$OUT/trace_shadow$M.log" 2>&1
done
python3 "$ROOT/tools/traffic_by_array.py" "$OUT" | tee "$OUT/summary.txt"
