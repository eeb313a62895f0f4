#!/bin/bash
# the knobs below exist in the test-hook build of the library only (csrc/mnv_knobs.h)
export MNV_LIB_PATH=${MNV_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/mega-nerf-viewer_amd/testhooks/libmnv.so}
# window / switch thresholds of the producer / consumer kernel re-swept (they were tuned while the consumers ran at a raised priority)
cd "$(dirname "$0")/../.."
for bm in 24 32 48 64 96; do for sm in 16 32 48; do
  r=$(MNV_FUSED_BATCH_MIN=$bm MNV_F2_SWITCH_MIN=$sm timeout 300 python3 tools/guided_bench.py 32 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['producer_consumer']; print(p['ms'], p['bit_identical'])")
  echo "batch_min $bm switch_min $sm : $r"
done; done
