#!/bin/bash
# consumers that share a group of rings by slot residue (MNV_F2_SHARE = 1 / 2 / 4): guided frame time with phases, then correctness of each
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
for v in s1 s2 s4; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so MNV_FUSED_DIAG=1 timeout 600 python3 tools/guided_bench.py 32 4 2>/dev/null | grep "^{" > gpurun_out/r04/share_$v.json
  python3 -c "
import json;d=json.load(open('gpurun_out/r04/share_$v.json'))['producer_consumer']
print('$v', d['ms'], 'diag', d.get('ms_with_diag'), 'runs', d.get('runs'), 'cols/run', d.get('columns_per_run'), 'us/run', d.get('us_per_run'), 'busy', d.get('consumer_busy_frac'), 'ringwait', d.get('producer_ring_wait_frac'), 'flushwait', d.get('producer_flush_wait_frac'), 'reloads', d.get('weight_reloads'), d['bit_identical'], 'watchdog', d.get('watchdog'))"
done
for v in s2 s4; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so timeout 900 python3 tools/fused_stress.py 10 2 > gpurun_out/r04/share_${v}_stress.txt 2>&1; tail -1 gpurun_out/r04/share_${v}_stress.txt | cut -c1-260
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so timeout 900 python3 -m pytest tests/test_guided_fused_gpu.py -x -q -m gpu -k "not many_frames" > gpurun_out/r04/share_${v}_pytest.txt 2>&1; tail -2 gpurun_out/r04/share_${v}_pytest.txt
done
