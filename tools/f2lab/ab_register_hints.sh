#!/bin/bash
# the two register-pressure hints of guided_fused2_kernel, each way: guided frame time (with and without diagnostics), then a stress of the best
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
for v in h00 h10 h01 h11; do
  MNV_LIB_PATH=$PWD/variants/libmnv_$v.so MNV_FUSED_DIAG=1 timeout 600 python3 tools/guided_bench.py 32 4 2>/dev/null | grep "^{" > gpurun_out/r04/hint_$v.json
  python3 -c "
import json;d=json.load(open('gpurun_out/r04/hint_$v.json'))['producer_consumer']
print('$v', d['ms'], 'diag', d.get('ms_with_diag'), 'us/run', d.get('us_per_run'), 'enc', d.get('us_per_run_encode_l0'), 'lay', d.get('us_per_run_layers'), 'eval', d.get('us_per_run_eval'), 'busy', d.get('consumer_busy_frac'), 'ringwait', d.get('producer_ring_wait_frac'), d['bit_identical'])"
done
MNV_LIB_PATH=$PWD/variants/libmnv_h11.so timeout 900 python3 tools/fused_stress.py 20 2 > gpurun_out/r04/hint_h11_stress.txt 2>&1; tail -1 gpurun_out/r04/hint_h11_stress.txt | cut -c1-300
