#!/bin/bash
# A/B of -D experiment switches of the march kernel (rebuilds on the GPU box); every run checks 2 poses against the oracle
run() { echo -n "[$1]: "; touch mega-nerf-viewer_amd/csrc/mnv_march_accel.hip; make -C mega-nerf-viewer_amd -j8 EXTRA="$1" > /tmp/mk.log 2>&1 || { echo build failed; tail -3 /tmp/mk.log; return; }; python3 bench.py --steps 6 --warmup 2 --cpu-poses 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], 'bad pixels', d['parity']['pixels_not_bit_identical'])"; }
for defs in "$@"; do run "$defs"; done
