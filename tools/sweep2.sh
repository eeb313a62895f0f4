#!/bin/bash
run() { echo -n "$*: "; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']/16, d['roofline']['avg_launch_ms'])"; }
run MNV_REFILL_TAIL=56 MNV_GRAB_TAIL=64
for RT in 8 16 32; do for GT in 8 16 32; do run MNV_REFILL_TAIL=$RT MNV_GRAB_TAIL=$GT; done; done
EXTRA="--streams 2" run MNV_REFILL_TAIL=56 MNV_GRAB_TAIL=64
EXTRA="--streams 3" run MNV_REFILL_TAIL=56 MNV_GRAB_TAIL=64
EXTRA="--streams 2" run MNV_REFILL_TAIL=16 MNV_GRAB_TAIL=16
