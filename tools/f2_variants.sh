#!/bin/bash
# guided_bench.py for the default library and every variants/libmnv_<tag>.so named on the command line (tools/build_variant.sh):
# one line per library with the producer / consumer kernel's numbers.  usage: tools/f2_variants.sh [tag ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/f2v
for v in "" "$@"; do
  if [ -n "$v" ]; then export MNV_LIB_PATH=$PWD/variants/libmnv_$v.so; else unset MNV_LIB_PATH; fi
  MNV_FUSED_DIAG=${MNV_FUSED_DIAG-1} timeout 300 python3 tools/guided_bench.py 32 4 > gpurun_out/f2v/gb_$v.json 2> gpurun_out/f2v/gb_$v.err || tail -3 gpurun_out/f2v/gb_$v.err
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/f2v/gb_{v}.json"))
    p = d["producer_consumer"]
    keep = ("ms", "bit_identical", "runs", "weight_reloads", "us_per_reload", "columns_per_run", "watchdog", "us_per_run", "us_per_run_encode_l0", "us_per_run_layers", "us_per_run_eval",
            "consumer_busy_frac", "producer_ring_wait_frac", "producer_flush_wait_frac", "producer_frac_walk_setup_step_push")
    print(v or "default", json.dumps({k: p.get(k) for k in keep}), "one_role", d["one_role"]["ms"], "four_step", d["four_step_ms"])
except Exception as e:
    print(v or "default", "FAILED", e)
PY
done
