#!/bin/bash
# L2 / fabric traffic of the TRACKER frame (the reference's every-frame call) with the inline cell words + brick records and -- MNV_BRICK_LEVELS=0 on the
# test-hook build -- without them (the node-word walk of rounds 1-5): separate --pmc passes, never with a trace.  usage (via gpurun): bash tools/prof_tracker_traffic.sh <tag>
set -u
TAG=${1:-r06_track}
export TMPDIR=/tmp
export MNV_LIB_PATH=$PWD/mega-nerf-viewer_amd/testhooks/libmnv.so
for v in words walk; do
  OUT=$PWD/gpurun_out/${TAG}_$v; mkdir -p "$OUT"
  if [ $v = walk ]; then export MNV_BRICK_LEVELS=0; else unset MNV_BRICK_LEVELS; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/tracker_frames.py 2 > "$OUT/trace.log" 2>&1
  pmc() { local name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 tools/tracker_frames.py 2 > "$OUT/pmc_$name.log" 2>&1; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
  pmc sq SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES
  python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
  grep "ms_per_tracker_frame" "$OUT/trace.log" >> "$OUT/summary.txt"
done
unset MNV_BRICK_LEVELS
python3 - <<PY
import json, re, sys
sys.path.insert(0, ".")
import bench
out = {"kernel_source_sha": bench.kernel_source_sha(), "what": "one 1920x1080 tracker frame of the cfg2 tree (render_voxels with both tracker tensors, packed layout), per-dispatch averages of separate rocprofv3 --pmc passes; words = inline cell words + brick records (round 6), walk = MNV_BRICK_LEVELS=0 on the test-hook build: the tracker kernel walks the node words as in rounds 1-5"}
for v in ("words", "walk"):
    txt = open("gpurun_out/${TAG}_%s/summary.txt" % v).read()
    def avg(name):
        m = re.search(r"\b%s\s+n=\s*\d+\s+avg=([0-9.e+]+)" % re.escape(name), txt)
        return float(m.group(1)) if m else None
    ms = re.search(r"'ms_per_tracker_frame': ([0-9.]+)", txt)
    d = {k: avg(k) for k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_MISS_sum", "TCC_HIT_sum", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU")}
    d["l2_fabric_bytes_per_frame"] = int(2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024) if d["FETCH_SIZE"] and d["WRITE_SIZE"] else None
    d["ms_per_frame_hip_events"] = float(ms.group(1)) if ms else None
    out[v] = d
json.dump(out, open("gpurun_out/${TAG}_traffic_track.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
