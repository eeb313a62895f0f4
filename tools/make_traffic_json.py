"""profiles/r02_traffic.json from a tools/prof.sh summary: HBM bytes per launch of the march kernel.
usage: make_traffic_json.py gpurun_out/<tag>/summary.txt frames_per_launch > profiles/r02_traffic.json
gfx950: FETCH_SIZE (KB) tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section), so read bytes =
2 x FETCH_SIZE; cross-checked against the exact request counters TCC_EA0_RDREQ_{128B,64B,32B} when the pass is present."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

txt = open(sys.argv[1]).read()
frames = int(sys.argv[2])
extra = sys.argv[3] if len(sys.argv) > 3 else ""


def avg(name):
    m = re.search(r"\b%s\s+n=\s*\d+\s+avg=([0-9.e+]+)" % re.escape(name), txt)
    return float(m.group(1)) if m else None


fetch, write, miss = avg("FETCH_SIZE"), avg("WRITE_SIZE"), avg("TCC_MISS_sum")
import bench  # noqa: E402  (kernel_source_sha: the profile is only valid for the kernel sources it was taken with)

out = {"kernel_source_sha": bench.kernel_source_sha(), "source": f"{sys.argv[1]} (rocprofv3 --pmc, separate passes, bench.py {extra})".replace(" )", ")"),
       "kernel": "march_accel_kernel<9,256,0,true>", "frames_per_launch": frames, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "TCC_MISS": miss,
       "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024)}
r128, r64, rall = avg("TCC_EA0_RDREQ_128B_sum"), avg("TCC_EA0_RDREQ_64B_sum"), avg("TCC_EA0_RDREQ_sum")
if r128 is not None and rall is not None:
    r32 = avg("TCC_EA0_RDREQ_32B_sum") or 0.0
    other = max(rall - r128 - (r64 or 0.0) - r32, 0.0)
    out["read_bytes_from_request_counters"] = int(r128 * 128 + (r64 or 0.0) * 64 + r32 * 32 + other * 64)
out["note"] = "gfx950: FETCH_SIZE tallies 128-B requests at 64 B, hence 2 x FETCH_SIZE; cross-check TCC_MISS x 128 B = %d" % int((miss or 0) * 128)
print(json.dumps(out, indent=1))
