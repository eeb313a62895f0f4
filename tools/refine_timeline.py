"""One refinement frame's dispatches in order, from a rocprofv3 kernel trace of tools/refine_frame_trace.py (tools/refine_trace.sh):
start (us since the frame's march began), duration, idle gap in front of it.  Says which of the serial tail is kernels and which is the
host waiting between them.      python3 tools/refine_timeline.py gpurun_out/refine_trace/trace_both [frame]
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    frame = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    f = max(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marches = [i for i, r in enumerate(rows) if "guided_fused" in r["Kernel_Name"] or "march_accel_kernel" in r["Kernel_Name"]]
    a, b = marches[frame], marches[frame + 1]
    t0 = prev = int(rows[a]["Start_Timestamp"])
    busy = 0
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::").replace("(anonymous namespace)::", "")
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {name[:110]}")
        if r is not rows[b]:
            busy += e - s
        prev = max(prev, e)
    print(f"frame {frame}: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from march to march, {busy / 1e3:.1f} us of it inside kernels, {b - a} dispatches")


if __name__ == "__main__":
    main()
