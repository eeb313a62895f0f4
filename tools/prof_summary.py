"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KEY = "march"


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== register / LDS / scratch usage of the march kernels: code-object metadata of libmnv.so (tools/kernel_resources.py; rocprofv3's VGPR_Count / LDS_Block_Size columns are wrong for these kernels on this stack) ==")
try:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kernel_resources

    for k in kernel_resources.kernels():
        if any(t in k["kernel"] for t in ("march_accel_kernel<9, 256, 0>", "march_accel_kernel<9, 256, 2>", "march_accel_kernel<9, 256, 3>", "march_ref_layout_kernel<9>")):
            print(k)
    print("dynamic LDS of march_accel_kernel<9,256,*>: 256 + (9 + 2) * 1024 + 2048 = 13568 bytes per workgroup (launch_accel)")
except Exception as e:  # noqa: BLE001
    print("unavailable:", e)
print("== kernel trace: duration of every dispatch of the march kernel, in order (ns; the first one runs cold: page tables, caches) ==")
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        rows = [r for r in csv.DictReader(fh) if KEY in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    print(durs, "mean of all but the first:", round(sum(durs[1:]) / max(1, len(durs) - 1)) if len(durs) > 1 else None)
print("== PMC counters, per-dispatch average over the march kernel's dispatches ==")
for d in find("pmc_*/"):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if KEY in row.get("Kernel_Name", ""):
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{os.path.basename(d.rstrip('/')):10s} {k:32s} n={len(v):4d} avg={sum(v) / len(v):.6g}")
