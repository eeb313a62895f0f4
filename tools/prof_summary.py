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
print("== kernel trace: register / LDS usage of the march kernel ==")
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        seen = set()
        for row in csv.DictReader(fh):
            n = row.get("Kernel_Name", "")
            if KEY in n and n not in seen:
                seen.add(n)
                print({k: row[k] for k in row if k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")})
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
