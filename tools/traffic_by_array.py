"""Summary of tools/traffic_by_array.sh: per variant the march kernel's L2 misses / hits / fabric read requests per launch (average over
its dispatches) and its duration; then the split of the baseline's misses by array from the shadow-load differences."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*/"))):
    name = os.path.basename(d.rstrip("/"))[4:]
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "march" in row.get("Kernel_Name", ""):
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    r = {k: sum(v) / len(v) for k, v in agg.items()}
    durs = []
    for f in glob.glob(os.path.join(out, "trace_" + name, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rows = [x for x in csv.DictReader(fh) if "march" in x.get("Kernel_Name", "")]
        rows.sort(key=lambda x: int(x["Start_Timestamp"]))
        durs = [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows]
    r["launch_ms_warm_mean"] = round(sum(durs[1:]) / max(1, len(durs) - 1) / 1e6, 4) if len(durs) > 1 else None
    res[name] = r
base = res.get("shadow0", {})
if "TCC_MISS_sum" in base:
    b = base["TCC_MISS_sum"]
    split = {arr: res[f"shadow{m}"]["TCC_MISS_sum"] - b for m, arr in ((8, "grid2"), (16, "nodes"), (32, "rows")) if f"shadow{m}" in res and "TCC_MISS_sum" in res[f"shadow{m}"]}
    res["split"] = {"baseline_misses_per_launch": b, "baseline_GB_per_launch": round(b * 128 / 1e9, 2), "shadow_delta_misses": split,
                    "shadow_delta_GB": {k: round(v * 128 / 1e9, 2) for k, v in split.items()},
                    "share_of_baseline": {k: round(v / b, 3) for k, v in split.items()},
                    "sum_of_deltas_over_baseline": round(sum(split.values()) / b, 3) if b else None,
                    "note": "a shadow copy also competes for the caches, so the deltas over-count a little; what is left (outputs, launch slots, scratch) is small"}
print(json.dumps(res, indent=1))
