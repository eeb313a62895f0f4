"""Copy what tools/profile_all.sh left under gpurun_out/ into profiles/ (the committed, judged copies), appending the vector-memory
path counters (tools/prof_mem.sh) and their derived figures to the batched summary."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.environ.get("ROUND", "r06")
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def avg(txt, name):
    m = re.search(r"\b%s\s+n=\s*\d+\s+avg=([0-9.e+]+)" % re.escape(name), txt)
    return float(m.group(1))


s = open(os.path.join(G, R + "_batch", "summary.txt")).read()
mem = open(os.path.join(G, R + "_mem.txt")).read()
lines = [ln for ln in mem.splitlines() if ln.startswith("pmc_")]
lat = avg(mem, "TCP_TCC_READ_REQ_LATENCY_sum") / avg(mem, "TCP_TCC_READ_REQ_sum")
cyc = avg(mem, "GRBM_GUI_ACTIVE") / 8
req = avg(mem, "TCP_TCC_READ_REQ_sum") / 256 / cyc
s += "== vector-memory path of the march kernel (tools/prof_mem.sh: bench.py --steps 3 --warmup 1 --frame-streams 0, separate --pmc passes, per-dispatch averages) ==\n"
s += "\n".join(lines) + "\n"
s += "derived: L1->L2 read latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = %.0f cycles; requests per compute unit and cycle = %.3g / 256 / %.3g = %.3f\n" % (
    lat, avg(mem, "TCP_TCC_READ_REQ_sum"), cyc, req)
s += "         -> %.0f L1 misses in flight per compute unit on average; TA busy %.0f %% (average unit); TCP_PENDING_STALL %.0f %% of the cycles of an average TCP\n" % (
    lat * req, 100 * avg(mem, "TA_BUSY_avr") / cyc, 100 * avg(mem, "TCP_PENDING_STALL_CYCLES_sum") / 256 / cyc)
open(os.path.join(P, R + "_rocprofv3_summary.txt"), "w").write(s)
shutil.copy(os.path.join(G, R + "_batch", "traffic.json"), os.path.join(P, R + "_traffic.json"))
for t in ("per_frame", "per_frame_1stream", "ref_layout", "tracker", "cfg3", "cfg4", "fog"):
    shutil.copy(os.path.join(G, "%s_%s" % (R, t), "summary.txt"), os.path.join(P, "%s_rocprofv3_summary_%s.txt" % (R, t)))
for wl in ("cfg3", "cfg4", "fog"):   # HBM bytes per launch of the 7.2 M-chunk tree's launches (bench.py: cfg3 / cfg4_n1 rooflines)
    shutil.copy(os.path.join(G, "%s_%s" % (R, wl), "traffic.json"), os.path.join(P, "%s_traffic_%s.json" % (R, wl)))
shutil.copy(os.path.join(G, R + "_guided", "summary.txt"), os.path.join(P, R + "_rocprofv3_summary_guided_fused.txt"))
d = json.load(open(os.path.join(G, R + "_bench_n1.json")))
json.dump(d, open(os.path.join(P, R + "_bench_n1.json"), "w"))
print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "per_frame", d["per_frame"]["value"])
for k in ("cfg3", "cfg4_n1", "fog"):
    print(k, d[k]["value"], {x: d[k]["roofline"][x] for x in ("frac", "algorithmic_over_peak", "frac_l2_fabric", "traffic")})
