"""Counters of the fused guided-sampling kernels from the --pmc passes of tools/profile_all.sh: per-dispatch averages per kernel and the
figures derived from them (matrix-pipe busy fraction, wavefronts per SIMD).  usage: guided_pmc_summary.py gpurun_out/<tag>"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pmc_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "guided_fused" in k:
            name = "guided_fused2_kernel (producer / consumer)" if "guided_fused2" in k else "guided_fused_kernel (one role)"
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC counters of the fused guided-sampling kernels (rocprofv3 --pmc, separate passes; per-dispatch averages) ==")
for name, c in agg.items():
    a = {k: sum(v) / len(v) for k, v in c.items()}
    print(name, {k: round(v, 1) for k, v in sorted(a.items())})
    if "GRBM_GUI_ACTIVE" in a:
        cyc = a["GRBM_GUI_ACTIVE"] / 8   # per XCD
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            print(f"   matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES {a['SQ_VALU_MFMA_BUSY_CYCLES']:.3g} / (1024 SIMDs x {cyc:.3g} cycles) = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.4f}")
        if "SQ_WAVE_CYCLES" in a:
            print(f"   wavefronts per SIMD (time average): SQ_WAVE_CYCLES x 4 {4 * a['SQ_WAVE_CYCLES']:.3g} / (1024 x {cyc:.3g}) = {4 * a['SQ_WAVE_CYCLES'] / 1024 / cyc:.2f}")
    if "SQ_INSTS_MFMA" in a and "SQ_INSTS_VALU" in a:
        print(f"   VALU instructions per MFMA: {a['SQ_INSTS_VALU'] / max(1.0, a['SQ_INSTS_MFMA']):.1f}")
