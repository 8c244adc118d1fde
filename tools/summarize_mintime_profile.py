#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/profile_mintime.sh) -> profiles/<tag>_kernel_stats.csv (rocprofv3 --stats, as is) and
profiles/<tag>_kernels.json (per kernel: calls, average duration, share, summed counters of the PMC pass)."""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "stats_kernel_stats.csv")))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(os.path.join(src, "pmc_counter_collection.csv"))):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
out = {"command": f"rocprofv3 --kernel-trace --stats / --pmc ... -- python3 tools/bench_mintime.py (tag {tag})",
       "units": "SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAIT_ANY in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles",
       "kernels": {}}
for name, r in stats.items():
    if "k_mt_" not in name and "k_qss" not in name:
        continue
    c = dict(agg.get(name, {}))
    k = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "share_pct": float(r["Percentage"]), "counters": c}
    if c.get("SQ_WAVE_CYCLES"):
        k["valu_active_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
        k["waiting_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            k["mfma_busy_of_wave_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"])
            k["cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_VALU_MFMA_F64"]
    out["kernels"][name] = k
json.dump(out, open(os.path.join(dst, f"{tag}_kernels.json"), "w"), indent=1)
print(json.dumps({n: {a: b for a, b in k.items() if a != "counters"} for n, k in out["kernels"].items()}, indent=1))
