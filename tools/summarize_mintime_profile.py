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
for f in ("pmc_counter_collection.csv", "fp64_counter_collection.csv"):
    if not os.path.exists(os.path.join(src, f)):
        continue
    for r in csv.DictReader(open(os.path.join(src, f))):
        if f.startswith("fp64") and r["Counter_Name"] == "SQ_INSTS_VALU_MFMA_F64":
            continue                     # already in the first PMC pass
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
    if "SQ_INSTS_VALU_FMA_F64" in c:
        # executed FP64 flops: 64 lanes x (ADD + MUL + TRANS + 2 FMA) wave-instructions (masked lanes count as full: an
        # upper bound) + 2048 per v_mfma_f64_16x16x4_f64 (16 x 16 x 4 multiply-adds)
        k["fp64_flops"] = 64.0 * (c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                                  c.get("SQ_INSTS_VALU_TRANS_F64", 0.0) + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]) + \
            2048.0 * c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)
        k["fp64_flops_mfma"] = 2048.0 * c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)
    out["kernels"][name] = k
# the run the counters belong to (tools/bench_mintime.py prints one JSON line): batch, nodes, iterations
for log in ("fp64.log", "pmc.log", "stats.log"):
    try:
        line = [l for l in open(os.path.join(src, log)) if l.startswith('{"metric"')][-1]
        run = json.loads(line)
        out["run"] = {k_: run[k_] for k_ in ("batch", "wall_s", "converged", "iterations_mean", "iterations_max") if k_ in run}
        break
    except Exception:
        pass
tot = sum(k.get("fp64_flops", 0.0) for k in out["kernels"].values())
if tot > 0 and "run" in out:
    out["fp64_flops_per_call"] = tot
    out["fp64_flops_mfma_per_call"] = sum(k.get("fp64_flops_mfma", 0.0) for k in out["kernels"].values())
    out["fp64_flops_per_instance"] = tot / out["run"]["batch"]
    json.dump({"source": f"profiles/{tag}_kernels.json", "batch": out["run"]["batch"], "iterations_mean": out["run"]["iterations_mean"],
               "fp64_flops_per_instance": out["fp64_flops_per_instance"],
               "fp64_flops_mfma_share": out["fp64_flops_mfma_per_call"] / tot,
               "kkt_avg_us": next((k["avg_us"] for n_, k in out["kernels"].items() if "k_mt_kkt" in n_), None),
               # the kernel set the counters belong to (> 100 calls = the iteration's kernels): bench.py prints it, so that a
               # profile of another build of the solver is visible in the line
               "iteration_kernels": sorted(n_.split("(")[0].replace("void ", "") for n_, k in out["kernels"].items() if k["calls"] > 100)},
              open(os.path.join(dst, "mintime_counters_latest.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"{tag}_kernels.json"), "w"), indent=1)
print(json.dumps({n: {a: b for a, b in k.items() if a != "counters"} for n, k in out["kernels"].items()}, indent=1))
