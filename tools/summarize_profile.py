#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (tools/profile_bench.sh) into the committed evidence under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --stats summary
  profiles/<tag>_counters.json         per-launch PMC averages of the sweep and global-QP kernels
  profiles/counters_latest.json        the same, where bench.py reads roofline.traffic / valu from
                                       (source + "measured_in_run": false are carried into the bench line)
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced streaming reads
by 2x.  The kernels' global reads are 8- and 16-byte per lane mixed, so both the raw and the
2x figure are recorded and the LARGER one is what bench.py reports as `traffic` (upper bound).
clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same pass (the guide's DVFS recipe)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
round_tag = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats_kernel_stats.csv"), os.path.join(dst, f"{round_tag}_kernel_stats.csv"))
for line in open(os.path.join(src, "stats.log")):
    if line.startswith('{"metric"'):
        open(os.path.join(dst, f"{round_tag}_bench_under_rocprof.json"), "w").write(line)


def match(name, pattern):
    """pattern = "substring" or "substring|end of the template argument list": k_sweep's instantiations differ in their
    last two template arguments (STRICT: the reference-order arithmetic; RAISE: its numpy-error-state variant, which
    returns at once for instances that do not need it)."""
    sub, _, last = pattern.partition("|")
    if sub not in name:
        return False
    return not last or name.split(">(")[0].rstrip().endswith(last)


def durations(fname, pattern):
    p = os.path.join(src, fname)
    if not os.path.exists(p):
        return []
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            for r in csv.DictReader(open(p)) if match(r["Kernel_Name"], pattern)]


def per_kernel(pattern):
    agg = collections.defaultdict(list)
    for f in ("fetch", "write", "sq1", "sq2", "sq3"):
        p = os.path.join(src, f + "_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if match(r["Kernel_Name"], pattern):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = durations("stats_kernel_trace.csv", pattern)
    if not dur:
        return None
    avg = {k: sum(v) / len(v) for k, v in agg.items()}
    srt = sorted(dur)
    # min and median next to the average: a cold first launch (and the profiler's own overhead on it) otherwise puts the
    # profile's figure above the driver's ms_per_step
    out = {"kernel": pattern, "launches_profiled": len(dur), "avg_ms": sum(dur) / len(dur), "min_ms": srt[0],
           "median_ms": srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2]),
           "counters_per_launch": avg}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        out["hbm_bytes_raw"] = (avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
        out["hbm_bytes_fetch_x2"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    d3 = durations("sq3_kernel_trace.csv", pattern)
    if "GRBM_GUI_ACTIVE" in avg and d3:
        out["clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / (sum(d3) / len(d3) * 1e-3) / 1e9
        out["avg_ms_in_counter_pass"] = sum(d3) / len(d3)
    return out


kernels = {}
# k_sweep<K, BLOCK, RINGS_LDS, JOINT, DUMP, SIGMA_LDS, STRICT, RAISE, LITE>: told apart by the last three arguments
for name, pat in (("k_sweep", "k_sweep<|false, false, false"), ("k_sweep_reference_order", "k_sweep<|true, false, false"),
                  ("k_sweep_reference_order_raise", "k_sweep<|true, true, false"), ("k_sweep_branch", "k_sweep<|true, false, true"),
                  ("k_global_qp", "k_global_qp"), ("k_global_xy", "k_global_xy<")):
    k = per_kernel(pat)
    if k:
        kernels[name] = k
doc = {"source": f"profiles/{round_tag}_counters.json (rocprofv3 --pmc passes of tools/profile_bench.sh, bench.py --steps 5)",
       "kernels": kernels}
json.dump(doc, open(os.path.join(dst, f"{round_tag}_counters.json"), "w"), indent=1)
json.dump(doc, open(os.path.join(dst, "counters_latest.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
