#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (tools/profile_bench.sh) into the committed evidence under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --stats summary
  profiles/<tag>_counters.json         per-launch PMC averages of the sweep kernel
  profiles/hbm_traffic.json            bytes per launch for bench.py's roofline.traffic
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced streaming reads
by 2x.  The sweep kernel's global reads are 8- and 16-byte per lane mixed, so both the raw and the
2x figure are recorded and the LARGER one is used as `bytes_per_launch` (upper bound)."""
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
def per_kernel(pattern):
    agg = collections.defaultdict(list)
    dur = []
    for f in ("fetch", "write", "sq1", "sq2"):
        p = os.path.join(src, f + "_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if pattern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(os.path.join(src, "stats_kernel_trace.csv"))):
        if pattern in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    if not dur:
        return None
    avg = {k: sum(v) / len(v) for k, v in agg.items()}
    out = {"kernel": pattern, "launches_profiled": len(dur), "avg_ms": sum(dur) / len(dur), "counters_per_launch": avg}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        out["hbm_bytes_raw"] = (avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
        out["hbm_bytes_fetch_x2"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    return out


out = per_kernel("k_sweep")
avg = out["counters_per_launch"]
if "hbm_bytes_raw" in out:
    json.dump({"bytes_per_launch": out["hbm_bytes_fetch_x2"], "raw_bytes_per_launch": out["hbm_bytes_raw"],
               "fetch_kib": avg["FETCH_SIZE"], "write_kib": avg["WRITE_SIZE"],
               "source": f"profiles/{round_tag}_counters.json",
               "note": "FETCH_SIZE x2 (gfx950 correction for wide coalesced reads) + WRITE_SIZE, KiB -> bytes"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
both = {"k_sweep": out}
g = per_kernel("k_global_qp")
if g:
    both["k_global_qp"] = g
json.dump(both if g else out, open(os.path.join(dst, f"{round_tag}_counters.json"), "w"), indent=1)
print(json.dumps(both, indent=1))
