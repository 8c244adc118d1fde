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
agg = collections.defaultdict(list)
dur = []
for f in ("fetch", "write", "sq1", "sq2"):
    p = os.path.join(src, f + "_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        if "k_sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(os.path.join(src, "stats_kernel_trace.csv"))):
    if "k_sweep" in r["Kernel_Name"]:
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
avg = {k: sum(v) / len(v) for k, v in agg.items()}
out = {"kernel": "k_sweep", "launches_profiled": len(dur), "avg_ms": sum(dur) / len(dur), "counters_per_launch": avg}
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    raw = (avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    corr = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    out["hbm_bytes_raw"] = raw
    out["hbm_bytes_fetch_x2"] = corr
    json.dump({"bytes_per_launch": corr, "raw_bytes_per_launch": raw, "fetch_kib": avg["FETCH_SIZE"],
               "write_kib": avg["WRITE_SIZE"], "source": f"profiles/{round_tag}_counters.json",
               "note": "FETCH_SIZE x2 (gfx950 correction for wide coalesced reads) + WRITE_SIZE, KiB -> bytes"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"{round_tag}_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
