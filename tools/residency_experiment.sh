#!/bin/bash
# k_sweep with the per-instance state in its three residencies: time + HBM-side traffic (FETCH_SIZE / WRITE_SIZE,
# separate rocprofv3 passes).  Run on the GPU box; output gpurun_out/residency/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/residency; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 0 1 2; do
  export RL_FORCE_RESIDENCY=$v
  python3 $R/bench.py --no-global --no-mintime --no-qss --no-cpu-baseline --steps 10 > $O/bench_$v.json 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch_$v -- python3 $R/bench.py --no-global --no-mintime --no-qss --no-cpu-baseline --steps 5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o write_$v -- python3 $R/bench.py --no-global --no-mintime --no-qss --no-cpu-baseline --steps 5 > /dev/null 2>&1
done
python3 - <<PY
import csv, json
for v in (0,1,2):
    d=json.load(open("$O/bench_%d.json"%v))
    out={"residency":v,"solves_per_s":d["value"],"kernel_ms":d["roofline"]["kernel_ms"],"lds":d["config"]["lds_bytes_per_workgroup"]}
    for c in ("fetch","write"):
        vals=[float(r["Counter_Value"]) for r in csv.DictReader(open("$O/%s_%d_counter_collection.csv"%(c,v))) if "k_sweep" in r["Kernel_Name"]]
        out[c+"_KiB"]=sum(vals)/len(vals)
    out["bytes_raw"]=(out["fetch_KiB"]+out["write_KiB"])*1024; out["bytes_fetch_x2"]=(2*out["fetch_KiB"]+out["write_KiB"])*1024
    print(json.dumps(out))
PY
