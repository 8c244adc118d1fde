#!/usr/bin/env python3
"""Where the wave of k_mt_kkt spends its cycles (diagnostic build -DMT_STAMPS, one instance):
   RL_LIB_PATH=<stamped library> python tools/stamp_mintime.py"""
import ctypes, json, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench_mintime.py", "1"]
runpy.run_path(os.path.join(ROOT, "tools", "bench_mintime.py"), run_name="__main__")
raw = ctypes.CDLL(os.environ["RL_LIB_PATH"])
out = (ctypes.c_ulonglong * 8)()
assert raw.rl_debug_mt_stamps(out) == 0
c = list(out)
names = ["fetch of the node's blocks", "inverse of the 16x16 pivot block", "a_j, P = E S^-1, Q = F S^-1",
         "store P, Q; r_last, S_last updates", "next node: S, F, r updates", "elimination total (with retries)",
         "x_last and a' = a - Q' x_last", "back substitution chain"]
tot = c[5] + c[6] + c[7]
print(json.dumps({n: {"cycles": v, "share": v / tot} for n, v in zip(names, c)}, indent=1))
