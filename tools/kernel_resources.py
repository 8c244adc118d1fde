#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a gfx950 assembly listing (hipcc -save-temps ... *.s).

usage: tools/kernel_resources.py FILE.s [substring]
"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    blk = ".agpr_count:" + blk
    f = dict(re.findall(r"\.(\w+):\s+([^\n]+)", blk))
    name = f.get("name", "?")
    if flt not in name:
        continue
    print(f"{name[:70]:70s} vgpr {f.get('vgpr_count'):>4s} agpr {f.get('agpr_count'):>3s} sgpr {f.get('sgpr_count'):>3s} "
          f"spill {f.get('vgpr_spill_count'):>4s} scratch {f.get('private_segment_fixed_size'):>5s} lds {f.get('group_segment_fixed_size'):>6s}")
