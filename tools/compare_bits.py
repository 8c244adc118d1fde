#!/usr/bin/env python3
"""Two outputs of tools/full_batch_bits.py gpu (two builds of the library on one box): are control points and success counts
the same bit patterns on every instance?  usage: tools/compare_bits.py A.npz B.npz"""
import json
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
assert float(a["widths_digest"]) == float(b["widths_digest"])
B = a["ctrl"].shape[0]
same = np.array([np.array_equal(a["ctrl"][i].view(np.int64), b["ctrl"][i].view(np.int64)) and np.array_equal(a["ns"][i], b["ns"][i])
                 for i in range(B)])
print(json.dumps({"instances": int(B), "bit_identical": int(same.sum()),
                  "max_control_point_deviation_m": float(np.abs(a["ctrl"] - b["ctrl"]).max()),
                  "kernel_ms": [float(a["kernel_ms"]), float(b["kernel_ms"])]}))
sys.exit(0 if same.all() else 1)
