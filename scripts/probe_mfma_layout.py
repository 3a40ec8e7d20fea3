#!/usr/bin/env python
"""Derive the lane <-> element maps of the fp64 MFMA shapes empirically."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from safeopt_amd import _hip
ctx = _hip.Context.default()
for which, name in ((1, "v_mfma_f64_4x4x4_4b_f64"), (0, "v_mfma_f64_16x16x4_f64")):
    print("==", name)
    nper = 1 if which == 1 else 4
    pairs = {}
    for la in range(64):
        a = np.zeros(64); a[la] = 1.0
        b = 1000.0 + np.arange(64)
        d = ctx.probe_mfma(which, a, b)
        nz = np.flatnonzero(d)
        pairs[la] = [(int(i // nper), int(i % nper), int(round(d[i] - 1000))) for i in nz]
    for la in (0, 1, 2, 3, 4, 5, 15, 16, 17, 20, 31, 32, 48, 63):
        print("A lane %2d -> (D lane, D reg, B lane):" % la, pairs[la])
    np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out",
                         "mfma_pairs_%d.npy" % which),
            np.array([[la] + list(t) for la, v in pairs.items() for t in v]))
