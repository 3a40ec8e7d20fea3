#!/usr/bin/env python
"""fp64 issue-rate probes on the MI355X (prints a small table).

    python scripts/microbench.py

Establishes the practical ceilings DESIGN.md quotes next to the vendor peak:
what one CU sustains for v_mfma_f64_16x16x4_f64 alone, for v_fma_f64 alone, and
for both interleaved, at 1 / 2 / 4 / max waves per SIMD.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safeopt_amd import _hip  # noqa: E402

NAMES = {0: "MFMA x8 chains", 1: "MFMA x4 chains", 5: "MFMA x16 chains",
         3: "v_fma_f64 only", 2: "MFMA + 8 v_fma/MFMA", 4: "MFMA + 2 v_fma/MFMA",
         6: "MFMA 4x4x4_4b x16 chains", 7: "MFMA 4x4x4_4b x4 chains",
         8: "MFMA 4x4x4_4b x2 chains", 9: "MFMA 4x4x4_4b distinct ops"}


def main():
    ctx = _hip.Context.default()
    print("%-22s %10s %12s %12s %12s" % ("mode", "waves/SIMD", "MFMA TF/s", "VALU TF/s", "sum"))
    for lds, occ in ((0, "regs"), (40 * 1024, "4"), (80 * 1024, "2"), (160 * 1024 - 64, "1")):
        for mode in (0, 1, 5, 6, 7, 8, 9, 3, 2, 4):
            it = 4000 if mode in (2, 3) else 10000
            m, v = ctx.microbench(mode, it, lds)
            print("%-22s %10s %12.1f %12.1f %12.1f" % (NAMES[mode], occ, m, v, m + v))


if __name__ == "__main__":
    main()
