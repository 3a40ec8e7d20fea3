#!/usr/bin/env python
"""One stage of the sweep's inner loop in isolation (k_stage_bench in
safeopt_amd/csrc/sweep.hip): what each ingredient costs next to the MFMAs.

    python scripts/stagebench.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safeopt_amd import _hip  # noqa: E402

NAMES = {20: "MFMAs + swizzles + operand reads", 21: "+ stage barrier",
         22: "+ LDS-DMA of next chunk + barrier", 23: "+ covariance evaluation",
         24: "evaluation + MFMAs only"}


def main():
    ctx = _hip.Context.default()
    print("%-36s %6s %12s %12s %10s" % ("stage contents", "slots", "MFMA TF/s", "ns/stage", "ideal ns"))
    for lo in (0, 9):
        for mode in (20, 21, 22, 23, 24):
            tf, ns = ctx.microbench(mode, 4000, lo)
            ideal = (16 - lo) * 16 * 512.0 * 2048 / 76.5e12 * 1e9
            print("%-36s %6d %12.1f %12.0f %10.0f" % (NAMES[mode], 16 - lo, tf, ns, ideal))


if __name__ == "__main__":
    main()
