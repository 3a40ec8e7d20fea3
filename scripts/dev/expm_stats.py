"""Counters of k_expander_many (a library built with -DEXPM_STATS: scripts/dev/build_variant.sh
stats "-DEXPM_STATS"; SAFEOPT_HIP_LIB=scripts/dev/ab/stats.so) on the converged states of
bench.py: how many blocks each test lets through and how full the contracted blocks are.

    SAFEOPT_HIP_LIB=scripts/dev/ab/stats.so python scripts/dev/expm_stats.py
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import safeopt_amd, safeopt_amd.gpy as gpy, safeopt_amd._hip as H
import _scenarios as sc

NAMES = ["waves (items)", "  unsafe rows in them", "block tests", "  passed", "pair tests (four groups each)"]
STATES = (("config2_scale_1e6_rows", 1000,
           dict(ls=0.7, rings=5, dring=0.3, dmid=0.8, dtop=0.4, r0=2.0, dout=1.4, plateau=0.6)),
          ("grid_320x320", 320, dict(r0=2.0, rings=8, ls=0.4, dmid=0.45, plateau=0.6)))
for name, side, kw in STATES:
    gp, grid = sc.converged_state(side, 0.05, ns=gpy, **kw)
    opt = safeopt_amd.SafeOpt(gp, grid, 0.0, threshold=0.1)
    opt.optimize()
    lib = H.lib()
    out = (C.c_ulonglong * 32)()
    lib.sgp_debug_expm_stats(out, 1)
    orig = opt._backend.expander_pass
    def wrapped(*a, _o=orig):
        r = _o(*a)
        lib.sgp_debug_expm_stats(out, 1)
        print("  pass: tested %d" % r[0])
        for off, title in ((0, "MODE 0: scan of the grid, out at the first possible pair"),
                           (16, "MODE 2: listed waves x chunks of 32 groups")):
            print("   %s" % title)
            for i, nm in enumerate(NAMES):
                print("    %-50s %12d" % (nm, out[off + i]))
        print("   MODE 1: items %d, blocks contracted %d" % (out[8], out[9]))
        return r
    opt._backend.expander_pass = wrapped
    print("%s: n = %d, %d rows" % (name, gp.X.shape[0], len(grid)))
    opt.optimize()
