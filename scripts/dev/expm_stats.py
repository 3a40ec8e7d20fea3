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

NAMES = ["waves", "  unsafe rows in them", "block tests", "  passed", "blocks that pass the pair test",
         "rows listed"]
STATES = (("config2_scale_1e6_rows", 1000,
           dict(ls=0.7, rings=5, dring=0.3, dmid=0.8, dtop=0.4, r0=2.0, dout=1.4, plateau=0.6)),
          ("grid_320x320", 320, dict(r0=2.0, rings=8, ls=0.4, dmid=0.45, plateau=0.6)))
for name, side, kw in STATES:
    gp, grid = sc.converged_state(side, 0.05, ns=gpy, **kw)
    opt = safeopt_amd.SafeOpt(gp, grid, 0.0, threshold=0.1)
    opt.optimize()
    lib = H.lib()
    out = (C.c_ulonglong * 16)()
    lib.sgp_debug_expm_stats(out, 1)
    orig = opt._backend.expander_pass
    def wrapped(*a, _o=orig):
        r = _o(*a)
        lib.sgp_debug_expm_stats(out, 1)
        print("  pass: tested %d" % r[0])
        for mode, title in ((0, "scan of the grid"), (1, "listed rows x group chunks")):
            print("   %s" % title)
            for i, nm in enumerate(NAMES[:6 - mode]):
                print("    %-50s %12d" % (nm, out[8 * mode + i]))
        return r
    opt._backend.expander_pass = wrapped
    print("%s: n = %d, %d rows" % (name, gp.X.shape[0], len(grid)))
    opt.optimize()
