"""A/B on one box: the arg-max of the step taken behind the result of the last pass (one round trip
less) against a separate sgp_grid_argmax -- median ms per optimize() in the converged states."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import safeopt_amd, safeopt_amd.gpy as gpy
import _scenarios as sc
STATES = ((1000, dict(ls=0.7, rings=5, dring=0.3, dmid=0.8, dtop=0.4, r0=2.0, dout=1.4, plateau=0.6)),
          (320, dict(r0=2.0, rings=8, ls=0.4, dmid=0.45, plateau=0.6)))
for side, kw in STATES:
    gp, grid = sc.converged_state(side, 0.05, ns=gpy, **kw)
    res = {}
    for spec in (True, False, True, False):
        opt = safeopt_amd.SafeOpt(gp, grid, 0.0, threshold=0.1)
        if not spec:
            orig = opt._backend.expander_pass
            opt._backend.expander_pass = lambda *a, _o=orig: _o(*a[:8])        # (no scaling: no arg-max)
        ctx = opt._backend.ctx
        opt.optimize(); opt.optimize()
        ts = []
        for _ in range(30):
            ctx.sync(); t0 = time.perf_counter(); opt.optimize(); ctx.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        res.setdefault(spec, []).append(float(np.median(ts)))
    print("side %d: arg-max with the pass %s ms, separate %s ms" % (
        side, ["%.3f" % v for v in res[True]], ["%.3f" % v for v in res[False]]), flush=True)
