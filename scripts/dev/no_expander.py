"""A converged-run state of the expander loop on a big grid: many candidates, NO expander.

The safe region is a disk whose rim is densely observed (values falling steeply through fmin
there), its inside sparsely: the wide rows inside are candidates (wider than every maximiser),
but an optimistic observation there moves no row outside the rim across fmin.  The reference
visits EVERY candidate in that state (gp_opt.py:557-612).

    python scripts/dev/no_expander.py [side] [n_rim] [n_in]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import safeopt_amd, safeopt_amd.gpy as gpy


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from _scenarios import rim_state as _rim, converged_state as _conv


def rim_state(side, **kw):
    return _rim(side, ns=gpy, **kw)


def converged_state(side, margin, **kw):
    return _conv(side, margin, ns=gpy, **kw)


if __name__ == "__main__":
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    kw = dict(a.split("=") for a in sys.argv[2:])
    kw = {k: (float(v) if "." in v else int(v)) for k, v in kw.items()}
    margin = kw.pop("margin", None)
    gp, grid = rim_state(side, **kw) if margin is None else converged_state(side, margin, **kw)
    for big in ((True,) if os.environ.get("ONLY_BIG") else (True, False)):
        lip = float(os.environ["LIP"]) if os.environ.get("LIP") else None      # Lipschitz certificates
        opt = safeopt_amd.SafeOpt(gp, grid, 0.0, lipschitz=lip, threshold=0.1)
        opt.big_passes = big
        if os.environ.get("PASS_SIZES"):
            opt.pass_sizes = tuple(int(v) for v in os.environ["PASS_SIZES"].split(","))
        ctx = opt._backend.ctx
        passes = []
        if big:
            attr = "lipschitz_pass" if lip else "expander_pass"
            orig = getattr(opt._backend, attr)
            setattr(opt._backend, attr, lambda *a, _o=orig: (lambda r: (passes.append(r[:2]), r)[1])(_o(*a)))
        for rep in range(int(os.environ.get("REPS", "2"))):
            ctx.sync(); t0 = time.perf_counter()
            x = opt.optimize()
            ctx.sync(); dt = time.perf_counter() - t0
        S, M, Gm = np.array(opt.S), np.array(opt.M), np.array(opt.G)
        Q = np.array(opt.Q)
        w = (Q[:, 1] - Q[:, 0]) / opt.scaling[0]
        max_var = w[M].max() if M.any() else np.inf
        cand = S & ~M & (w > max_var) & (Q[:, 1] - Q[:, 0] > 0.1 * 2.0)
        if big:
            print("    passes (tested, hits):", passes[len(passes) // 2:])
        print("big %d side %d %s n %d rows %d: |S| %d |M| %d cand %d unsafe %d |G| %d %s optimize %.3f ms  x %s" % (
            big, side, kw, gp.X.shape[0], len(grid), S.sum(), M.sum(), cand.sum(), (~S).sum(), Gm.sum(),
            np.flatnonzero(Gm)[:3], dt * 1e3, x), flush=True)
        if os.environ.get("FULL"):
            if big:
                opt.compute_sets(full_sets=True)            # (first call: the scratch buffers grow)
            ctx.sync(); t0 = time.perf_counter()
            opt.compute_sets(full_sets=True)
            ctx.sync(); dt = time.perf_counter() - t0
            Gf = np.array(opt.G)
            print("    full_sets: |G| %d  %.3f ms  (hash %d)" % (Gf.sum(), dt * 1e3, hash(Gf.tobytes()) & 0xffff))
