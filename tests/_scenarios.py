"""Problem states for the tests of the expander loop (tests/test_gpu_expander_passes.py,
bench.py's `no_expander_state`, scripts/dev/no_expander.py).  NumPy only; `ns` is the GP
namespace (oracle.gp_numpy or safeopt_amd.gpy -- the same constructor surface)."""
import numpy as np


def _grid(bounds, num):
    # (safeopt_amd.linearly_spaced_combinations without importing the package here)
    axes = [np.linspace(b[0], b[1], n) for b, n in zip(bounds, num)]
    return np.array([x.ravel() for x in np.meshgrid(*axes)]).T


def truth(r, R, plateau=0.45):
    return plateau + (1 - plateau) * np.exp(-(r / 0.8) ** 2) - (np.minimum(r, R + 1.0) / R) ** 8


def make_gp(ns, data):
    """The GP of a state in namespace `ns` (oracle.gp_numpy / safeopt_amd.gpy)."""
    k = getattr(ns, "kern", ns).RBF(2, 2.0, [data["ls"]] * 2, ARD=True)
    return ns.GPRegression(data["X"], data["Y"], k, noise_var=data["noise"])


def rim_data(side=1000, seed=3, R=3.0, noise=0.01 ** 2, ls=0.5, rings=6, dring=0.2,
             dout=0.7, dmid=0.55, dtop=0.25, r0=None, plateau=0.45):
    """Truth: a narrow peak (1.0) on a plateau (0.45) that falls through fmin = 0 at r = R.  The peak
    and the rim are observed densely (narrow intervals: the maximisers, the rows next to the
    unsafe set), the plateau on a coarse lattice (wide intervals below the best lower bound:
    candidates by the thousand), the outside coarsely."""
    rng = np.random.default_rng(seed)
    pts = []
    r0 = R - dring * (rings // 2) if r0 is None else r0
    for k in range(rings):
        r = r0 + dring * k
        m = int(2 * np.pi * r / (0.8 * ls))
        th = 2 * np.pi * (np.arange(m) + 0.5 * (k & 1)) / m
        pts.append(np.c_[r * np.cos(th), r * np.sin(th)])
    def lattice(step):
        g = np.arange(-5, 5.001, step)
        return np.array([(a, b) for a in g for b in g])
    Xo = lattice(dout); pts.append(Xo[np.hypot(Xo[:, 0], Xo[:, 1]) > r0 + dring * rings + 0.1])
    Xm = lattice(dmid) + 0.07; rm = np.hypot(Xm[:, 0], Xm[:, 1]); pts.append(Xm[(rm > 0.9) & (rm < r0 - 0.15)])
    Xt = lattice(dtop) + 0.03; pts.append(Xt[np.hypot(Xt[:, 0], Xt[:, 1]) <= 0.9])
    X = np.vstack(pts)
    r = np.hypot(X[:, 0], X[:, 1])
    Y = (truth(r, R, plateau) + np.sqrt(noise) * rng.normal(size=len(X)))[:, None]
    return dict(X=X, Y=Y, ls=float(ls), noise=float(noise), grid=_grid([(-5., 5.)] * 2, [side, side]))


def rim_state(side=1000, ns=None, **kw):
    data = rim_data(side, **kw)
    return make_gp(ns, data), data["grid"]


def converged_rows(gp, grid, margin=0.05, beta=2.0):
    """Rows to KEEP so that no unsafe row is left within `margin` below fmin = 0."""
    mu, var = gp.predict_noiseless(grid)
    l = mu[:, 0] - beta * np.sqrt(var[:, 0])
    return ~((l > -margin) & (l <= 0.0))


def converged_state(side=320, margin=0.05, ns=None, **kw):
    """rim_state with the rows whose lower bound sits within `margin` below fmin = 0 taken out
    of the parameter set: the unsafe rows that are left cannot be lifted across fmin by any
    candidate -- a converged run, the reference visits every candidate and marks none."""
    gp, grid = rim_state(side, ns=ns, **kw)
    return gp, np.ascontiguousarray(grid[converged_rows(gp, grid, margin)])
