"""Helpers shared by the tests: golden-vector loading and kernel construction."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def make_kernel(ns, spec):
    """Build a kernel from a fixture spec; ``ns`` is a GPy-like namespace
    (``oracle.gp_numpy`` or ``safeopt_amd.gpy``.kern)."""
    out = None
    for i, p in enumerate(spec):
        cls = getattr(ns, p["kind"])
        ls = p["lengthscale"]
        kw = dict(variance=p["variance"], lengthscale=ls if p["ARD"] else ls[0],
                  ARD=p["ARD"], active_dims=p["active_dims"])
        if len(spec) > 1:
            kw["name"] = "part%d" % i
        k = cls(p["input_dim"], **kw)
        out = k if out is None else out * k
    return out
