"""NumPy stand-in for the rank-local device backend of ``SafeOpt``.

TEST INFRASTRUCTURE: lets the CPU suite drive the product's sharded host logic
(``SafeOpt.compute_sets`` phase driver, shard ranges, top-k / arg-max merges,
communicator calls) with world_size > 1 over gloo, with the oracle's GP
arithmetic standing in for the HIP kernels.  It mirrors the phase interface of
``safeopt_amd.gp_opt._HipGridBackend`` one to one.
"""
import numpy as np

from oracle import safeopt_numpy as son

Q_, S_, M_, G_ = 0, 1, 2, 3


class OracleGridBackend(object):
    def __init__(self, gps, inputs_shard, global_offset):
        self.gps = gps
        self.x = np.array(inputs_shard, dtype=float)
        self.lo = int(global_offset)
        self.hi = self.lo + self.x.shape[0]
        N, G = self.x.shape[0], len(gps)
        self.Q = np.zeros((N, 2 * G))
        self.mean = np.zeros((N, G))
        self.var = np.zeros((N, G))
        self.S = np.zeros(N, bool); self.M = self.S.copy(); self.G = self.S.copy()
        self.cand = self.S.copy(); self.w = np.zeros(N)

    def owns(self, gidx):
        return self.lo <= gidx < self.hi

    def set_context(self, c):
        c = np.atleast_1d(c)
        self.x[:, -c.size:] = c

    def _safe(self, fmin):
        self.S = son.safe_set(self.Q, fmin)
        l0 = self.Q[:, 0]
        return (l0[self.S].max() if self.S.any() else -np.inf), bool(self.S.any())

    def confidence(self, beta, fmin):
        for i, gp in enumerate(self.gps):
            m, v = gp.predict_noiseless(self.x)
            self.mean[:, i], self.var[:, i] = m.ravel(), v.ravel()
            sd = np.sqrt(v.ravel())
            self.Q[:, 2 * i] = m.ravel() - beta * sd
            self.Q[:, 2 * i + 1] = m.ravel() + beta * sd
        return self._safe(fmin)

    def upload_Q(self, Q, fmin):
        self.Q[:] = Q
        return self._safe(fmin)

    def upload_mask(self, what, mask):
        {S_: self.S, M_: self.M, G_: self.G}[what][:] = np.asarray(mask, dtype=bool)

    def maximizers(self, max_l):
        l0, u0 = self.Q[:, 0], self.Q[:, 1]
        self.M = self.S & (u0 >= max_l)
        return (u0[self.M] - l0[self.M]).max() if self.M.any() else -np.inf

    def candidates(self, max_var, scaling, thr_beta, full_sets):
        wd = self.Q[:, 1::2] - self.Q[:, ::2]
        if full_sets:
            self.cand = self.S.copy()
        else:
            self.cand = (self.S & ~self.M & (np.max(wd / np.asarray(scaling), axis=1) > max_var)
                         & np.any(wd > np.asarray(thr_beta), axis=1))
        self.w = wd.max(axis=1)
        self.G[:] = False
        return int(self.cand.sum()), int((~self.S).sum())

    def topk(self, mode, cut_w, cut_idx, k):
        idx = np.flatnonzero(self.cand) + self.lo
        w = self.w[idx - self.lo]
        if mode == 1:
            keep = idx > cut_idx
            order = np.argsort(idx[keep], kind="stable")[:k]
            return -idx[keep][order].astype(float), idx[keep][order]
        keep = (w < cut_w) | ((w == cut_w) & (idx < cut_idx))
        w, idx = w[keep], idx[keep]
        order = np.lexsort((-idx, -w))[:k]
        return w[order], idx[order]

    def gather_rows(self, gidx):
        li = np.asarray(gidx) - self.lo
        return self.x[li], self.mean[li], self.var[li], self.Q[li]

    def expander_check(self, beta, fmin, xc, mu_c, u_c, near_frac=0.0):
        m, G = xc.shape[0], len(self.gps)
        flags = np.zeros((m, G), dtype=np.int32)
        unsafe = ~self.S
        for c in range(m):
            for i, gp in enumerate(self.gps):
                if fmin[i] == -np.inf or not unsafe.any():
                    continue
                gp.set_XY(np.vstack([gp.X, xc[[c]]]), np.vstack([gp.Y, [[u_c[c, i]]]]))
                rows = self.x[unsafe]
                if near_frac > 0:        # the product's cheap first probe
                    kx = gp.kern.K(rows, xc[[c]]).ravel()
                    rows = rows[kx >= near_frac * gp.kern.Kdiag(xc[[c]])[0]]
                hit = False
                if rows.shape[0]:
                    m2, v2 = gp.predict_noiseless(rows)
                    hit = np.any(m2.ravel() - beta * np.sqrt(v2.ravel()) >= fmin[i])
                gp.set_XY(gp.X[:-1], gp.Y[:-1])
                flags[c, i] = hit
        return flags

    def lipschitz_check(self, fmin, lipschitz, xc, u_c):
        from scipy.spatial.distance import cdist
        m, G = xc.shape[0], len(self.gps)
        flags = np.zeros((m, G), dtype=np.int32)
        unsafe = ~self.S
        if unsafe.any():
            d = cdist(xc, self.x[unsafe])
            for i in range(G):
                if fmin[i] == -np.inf:
                    continue
                flags[:, i] = np.any(u_c[:, [i]] - lipschitz[i] * d >= fmin[i], axis=1)
        return flags

    # -- the big passes of the expander loop on N ranks (SafeOpt._visit_in_big_passes_nrank) --
    def _behind(self, mode, cut_w, cut_idx):
        """Global rows and keys (width; mode & 1: minus the row index) of this shard's
        candidates strictly behind the cut -- pass_key of csrc/sets.hip."""
        idx = np.flatnonzero(self.cand) + self.lo
        key = -idx.astype(float) if (mode & 1) else self.w[idx - self.lo]
        keep = (key < cut_w) | ((key == cut_w) & (idx < cut_idx))
        return idx[keep], key[keep]

    def pass_hist(self, mode, cut_w, cut_idx, key_lo, key_hi, nbins=4096):
        _idx, key = self._behind(mode, cut_w, cut_idx)
        b = ((key - key_lo) * (float(nbins) / (key_hi - key_lo))).astype(np.int64)
        return np.bincount(np.clip(b, 0, nbins - 1), minlength=nbins).astype(np.uint32)

    def pass_list(self, mode, cut_w, cut_idx, thr, cap):
        idx, key = self._behind(mode, cut_w, cut_idx)
        keep = key >= thr
        idx, key = idx[keep], key[keep]                   # (row order)
        assert idx.size <= max(int(cap), 1), (idx.size, cap)
        li = idx - self.lo
        u = self.Q[li, 1::2]
        return idx, key, self.x[li], (u if (mode & 2) else u - self.mean[li])

    def pass_test(self, beta, fmin, xc, resid):
        xc = np.asarray(xc, dtype=float).reshape(-1, self.x.shape[1])
        mu_c = np.column_stack([gp.predict_noiseless(xc)[0].ravel() for gp in self.gps])
        return self.expander_check(beta, fmin, xc, mu_c, mu_c + np.asarray(resid))

    def pass_lipschitz_test(self, fmin, lipschitz, xc, u_c):
        xc = np.asarray(xc, dtype=float).reshape(-1, self.x.shape[1])
        return self.lipschitz_check(fmin, lipschitz, xc, np.asarray(u_c).reshape(xc.shape[0], -1))

    def sets_front(self, max_l, max_var, scaling, thr_beta):
        width = 0.0
        if max_var is None:
            width = self.maximizers(max_l)
            max_var = width / np.asarray(scaling)[0]
        nc, nu = self.candidates(max_var, scaling, thr_beta, False)
        w, idx = self.topk(0, np.inf, np.iinfo(np.int64).max, 1)
        d, G = self.x.shape[1], len(self.gps)
        x, mean, q = np.zeros(d), np.zeros(G), np.zeros(2 * G)
        out5 = np.array([width, nc, nu, -np.inf, -1.0, 0.0])
        if idx.size:
            xs, ms, _v, qs = self.gather_rows(idx)
            x, mean, q = xs[0], ms[0], qs[0]
            out5[3], out5[4] = w[0], float(idx[0])
            out5[5] = float(np.sum(self.cand & (self.w == w[0])))   # tied with it
        return out5, x, mean, q

    def sets_back(self, beta, fmin, xc, mu_c, u_c, near_frac, gidx_c, scaling, mark):
        flags = self.expander_check(beta, np.asarray(fmin), np.atleast_2d(xc),
                                    np.atleast_2d(mu_c), np.atleast_2d(u_c),
                                    near_frac)[0]
        active = np.asarray(fmin) != -np.inf
        if mark and active.any() and np.all(flags[active] != 0):
            self.mark_expanders([gidx_c])
        v, i = self.argmax(0, scaling)
        return flags, v, i

    def mark_expanders(self, gidx):
        self.G[np.asarray(gidx, dtype=np.int64) - self.lo] = True

    def unmark_expanders(self, gidx):
        self.G[np.asarray(gidx, dtype=np.int64) - self.lo] = False

    def candidate_widths(self):
        return self.cand.copy(), self.w.copy()

    def argmax(self, mode, scaling):
        if mode == 0:
            mask = self.M | self.G
            val = np.max((self.Q[:, 1::2] - self.Q[:, ::2]) / np.asarray(scaling), axis=1)
        else:
            mask = self.S
            val = self.Q[:, 1] if mode == 1 else self.Q[:, 0]
        if not mask.any():
            return -np.inf, -1
        ids = np.flatnonzero(mask)
        j = ids[np.argmax(val[mask])]
        return float(val[j]), int(j + self.lo)

    def download(self, what):
        return {Q_: self.Q, S_: self.S, M_: self.M, G_: self.G}[what]


def use_oracle_backend():
    """Install the NumPy stand-in as SafeOpt's grid backend (the product's hook for the
    CPU test-suite, ``safeopt_amd.gp_opt._BACKEND_FACTORY``); ``reset_backend`` undoes it
    (tests/conftest.py does so after every test)."""
    import safeopt_amd.gp_opt as go
    go._BACKEND_FACTORY = OracleGridBackend


def reset_backend():
    import safeopt_amd.gp_opt as go
    go._BACKEND_FACTORY = None
