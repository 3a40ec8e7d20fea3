"""SafeOpt / SafeOptSwarm on top of the MI355X HIP path.

Class surface = ``/root/reference/safeopt/gp_opt.py`` (constructor signatures,
method names, attribute names incl. the ``liptschitz`` spelling, exception
types).  What differs is where the arithmetic runs:

* ``SafeOpt.inputs`` lives in HBM (row-sharded over the ranks of a
  communicator), together with ``Q``, ``S``, ``M``, ``G``; the NumPy
  attributes of the same names are host mirrors that are refreshed lazily
  when read.
* ``update_confidence_intervals`` + ``compute_safe_set`` are ONE fused HIP
  kernel (``posterior_sweep``) per call; ``compute_sets`` /
  ``get_new_query_point`` are a handful of HBM-bound passes whose only host
  visible results are scalars.
* the expander loop (``gp_opt.py:557-612``) keeps its sequential semantics
  (first expander in descending-width order wins) but tests
  ``SGP_TOPK`` = 16 candidates per device pass with a closed-form rank-1
  posterior update instead of two O(n^3) re-factorisations per candidate.

The host code below is orchestration only; it never touches an ``(N, .)``
array unless the user reads one of the mirrors.
"""
from __future__ import annotations

import collections
import logging
from functools import partial

import weakref

import numpy as np

from . import _hip
from .dist import LocalComm, merge_argmax, merge_topk, shard_range
from .swarm import SwarmOptimization, DeviceSwarmOptimization

__all__ = ['SafeOpt', 'SafeOptSwarm']

_I64_MAX = np.iinfo(np.int64).max


def _one_per_gp(value, count):
    """``value`` as a 1-d array with one entry per GP: a list is taken as it
    is, anything else is repeated for every GP (``fmin``, ``lipschitz``)."""
    items = value if isinstance(value, list) else [value] * count
    return np.atleast_1d(np.asarray(items).squeeze())


def _prior_std(gps):
    """``sqrt(k(x, x))`` of every GP (stationary kernels: taken at the origin);
    this is what ``scaling='auto'`` means in the reference (gp_opt.py:80-83)."""
    origin = np.zeros((1, gps[0].input_dim))
    return np.sqrt(np.array([g.kern.Kdiag(origin)[0] for g in gps]))


def _with_context(x, context):
    """``x`` (m, d) with the context columns appended to every row."""
    context = np.atleast_2d(context)
    out = np.empty((x.shape[0], x.shape[1] + context.shape[1]), dtype=float)
    out[:, :x.shape[1]] = x
    out[:, x.shape[1]:] = context
    return out


class GaussianProcessOptimization(object):
    """What SafeOpt and SafeOptSwarm share: the GP handles, ``fmin``,
    ``beta(t)``, ``scaling``, ``threshold`` and the measurement log.

    Same constructor, attributes and methods as the base class of the
    reference (``gp_opt.py:30-278``): ``gps`` / ``gp``, ``fmin`` (one per GP),
    ``beta`` (always callable), ``scaling``, ``x`` / ``y`` / ``data`` / ``t``,
    ``add_new_data_point`` / ``remove_last_data_point``.
    """

    def __init__(self, gp, fmin, beta=2, num_contexts=0, threshold=0,
                 scaling='auto'):
        self.gps = gp if isinstance(gp, list) else [gp]
        if len(self.gps) > _hip.MAX_GPS:
            raise ValueError("at most %d GPs are supported" % _hip.MAX_GPS)
        self.gp = self.gps[0]                    # GP 0 = the objective
        n_gps = len(self.gps)

        self.fmin = _one_per_gp(fmin, n_gps)
        self.beta = beta if callable(beta) else (lambda t: beta)
        self.threshold = threshold
        self.num_contexts = num_contexts

        if isinstance(scaling, str) and scaling == 'auto':
            self.scaling = _prior_std(self.gps)
        else:
            self.scaling = np.asarray(scaling)
            if self.scaling.shape[0] != n_gps:
                raise ValueError("The number of scaling values should be "
                                 "equal to the number of GPs")

        # filled in by the subclasses that discretise the domain
        self._parameter_set = None
        self.bounds = None
        self.num_samples = 0

        # measurement log: every GP must start from the same inputs
        for other in self.gps[1:]:
            if not np.allclose(self.gp.X, other.X):
                raise NotImplementedError('The GPs have different '
                                          'measurements.')
        self._x = self.gp.X
        self._y = np.concatenate([g.Y for g in self.gps], axis=1)

    # -- measurement log ----------------------------------------------------------
    x = property(lambda self: self._x)
    y = property(lambda self: self._y)

    @property
    def data(self):
        """The measurements ``(x, y)`` seen so far."""
        return self._x, self._y

    @property
    def t(self):
        """Time step = number of measurements."""
        return self._x.shape[0]

    def plot(self, *args, **kwargs):
        raise NotImplementedError(
            "plotting is outside the accelerated path (SURVEY.md section 2, "
            "row 13); read opt.Q / opt.S / opt.M / opt.G and plot with "
            "matplotlib directly")

    def _add_context(self, x, context):
        return _with_context(x, context)

    def _add_data_point(self, gp, x, y, context=None):
        """Append ``(x, y)`` to one GP only (``self.x`` / ``self.y`` untouched).
        One more row than before: the handle turns this ``set_XY`` into a
        bordered update of the factor (``sgp_gp_append``)."""
        rows = x if context is None else _with_context(x, context)
        gp.set_XY(np.vstack([gp.X, rows]), np.vstack([gp.Y, y]))

    def add_new_data_point(self, x, y, context=None):
        """Log a measurement and hand it to the GPs; a ``nan`` in column ``i``
        of ``y`` means GP ``i`` did not observe it."""
        x, y = np.atleast_2d(x), np.atleast_2d(y)
        if self.num_contexts:
            x = _with_context(x, context)
        for i, gp in enumerate(self.gps):
            seen = ~np.isnan(y[:, i])
            if seen.any():
                self._add_data_point(gp, x[seen, :], y[seen, [i]])
        self._x = np.concatenate((self._x, x), axis=0)
        self._y = np.concatenate((self._y, y), axis=0)

    def _remove_last_data_point(self, gp):
        """Drop the newest row of one GP (a one-row ``sgp_gp_pop``)."""
        gp.set_XY(gp.X[:-1, :], gp.Y[:-1, :])

    def remove_last_data_point(self):
        """Undo the last ``add_new_data_point``: every GP that observed the
        newest measurement forgets it, then the log is shortened."""
        for gp, value in zip(self.gps, self._y[-1]):
            if not np.isnan(value):
                self._remove_last_data_point(gp)
        self._x, self._y = self._x[:-1, :], self._y[:-1, :]


#: Hook of the CPU test-suite (tests/_oracle_backend.py): a callable ``(gps, rows of
#: this rank, global offset) -> backend`` that stands in for ``_HipGridBackend`` so the
#: sharded host logic can run without a GPU.  None in the product.
_BACKEND_FACTORY = None


class _WriteBackView(np.ndarray):
    """View of the host mirror of ``Q`` (or of ``S`` / ``M`` / ``G``) that remembers
    element-wise writes: the reference mutates these arrays in place
    (``gp_opt.py:374-390, 475-476, 481, 505-506, 511, 615``) and user code may do the same
    (``opt.Q[:, 0] = ...``, ``opt.S[:] = ...``); the optimiser uploads the mirror before
    the next device pass that reads it.  (Writes that bypass ``__setitem__`` --
    ``np.add(..., out=opt.Q)`` -- are not seen: assign ``opt.Q = array`` for those.)"""

    _owner = None
    _field = 'Q'

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, '_owner', None)
        self._field = getattr(obj, '_field', 'Q')

    def __setitem__(self, key, value):
        owner = self._owner() if self._owner is not None else None
        field = self._field
        # only writes that land IN the mirror count: a copy of opt.Q, or an array
        # computed from it, inherits this class but not the memory
        if owner is not None and np.may_share_memory(self, getattr(owner, '_' + field)):
            # the mirror may lag behind the device (a sweep since this view was
            # taken): bring it up to date first -- in the reference the array is live
            owner._mirror(field, getattr(_hip, field))
            np.ndarray.__setitem__(self, key, value)
            if field == 'Q':
                owner._q_written = True
            else:
                owner._masks_written.add(field)
        else:
            np.ndarray.__setitem__(self, key, value)


#: what the front half of ``compute_sets`` found (``SafeOpt._front_*``): counts of
#: candidates / unsafe rows over the whole grid, the first candidate in visiting order
#: (width, global index, its row, means, intervals), the probe flags + arg-max when they
#: came with it (``fused``), and the number of candidates tied with it (None: unknown)
_Front = collections.namedtuple(
    '_Front', 'n_cand n_unsafe w_c idx_c x_c mu_c q_c fused n_tied')


class _HipGridBackend(object):
    """Rank-local device state of a ``SafeOpt``: the shard of ``inputs`` in
    HBM plus the GP handles.  (Tests drive the same phase interface with a
    NumPy stand-in to exercise the sharded host logic without a GPU.)"""

    def __init__(self, gps, inputs_shard, global_offset, ctx=None, axes=None):
        # the grid must live in the context of its GPs (stream ordering, device
        # pointers) and of the communicator (in-stream collectives)
        gp_ctx = getattr(gps[0], '_ctx', None)
        self.ctx = ctx or gp_ctx or _hip.Context.default()
        if gp_ctx is not None and gp_ctx is not self.ctx:
            raise ValueError("the GPs live on HIP device %d but the communicator"
                             " / grid context is device %d"
                             % (gp_ctx.device, self.ctx.device))
        self.gps = gps
        self.grid = _hip.DeviceGrid(self.ctx, inputs_shard, len(gps),
                                    global_offset)
        #: the rows are a tensor grid (what linearly_spaced_combinations builds): RBF
        #: kernels are swept through per-axis factor tables
        self.tensor_grid = self.grid.set_axes(axes)
        self.lo = int(global_offset)
        self.hi = self.lo + self.grid.N
        # which data version of every GP the resident mean/var reflect
        self._seen = [None] * len(gps)
        self._rank1_streak = 0
        #: after one new observation update the resident posterior in closed
        #: form (O(n) per row) instead of re-running the O(n^2) sweep; a full
        #: sweep is forced every `refresh_every` updates to bound the drift
        self.incremental = True
        self.refresh_every = 16

    def _dev(self):
        return [g._fitted() for g in self.gps]

    def _tags(self, devs):
        # (serial number of the device GP, its data version): serials are never
        # reused, unlike id(), so a re-created handle cannot alias an old tag
        return [(dv.serial, dv.version) for dv in devs]

    def posterior_is_current(self):
        """Do the resident mean / var describe the GPs as they are now?"""
        return self._seen == self._tags(self._dev())

    def refresh_posterior(self):
        """Resident mean / var of every GP from a fresh sweep, ``Q`` and ``S``
        untouched (after ``opt.Q = ...`` or a data change without an interval
        update: the GP expander test reads mean / var, gp_opt.py:585-602)."""
        devs = self._dev()
        self.grid.posterior(devs)
        self._seen = self._tags(devs)
        self._rank1_streak = 0

    def owns(self, gidx):
        return self.lo <= gidx < self.hi

    def set_context(self, c):
        self.grid.set_context(c)
        self._seen = [None] * len(self.gps)      # rows changed: full sweep

    def confidence(self, beta, fmin, defer=False):
        devs = self._dev()
        tags = self._tags(devs)
        which = [0] * len(devs)
        rank1 = self.incremental and self._rank1_streak < self.refresh_every
        for i, dv in enumerate(devs):
            if self._seen[i] == tags[i]:
                continue                                  # up to date
            if (rank1 and dv.appended and self._seen[i] is not None
                    and self._seen[i] == (dv.serial, dv.version - 1)):
                which[i] = 1                              # one append behind
            else:
                rank1 = False
        self._seen = [None] * len(devs)          # unknown until the call is through
        if rank1 and any(which):
            out = self.grid.rank1_update(devs, which, beta, fmin, defer)
            self._rank1_streak += 1
        else:
            out = self.grid.confidence(devs, beta, fmin, defer)
            self._rank1_streak = 0
        self._seen = tags
        return out

    def upload_Q(self, Q, fmin):
        self._seen = [None] * len(self.gps)      # mean/var no longer match Q
        return self.grid.upload_Q(Q, fmin)

    def upload_mask(self, what, mask):
        self.grid.upload_mask(what, mask)

    def maximizers(self, max_l):
        return self.grid.maximizers(max_l)

    def candidates(self, max_var, scaling, thr_beta, full_sets):
        return self.grid.candidates(max_var, scaling, thr_beta, full_sets)

    def topk(self, mode, cut_w, cut_idx, k):
        return self.grid.topk(mode, cut_w, cut_idx, k)

    def gather_rows(self, gidx):
        return self.grid.gather_rows(gidx)

    def expander_check(self, beta, fmin, xc, mu_c, u_c, near_frac=0.0):
        return self.grid.expander_check(self._dev(), beta, fmin, xc, mu_c, u_c,
                                        near_frac)

    def lipschitz_check(self, fmin, lipschitz, xc, u_c):
        return self.grid.lipschitz_check(fmin, lipschitz, xc, u_c)

    def expander_batch(self, beta, fmin, mode, cut_w, cut_idx, k):
        return self.grid.expander_batch(self._dev(), beta, fmin, mode, cut_w, cut_idx, k)

    def expander_pass(self, beta, fmin, mode, cut_w, cut_idx, key_lo, key_hi, want,
                      scaling=None):
        return self.grid.expander_pass(self._dev(), beta, fmin, mode, cut_w, cut_idx,
                                       key_lo, key_hi, want, scaling)

    def lipschitz_pass(self, fmin, lipschitz, mode, cut_w, cut_idx, key_lo, key_hi, want,
                       scaling=None):
        return self.grid.lipschitz_pass(fmin, lipschitz, mode, cut_w, cut_idx, key_lo, key_hi,
                                        want, scaling)

    def pass_lipschitz_test(self, fmin, lipschitz, xc, u_c):
        return self.grid.pass_lipschitz_test(fmin, lipschitz, xc, u_c)

    def pass_hist(self, mode, cut_w, cut_idx, key_lo, key_hi):
        return self.grid.pass_hist(mode, cut_w, cut_idx, key_lo, key_hi)

    def pass_list(self, mode, cut_w, cut_idx, thr, cap):
        return self.grid.pass_list(mode, cut_w, cut_idx, thr, cap)

    def pass_test(self, beta, fmin, xc, resid):
        return self.grid.pass_test(self._dev(), beta, fmin, xc, resid)

    def small_grid(self):
        """At most 16384 rows and 48 observations per GP: every candidate can be tested at
        once (``sgp_grid_expanders_small``)."""
        return (self.grid.N <= 16384 and not getattr(self.ctx, 'sweep_forced', False)
                and all(g._fitted().n <= 48 for g in self.gps))

    def expanders_small(self, beta, fmin, gidx):
        return self.grid.expanders_small(self._dev(), beta, fmin, gidx)

    def expanders_small_all(self, beta, fmin, cap=4096):
        return self.grid.expanders_small_all(self._dev(), beta, fmin, cap)

    def mark_expanders(self, gidx):
        self.grid.mark_expanders(gidx)

    def unmark_expanders(self, gidx):
        self.grid.unmark_expanders(gidx)

    def candidate_widths(self):
        """This shard's candidate mask and ``max_i(u_i - l_i)`` per row."""
        return self.grid.download(_hip.CAND), self.grid.download(_hip.WIDTH)

    # fused passes (one stream sync each)
    def sets_front(self, max_l, max_var, scaling, thr_beta):
        return self.grid.sets_front(max_l, max_var, scaling, thr_beta)

    def sets_front_comm(self, scaling, thr_beta):
        return self.grid.sets_front_comm(scaling, thr_beta)

    def sets_fused(self, beta, fmin, max_l, scaling, thr_beta, near_frac):
        return self.grid.sets_fused(self._dev(), beta, fmin, max_l, scaling,
                                    thr_beta, near_frac)

    def sets_fused_comm(self, beta, fmin, scaling, thr_beta, near_frac):
        return self.grid.sets_fused_comm(self._dev(), beta, fmin, scaling,
                                         thr_beta, near_frac)

    #: budget of the one-launch step, in per-thread instructions of its posterior phase
    #: (~16 cycles each with two waves per SIMD): above it the nine launches of the
    #: large-grid path (~90 us whatever the grid) are the faster way
    SMALL_STEP_BUDGET = 5000

    def small_step_devs(self):
        """The fitted device GPs when a whole step of this grid is better taken in ONE
        launch (``sgp_grid_step_small``: at most 16384 rows, GPs with at most 48
        observations -- and few enough of both that one workgroup is through in ~30 us:
        the 1000-point grid with 20 observations of the reference's examples), else None."""
        N = self.grid.N
        if N > 16384 or getattr(self.ctx, 'sweep_forced', False):
            return None
        devs = [g._fitted() for g in self.gps]
        cost = 0
        for dv in devs:
            n = dv.n
            if n > 48:
                return None
            cost += n * (30 + n // 2)
        if -(-N // 512) * cost > self.SMALL_STEP_BUDGET:     # (512 threads: step_small.hip)
            return None
        return devs

    def step_small(self, devs, beta, fmin, scaling, thr_beta):
        self._seen = [None] * len(devs)          # unknown until the call is through
        out = self.grid.step_small(devs, beta, fmin, scaling, thr_beta)
        self._seen = self._tags(devs)            # (a full sweep: the posterior is current)
        self._rank1_streak = 0
        return out

    def sets_back(self, beta, fmin, xc, mu_c, u_c, near_frac, gidx_c, scaling,
                  mark):
        return self.grid.sets_back(self._dev(), beta, fmin, xc, mu_c, u_c,
                                   near_frac, gidx_c, scaling, mark)

    def argmax(self, mode, scaling):
        return self.grid.argmax(mode, scaling)

    def download(self, what):
        return self.grid.download(what)


class SafeOpt(GaussianProcessOptimization):
    """Safe Bayesian optimisation over a discretised parameter set.

    Parameters
    ----------
    gp : GP handle or list of GP handles (``safeopt_amd.gpy``)
        First GP = objective, the others = safety constraints.
    parameter_set : 2d-array
        Candidate parameters, one per row (every rank passes the full set;
        each rank keeps its contiguous row block on its GPU).
    fmin : float or list of floats
        Safety thresholds (``-inf`` disables the constraint of a GP).
    lipschitz : float or list of floats, optional
    beta : float or callable ``beta(t)``
    num_contexts : int
    threshold : float or list of floats
    scaling : list of floats or ``'auto'``
    comm : communicator, optional
        ``safeopt_amd.dist.RcclComm`` for multi-GPU runs (default: 1 GPU).

    Examples
    --------
    >>> from safeopt_amd import SafeOpt, linearly_spaced_combinations
    >>> import safeopt_amd.gpy as GPy          # doctest: +SKIP
    >>> gp = GPy.models.GPRegression(np.array([[0.]]), np.array([[1.]]),
    ...                              noise_var=0.01**2)      # doctest: +SKIP
    >>> ps = linearly_spaced_combinations([[-1., 1.]], 100)
    >>> opt = SafeOpt(gp, ps, fmin=[0.])                      # doctest: +SKIP
    >>> x = opt.optimize()                                    # doctest: +SKIP
    >>> opt.add_new_data_point(x, np.array([[1.]]))           # doctest: +SKIP
    """

    def __init__(self, gp, parameter_set, fmin, lipschitz=None, beta=2,
                 num_contexts=0, threshold=0, scaling='auto', comm=None,
                 ):
        super(SafeOpt, self).__init__(gp, fmin=fmin, beta=beta,
                                      num_contexts=num_contexts,
                                      threshold=threshold, scaling=scaling)
        # candidate rows = parameters (+ zeroed context columns that the context
        # setter fills); `parameter_set` stays a view of the parameter columns
        params = np.asarray(parameter_set)
        if self.num_contexts:
            self.inputs = np.hstack(
                (params, np.zeros((params.shape[0], self.num_contexts),
                                  dtype=params.dtype)))
            self.parameter_set = self.inputs[:, :params.shape[1]]
        else:
            self.inputs = self.parameter_set = params

        # attribute name as spelled by the reference (gp_opt.py:366)
        self.liptschitz = (None if lipschitz is None
                           else _one_per_gp(lipschitz, len(self.gps)))
        self._use_lipschitz = lipschitz is not None

        # host mirrors of the resident arrays, with the reference's dtypes and
        # shapes (gp_opt.py:374-390); refreshed from HBM only when read
        N = self.inputs.shape[0]
        self._Q = np.empty((N, 2 * len(self.gps)), dtype=float)
        self._S, self._M, self._G = (np.zeros(N, dtype=bool) for _ in range(3))
        self._stale = dict(Q=False, S=False, M=False, G=False)
        self._s_user = False       # the device's S was assigned by user code (opt.S[...] = ...)
        self._masks_written = set()
        self._q_written = False

        # this rank's contiguous block of rows, resident on its GPU
        self._comm = comm if comm is not None else LocalComm()
        if N < self._comm.world:
            # every rank owns at least one row (a shard without rows has no resident
            # state to launch on; raised on ALL ranks, before any collective)
            raise ValueError("parameter_set has %d rows for %d ranks" % (N, self._comm.world))
        self._shard = shard_range(N, self._comm.rank, self._comm.world)
        lo, hi = self._shard
        if _BACKEND_FACTORY is not None:          # CPU tests: NumPy stand-in
            self._backend = _BACKEND_FACTORY(self.gps, self.inputs[lo:hi], lo)
        else:
            self._backend = _HipGridBackend(self.gps, self.inputs[lo:hi], lo,
                                            ctx=getattr(self._comm, 'ctx', None),
                                            axes=_hip.tensor_grid_axes(self.inputs))
        # every rank must sweep with the same arithmetic (the N-rank run equals the one-rank
        # run bit for bit): a parameter set that is a tensor grid on some shards only -- the
        # device check is per shard -- is a tensor grid on none
        if self._comm.world > 1 and hasattr(self._backend, 'tensor_grid'):
            bad = self._comm.allreduce_max(
                np.array([0.0 if self._backend.tensor_grid else 1.0]))[0] > 0
            if bad and self._backend.tensor_grid:
                self._backend.grid.clear_axes()
                self._backend.tensor_grid = False
        self._any_safe = False
        self._max_l = -np.inf
        self._ci_fresh = False
        self._argmax_cache = None
        #: small grids with few observations take the whole step in one launch
        #: (``sgp_grid_step_small``); False keeps them on the large-grid path
        self.small_step = True
        self._thr_beta_key = self._thr_beta = None

    # -- host mirrors ---------------------------------------------------------
    def _mirror(self, name, what):
        if self._stale[name]:
            arr = getattr(self, '_' + name)
            part = self._backend.download(what)
            if self._comm.world == 1:
                np.copyto(arr, part)
            else:
                lo, hi = self._shard
                counts = [np.subtract(*shard_range(arr.shape[0], r,
                                                   self._comm.world)[::-1])
                          for r in range(self._comm.world)]
                pad = max(counts)
                buf = np.zeros((pad,) + arr.shape[1:], dtype=arr.dtype)
                buf[:hi - lo] = part
                allp = self._comm.allgather(buf)
                off = 0
                for r, c in enumerate(counts):
                    arr[off:off + c] = allp[r][:c]
                    off += c
            self._stale[name] = False
        # (the properties hand out write-back views of this array)
        return getattr(self, '_' + name)

    @property
    def Q(self):
        """Confidence intervals ``[l_0, u_0, l_1, u_1, ...]`` per row.  Writable:
        element-wise writes are uploaded before the next pass that reads them."""
        return self._view('Q', _hip.Q)

    def _view(self, name, what):
        view = self._mirror(name, what).view(_WriteBackView)
        view._owner = weakref.ref(self)
        view._field = name
        return view

    def _flush_Q(self):
        """Upload ``Q`` if user code wrote into the mirror since the last pass."""
        if self._q_written:
            self._q_written = False
            self.Q = self._Q.copy()

    @Q.setter
    def Q(self, value):
        """Assigning ``opt.Q`` uploads the intervals and recomputes ``S``."""
        value = np.asarray(value, dtype=float).reshape(self._Q.shape)
        lo, hi = self._shard
        m, a = self._backend.upload_Q(value[lo:hi], self.fmin)
        red = self._comm.allreduce_max(np.array([m, float(a)]))
        self._max_l, self._any_safe = red[0], bool(red[1] > 0)
        np.copyto(self._Q, value)
        self._q_written = False
        self._stale.update(Q=False, S=True)
        self._ci_fresh = True
        self._argmax_cache = None

    @property
    def S(self):
        """Safe set mask.  Writable like in the reference (``opt.S[:] = ...``): the next
        ``get_new_query_point`` reads the edited mask; ``compute_sets`` recomputes it from
        the intervals, as the reference's ``compute_safe_set`` does (gp_opt.py:478-481)."""
        return self._view('S', _hip.S)

    @property
    def M(self):
        """Potential maximisers mask (writable; ``compute_sets`` recomputes it)."""
        return self._view('M', _hip.M)

    @property
    def G(self):
        """Expanders mask (writable; ``compute_sets`` recomputes it)."""
        return self._view('G', _hip.G)

    def _flush_masks(self, for_sets=False):
        """Element-wise writes into ``opt.S / M / G`` since the last pass reach the device.
        ``for_sets``: ``compute_sets`` is about to overwrite M and G and recomputes S from
        the intervals (gp_opt.py:478-481, 505-615): an edited S is dropped, not uploaded."""
        written, self._masks_written = self._masks_written, set()
        if for_sets and (self._s_user or 'S' in written):
            # (an S that user code assigned -- now or before an earlier arg-max -- is
            # recomputed from the intervals, like compute_safe_set does in the reference)
            self._s_user = False
            self._argmax_cache = None
            self.Q = self._mirror('Q', _hip.Q).copy()
            return
        if not written:
            return
        self._argmax_cache = None
        if for_sets:
            return
        lo, hi = self._shard
        for name in sorted(written):
            self._backend.upload_mask(getattr(_hip, name), getattr(self, '_' + name)[lo:hi])
        if 'S' in written:
            self._any_safe = bool(self._S.any())
            self._s_user = True

    # -- reference properties ---------------------------------------------------
    @property
    def use_lipschitz(self):
        """True: expanders are certified with the Lipschitz constant, False:
        with the GP confidence intervals."""
        return self._use_lipschitz

    @use_lipschitz.setter
    def use_lipschitz(self, value):
        if value and self.liptschitz is None:
            raise ValueError('Lipschitz constant not defined')
        self._use_lipschitz = value

    @property
    def parameter_set(self):
        """Discrete parameter samples (without context columns)."""
        return self._parameter_set

    @parameter_set.setter
    def parameter_set(self, parameter_set):
        cols = parameter_set.T
        self._parameter_set = parameter_set
        self.bounds = [(c.min(), c.max()) for c in cols]
        self.num_samples = [np.unique(c).size for c in cols]

    def _context_columns(self):
        return self.inputs[0, self.inputs.shape[1] - self.num_contexts:]

    @property
    def context_fixed_inputs(self):
        """``[(column, value), ...]`` of the inputs the current context fixes
        (counted down from the last GP input, as the reference's plots expect)."""
        if self.num_contexts > 0:
            last = self.gp.input_dim - 1
            return [(last - k, v) for k, v in enumerate(self._context_columns())]

    @property
    def context(self):
        """Current context variables."""
        if self.num_contexts:
            return self._context_columns()

    @context.setter
    def context(self, context):
        if not self.num_contexts:
            return
        if context is None:
            raise ValueError('Need to provide value for context.')
        self.inputs[:, self.inputs.shape[1] - self.num_contexts:] = context
        self._backend.set_context(np.asarray(self._context_columns(), dtype=float))
        self._ci_fresh = False

    # -- the hot path -------------------------------------------------------------
    def update_confidence_intervals(self, context=None, _defer=False):
        """Posterior sweep of every GP over all candidates -> ``Q`` (and ``S``).

        One fused kernel per rank; the only values that come back are
        ``max(l_0[S])`` and ``any(S)``.  (``_defer``: internal to
        :meth:`optimize` -- the sweep is only enqueued and the two scalars
        arrive with the set passes, one device round trip for the whole step.)
        """
        beta = self.beta(self.t)
        self.context = context
        if _defer:
            m, a = self._backend.confidence(beta, self.fmin, defer=True)
        else:
            m, a = self._backend.confidence(beta, self.fmin)
        if m is None:
            self._max_l, self._any_safe = None, None      # known after the sets
        else:
            red = self._comm.allreduce_max(np.array([m, float(a)]))
            self._max_l, self._any_safe = red[0], bool(red[1] > 0)
        self._stale.update(Q=True, S=True)
        self._q_written = False          # (the sweep overwrites the intervals)
        self._s_user = False
        self._ci_fresh = True
        self._argmax_cache = None

    def compute_safe_set(self):
        """``S = all(l_i > fmin_i)``; fused into the sweep, nothing to redo."""
        self._flush_Q()
        self._flush_masks(for_sets=True)
        if not self._ci_fresh:
            self.update_confidence_intervals(context=self.context)

    def _sum_over_ranks(self, values):
        v = np.asarray(values, dtype=np.float64)
        return self._comm.allgather(v).sum(axis=0)

    def compute_sets(self, full_sets=False):
        """Maximisers ``M`` and expanders ``G`` from the current intervals.

        ``full_sets=True`` evaluates every safe point as an expander candidate
        (plotting mode of the reference).

        The common case -- GP certificates, not ``full_sets`` -- runs as a FRONT half
        (maximisers, candidate mask, the first candidate in visiting order) and the
        certification of that candidate; which entry points carry the front half depends
        on where the step runs (one method each):

        ==========================  =====================================================
        one rank                    ``_front_one_rank_fused``: both halves in one device
                                    round trip (``sgp_grid_sets_fused``)
        one rank, no GP active      ``_front_one_rank``
        N ranks, in-stream comm     ``_front_n_ranks_in_stream``: one round trip, merges on
                                    the device behind the collectives
        N ranks otherwise           ``_front_n_ranks_host``: packed host collectives
        ==========================  =====================================================
        """
        beta = self.beta(self.t)
        self.compute_safe_set()
        be = self._backend
        G = len(self.gps)
        # the GP expander test updates the RESIDENT posterior in closed form
        # (the reference re-predicts from the GP, gp_opt.py:585-602): bring it
        # up to date if the intervals were assigned by hand or the data changed
        if (not self.use_lipschitz and hasattr(be, 'posterior_is_current')
                and not be.posterior_is_current()):
            be.refresh_posterior()
        thr_beta = np.broadcast_to(
            np.asarray(self.threshold, dtype=float) * beta, (G,)).copy()

        self._argmax_cache = None
        if self._max_l is not None and not self._any_safe:
            # M = G = False everywhere
            be.maximizers(np.inf)
            be.candidates(np.inf, self.scaling, thr_beta, False)
            self._stale.update(M=True, G=True)
            return

        active = self.fmin != -np.inf
        world = self._comm.world
        if full_sets or self.use_lipschitz or not hasattr(be, 'sets_front'):
            return self._general_sets(beta, active, thr_beta, full_sets)
        if world == 1 and np.any(active) and hasattr(be, 'sets_fused'):
            front = self._front_one_rank_fused(beta, thr_beta)
        elif world == 1:
            front = self._front_one_rank(thr_beta)
        elif (self._max_l is None and np.any(active) and hasattr(be, 'sets_fused_comm')):
            front = self._front_n_ranks_in_stream(beta, thr_beta)
        else:
            front = self._front_n_ranks_host(thr_beta)
        if front is not None:                 # (None: no safe row, M = G = False)
            self._certify_first_candidate(beta, active, front)

    # -- the front half of compute_sets, one method per place the step runs ----------
    @staticmethod
    def _front_of(out5, x_c, mu_c, q_c, fused):
        return _Front(out5[1], out5[2], float(out5[3]), int(out5[4]), x_c, mu_c, q_c,
                      fused, int(out5[5]) if len(out5) > 5 else None)

    def _deferred_max_l(self, max_l):
        """``max l0[S]`` arrives with the front half after a deferred confidence pass;
        False when no row is safe (the passes ran with ``-inf``: ``M = G = False``)."""
        self._max_l, self._any_safe = max_l, bool(max_l > -np.inf)
        if not self._any_safe:
            self._stale.update(M=True, G=True)
        return self._any_safe

    def _front_one_rank_fused(self, beta, thr_beta):
        """One rank: both halves in one device round trip."""
        (out5, x_c, mu_c, q_c, f_flags, f_val, f_idx,
         max_l) = self._backend.sets_fused(beta, self.fmin, self._max_l, self.scaling,
                                           thr_beta, 0.5)
        if self._max_l is None and not self._deferred_max_l(max_l):
            return None
        return self._front_of(out5, x_c, mu_c, q_c, (f_flags, f_val, f_idx))

    def _front_one_rank(self, thr_beta):
        """One rank without an active GP certificate: the front half alone."""
        out5, x_c, mu_c, q_c = self._backend.sets_front(self._max_l, None, self.scaling,
                                                        thr_beta)
        return self._front_of(out5, x_c, mu_c, q_c, None)

    def _front_n_ranks_in_stream(self, beta, thr_beta):
        """N ranks behind a deferred confidence pass, communicator in stream: the whole
        certified step in one device round trip -- the first-candidate merge, the probe
        flags and the arg-max merge run on the device behind the collectives, every rank
        reads back the same (global) numbers (gp_opt.py:511-513, 542-557, 611-612, 642-644)."""
        (out5, x_c, mu_c, q_c, f_flags, f_val, f_idx,
         max_l) = self._backend.sets_fused_comm(beta, self.fmin, self.scaling, thr_beta, 0.5)
        if not self._deferred_max_l(max_l):
            return None
        return self._front_of(out5, x_c, mu_c, q_c, (f_flags, f_val, f_idx))

    def _front_n_ranks_host(self, thr_beta):
        """N ranks, collectives on the host side of the step: every rank's first candidate
        with its rows goes through one packed all-gather and is merged in NumPy."""
        be = self._backend
        d, G = self.inputs.shape[1], len(self.gps)
        if self._max_l is None:
            # deferred confidence pass: max l0 and the maximiser width
            # are all-reduced in stream, one round trip for this half
            out5, x_l, mu_l, q_l, max_l = be.sets_front_comm(self.scaling, thr_beta)
            if not self._deferred_max_l(max_l):
                return None
        else:
            width = self._comm.allreduce_max(np.array([be.maximizers(self._max_l)]))[0]
            out5, x_l, mu_l, q_l = be.sets_front(self._max_l, width / self.scaling[0],
                                                 self.scaling, thr_beta)
        tied_l = out5[5] if len(out5) > 5 else np.nan
        pk = self._comm.allgather(np.concatenate([out5[1:5], [tied_l], x_l, mu_l, q_l]))
        n_cand, n_unsafe = pk[:, 0].sum(), pk[:, 1].sum()
        w_b, i_b = merge_topk(pk[:, 2], pk[:, 3].astype(np.int64), 1)
        idx_c = int(i_b[0]) if i_b.size else -1
        w_c = float(w_b[0]) if i_b.size else -np.inf
        r = int(np.flatnonzero(pk[:, 3].astype(np.int64) == idx_c)[0]) if idx_c >= 0 else 0
        x_c, mu_c, q_c = pk[r, 5:5 + d], pk[r, 5 + d:5 + d + G], pk[r, 5 + d + G:]
        # candidates of ALL shards tied with the first one: a shard whose
        # own first candidate is narrower holds none of that width
        holds = (pk[:, 3] >= 0) & (pk[:, 2] == w_c)
        tied = pk[holds, 4].sum()
        return _Front(n_cand, n_unsafe, w_c, idx_c, x_c, mu_c, q_c, None,
                      None if np.isnan(tied) else int(tied))

    def _certify_first_candidate(self, beta, active, front, exact=False):
        """Second half, shared by every variant: is the first candidate an expander
        (gp_opt.py:579-612)?  ``front.fused``: the probe flags and the arg-max came with the
        front half (already global); otherwise ``sets_back`` runs them now.  ``exact``: the
        flags are those of the scan over ALL unsafe rows, not of the probe."""
        be = self._backend
        G = len(self.gps)
        world = self._comm.world
        n_cand, n_unsafe, w_c, idx_c, x_c, mu_c, q_c, fused, n_tied = front
        self._stale.update(M=True, G=True)
        if n_cand == 0 or n_unsafe == 0 or not np.any(active) or idx_c < 0:
            if fused is not None:       # G untouched: the arg-max over M holds
                self._argmax_cache = (fused[1], int(fused[2]))
            return
        if fused is not None:
            flags, val, idx = fused
        else:
            flags, val, idx = be.sets_back(beta, self.fmin, x_c, mu_c, q_c[1::2], 0.5,
                                           idx_c, self.scaling, world == 1)
        merged = fused is not None       # (flags and arg-max already global)
        if world > 1 and not merged:
            pk = self._comm.allgather(np.concatenate(
                [flags.astype(np.float64), [val, float(idx)]]))
            flags = pk[:, :G].max(axis=0)
        if np.all(flags[active] != 0):
            if world > 1 and not merged:
                if be.owns(idx_c):
                    be.mark_expanders(np.array([idx_c], dtype=np.int64))
                # arg-max over M on every rank + the certified expander
                v_c = np.max((q_c[1::2] - q_c[::2]) / self.scaling)
                val, idx = merge_argmax(
                    np.append(pk[:, G], v_c),
                    np.append(pk[:, G + 1].astype(np.int64), idx_c))
            self._argmax_cache = (val, int(idx))
            if self._settle_ties(beta, active, w_c, idx_c, n_tied) != idx_c:
                self._argmax_cache = None
            return
        # not certified by the probe.  One rank with big passes: the exact test of this candidate
        # rides in the first pass (the cut in FRONT of it) -- a candidate that lifts no row close
        # to it rarely lifts a far one, and a pass costs little more than its scan
        if (not exact and world == 1 and self.big_passes and not self.use_lipschitz
                and hasattr(be, 'expander_pass') and np.isfinite(w_c)):
            return self._visit_in_big_passes(beta, active, False, w_c, idx_c + 1)
        # ... otherwise: exact scan, then the general loop
        hit = [False] if exact else self._expander_flags(
            beta, x_c[None, :], mu_c[None, :], q_c[None, 1::2], active, probe=False)
        if hit[0]:
            if be.owns(idx_c):
                be.mark_expanders(np.array([idx_c], dtype=np.int64))
            self._settle_ties(beta, active, w_c, idx_c, n_tied)
            return
        self._visit_candidates(beta, active, False, w_c, idx_c)

    def _general_sets(self, beta, active, thr_beta, full_sets):
        """Step by step (``full_sets``, Lipschitz certificates, backends without the fused
        passes): maximisers, candidate mask, then the expander loop from its start."""
        be = self._backend
        width = self._comm.allreduce_max(
            np.array([be.maximizers(self._max_l)]))[0]
        max_var = width / self.scaling[0]
        n_cand, n_unsafe = self._sum_over_ranks(
            be.candidates(max_var, self.scaling, thr_beta, full_sets))
        self._stale.update(M=True, G=True)

        if n_cand == 0 or n_unsafe == 0 or not np.any(active):
            # no candidate, or nothing unsafe to certify (any([]) is False),
            # or no safety constraint at all: G stays empty
            return
        self._visit_candidates(beta, active, full_sets, np.inf,
                               -1 if full_sets else _I64_MAX)

    def _visit_candidates(self, beta, active, full_sets, cut_w, cut_idx):
        """Expander loop of gp_opt.py:557-612 from the cut onwards."""
        be = self._backend
        G = len(self.gps)
        mode = 1 if full_sets else 0
        # The first expander in visiting order is very often the very first
        # candidate, so the first pass fetches and tests only that one; later
        # passes take SGP_TOPK candidates at a time.
        K = _hip.TOPK if (full_sets or cut_idx != _I64_MAX) else 1
        if (self._comm.world == 1 and not self.use_lipschitz and self.small_step
                and hasattr(be, 'expanders_small') and be.small_grid()):
            return self._visit_all_candidates(beta, active, full_sets, cut_idx)
        if (self._comm.world == 1 and not self.use_lipschitz
                and hasattr(be, 'expander_batch')):
            # one rank: a pass of the loop -- the next K candidates, their rows, the exact
            # test -- is ONE device round trip
            big = hasattr(be, 'expander_pass') and self.big_passes
            if big and full_sets:
                return self._visit_in_big_passes(beta, active, True, np.inf, -1)
            if big and np.isfinite(cut_w):
                # behind a first candidate that is no expander: a pass of 256 candidates costs
                # less than the 16 of sgp_grid_expander_batch (whose scan knows neither the
                # block test nor the posterior Cauchy-Schwarz bound)
                return self._visit_in_big_passes(beta, active, False, cut_w, cut_idx)
            while True:
                w_b, i_b, fl = be.expander_batch(beta, self.fmin, mode, cut_w, cut_idx, K)
                m = i_b.size
                if m == 0:
                    break
                is_exp = np.all(fl[:, active] != 0, axis=1)
                if full_sets:
                    be.mark_expanders(i_b[is_exp])
                elif is_exp.any():
                    first = int(np.argmax(is_exp))
                    be.mark_expanders(i_b[first:first + 1])
                    self._settle_ties(beta, active, float(w_b[first]), int(i_b[first]))
                    break
                if m < K:
                    break
                cut_w, cut_idx = float(w_b[-1]), int(i_b[-1])
                if big and K == _hip.TOPK:
                    # SGP_TOPK candidates in, no expander: from here on hundreds, then
                    # thousands of candidates per pass
                    return self._visit_in_big_passes(beta, active, False, cut_w, cut_idx)
                K = _hip.TOPK
            return
        big_n = (self._comm.world > 1 and self.big_passes and
                 hasattr(be, 'pass_lipschitz_test' if self.use_lipschitz else 'pass_test'))
        if big_n and full_sets:
            return self._visit_in_big_passes_nrank(beta, active, True, np.inf, -1)
        # Lipschitz certificates, one rank: the same big passes (sgp_grid_lipschitz_pass)
        big_l = (self._comm.world == 1 and self.use_lipschitz and self.big_passes
                 and hasattr(be, 'lipschitz_pass'))
        if big_l and full_sets:
            return self._visit_in_big_passes(beta, active, True, np.inf, -1)
        if big_l and np.isfinite(cut_w):
            return self._visit_in_big_passes(beta, active, False, cut_w, cut_idx)
        while True:
            w_loc, i_loc = be.topk(mode, cut_w, cut_idx, K)
            if self._comm.world > 1:
                wp = np.full(K, -np.inf)
                ip = np.full(K, -1, dtype=np.int64)
                wp[:w_loc.size] = w_loc
                ip[:i_loc.size] = i_loc
                w_b, i_b = merge_topk(self._comm.allgather(wp),
                                      self._comm.allgather(ip), K,
                                      by_index=full_sets)
            else:
                w_b, i_b = w_loc, i_loc
            m = i_b.size
            if m == 0:
                break

            # rows of the candidates from their owners
            own = np.array([be.owns(int(i)) for i in i_b])
            xc = np.zeros((m, self.inputs.shape[1]))
            mu_c = np.zeros((m, G))
            u_c = np.zeros((m, G))
            if own.any():
                x_o, mean_o, _var_o, Q_o = be.gather_rows(i_b[own])
                xc[own], mu_c[own], u_c[own] = x_o, mean_o, Q_o[:, 1::2]
            if self._comm.world > 1:
                packed = np.concatenate([xc, mu_c, u_c], axis=1)
                packed = self._comm.allgather(packed).sum(axis=0)
                d = self.inputs.shape[1]
                xc, mu_c, u_c = (packed[:, :d], packed[:, d:d + G],
                                 packed[:, d + G:])

            is_exp = self._expander_flags(beta, xc, mu_c, u_c, active,
                                          probe=not full_sets)

            if full_sets:
                mine = [int(i) for i, e, o in zip(i_b, is_exp, own) if e and o]
                be.mark_expanders(np.asarray(mine, dtype=np.int64))
            elif is_exp.any():
                first = int(np.argmax(is_exp))
                if own[first]:
                    be.mark_expanders(i_b[first:first + 1])
                self._settle_ties(beta, active, float(w_b[first]), int(i_b[first]))
                break
            if m < K:
                break
            cut_w, cut_idx = float(w_b[-1]), int(i_b[-1])
            if big_n and K == _hip.TOPK:
                return self._visit_in_big_passes_nrank(beta, active, False, cut_w, cut_idx)
            if big_l:
                # behind a first candidate that is no expander
                return self._visit_in_big_passes(beta, active, False, cut_w, cut_idx)
            K = _hip.TOPK

    #: candidates per pass of ``_visit_in_big_passes`` (the last entry repeats); None: by the
    #: number of observations (``_pass_size``)
    pass_sizes = None
    #: False: the expander loop of a large grid stays at SGP_TOPK candidates per round trip
    big_passes = True

    def _pass_size(self, k):
        """Candidates of pass k.  A pass costs a scan of the unsafe rows whatever its size
        (0.1-0.2 ms at 1e6 rows) plus the candidates' operands, 4 n^2 flop each: the first pass
        takes what keeps the operands at about 1e9 flop -- 8192 candidates up to n = 250, 1024
        at n = 640, 256 from n = 1000 on, so that an expander among the first candidates stays
        cheap to find --, the following ones 8 times as many each, up to 8192."""
        if self.pass_sizes is not None:
            return self.pass_sizes[min(k, len(self.pass_sizes) - 1)]
        if self.use_lipschitz:
            return 8192                 # (no operands: the distance test alone)
        n = max(int(gp.X.shape[0]) for gp in self.gps)
        first = 256
        while first < 8192 and 2 * first * n * n <= 5e8:
            first *= 2
        return min(8192, first * 8 ** min(k, 2))

    def _visit_in_big_passes(self, beta, active, full_sets, cut_w, cut_idx):
        """The expander loop of gp_opt.py:557-612 behind the cut, one rank, where it goes
        far -- no expander among the first SGP_TOPK candidates (a converged run has none at
        all), or ``full_sets`` (:553-555: every safe row is visited).  A pass takes the next
        few hundred to few thousand candidates in visiting order (``pass_sizes``; chosen on
        the device by a histogram of the widths, not a sort) and tests ALL of them in one
        scan of the unsafe rows (``sgp_grid_expander_pass``): the first expander in visiting
        order is the widest hit of the first pass that has one -- every candidate in front of
        it has been tested -- and exact ties among equal widths are settled as always."""
        be = self._backend
        if full_sets:
            n_all = float(self.inputs.shape[0])
            lo, hi, mode = -(n_all + 1.0), 1.0, 1           # keys: minus the row index
        else:
            lo, hi, mode = 0.0, float(cut_w), 0             # keys: the interval widths
        for k in range(1 << 30):
            want = self._pass_size(k)
            # (the arg-max of the step comes back with the result of the pass: when the pass ends
            # the loop without an expander G is final and the next query point is known)
            sc = None if full_sets else self.scaling
            if self.use_lipschitz:
                tested, hits, key, row, left, amax = be.lipschitz_pass(
                    self.fmin, self.liptschitz, mode, cut_w, cut_idx, lo, hi, want, sc)
            else:
                tested, hits, key, row, left, amax = be.expander_pass(
                    beta, self.fmin, mode, cut_w, cut_idx, lo, hi, want, sc)
            if hits and not full_sets:
                be.mark_expanders(np.array([row], dtype=np.int64))
                self._settle_ties(beta, active, key, row)
                return
            if tested == 0 or left == -np.inf:
                if amax >= 0 and tested > 0:
                    self._argmax_cache = (None, amax)
                return
            cut_w, cut_idx = left, -1
            if not full_sets:
                hi = left

    def _visit_in_big_passes_nrank(self, beta, active, full_sets, cut_w, cut_idx):
        """``_visit_in_big_passes`` on a row-sharded grid.  Per pass: the ranks sum their
        histograms of the keys behind the cut and pick ONE threshold (every rank computes the
        same one from the same sum); every rank lists its candidates above it and the lists are
        gathered -- the same candidates in the same order everywhere; every rank tests ALL of
        them against its own unsafe rows (the operands of the test are recomputed on every rank:
        a few MFLOP per candidate against the scan) and the flags are or-ed over the ranks
        (gp_opt.py:602: ``np.any`` over all unsafe rows).  Three small collectives per pass
        instead of three per 16 candidates."""
        be, comm = self._backend, self._comm
        G, d = len(self.gps), self.inputs.shape[1]
        if full_sets:
            n_all = float(self.inputs.shape[0])
            lo, hi, mode = -(n_all + 1.0), 1.0, 1
        else:
            lo, hi, mode = 0.0, float(cut_w), 0
        nbins = 4096
        for k in range(1 << 30):
            want = self._pass_size(k)
            hist = comm.allgather(be.pass_hist(mode, cut_w, cut_idx, lo, hi).astype(np.float64))
            from_top = np.cumsum(hist.sum(axis=0)[::-1])
            if from_top[-1] == 0:
                return
            # the highest bin at which the count from the top reaches `want` (k_pass_pick)
            b = nbins - 1 - int(np.argmax(from_top >= want)) if from_top[-1] >= want else 0
            thr = -np.inf if b == 0 else lo + (hi - lo) * (float(b) / nbins)
            # (rounding at a bin edge can move a few candidates across it: room for the
            # bin below as well)
            cap = int(from_top[min(nbins - 1, nbins - b)] if b > 0 else from_top[-1]) + 64
            # (Lipschitz certificates: the candidates' upper bounds themselves, mode | 2)
            gi, key, xc, resid = be.pass_list(mode | (2 if self.use_lipschitz else 0), cut_w,
                                              cut_idx, thr, cap)
            counts = comm.allgather(np.array([float(gi.size)]))[:, 0].astype(int)
            pad = int(counts.max())
            if pad == 0:
                return
            buf = np.zeros((pad, 2 + d + G))
            buf[:gi.size, 0], buf[:gi.size, 1] = gi, key            # (row indices < 2^53)
            buf[:gi.size, 2:2 + d], buf[:gi.size, 2 + d:] = xc, resid
            allp = comm.allgather(buf)
            rows = np.concatenate([allp[r][:c] for r, c in enumerate(counts)])
            if self.use_lipschitz:
                flags = be.pass_lipschitz_test(self.fmin, self.liptschitz, rows[:, 2:2 + d],
                                               rows[:, 2 + d:])
            else:
                flags = be.pass_test(beta, self.fmin, rows[:, 2:2 + d], rows[:, 2 + d:])
            flags = comm.allreduce_max(flags.astype(np.float64)) > 0
            hits = np.all(flags[:, active], axis=1)
            gidx_all = rows[:, 0].astype(np.int64)
            if full_sets:
                mine = [int(i) for i in gidx_all[hits] if be.owns(int(i))]
                for a in range(0, len(mine), 4096):
                    be.mark_expanders(np.asarray(mine[a:a + 4096], dtype=np.int64))
            elif hits.any():
                hk, hg = rows[hits, 1], gidx_all[hits]
                first = np.lexsort((hg, hk))[-1]             # widest, then the larger row
                row, w_star = int(hg[first]), float(hk[first])
                if be.owns(row):
                    be.mark_expanders(np.array([row], dtype=np.int64))
                self._settle_ties(beta, active, w_star, row)
                return
            if thr == -np.inf:
                return
            cut_w, cut_idx = thr, -1
            if not full_sets:
                hi = thr

    def _visit_all_candidates(self, beta, active, full_sets, cut_idx, chunk=1024):
        """The expander loop of a SMALL grid on one rank: every candidate is tested at once
        (two launches, one round trip per ``chunk`` candidates) and the flags are walked in
        the reference's own visiting order -- ``argsort()[::-1]`` of the candidate widths
        (gp_opt.py:542-552), exact ties included, so nothing is left to settle.  ``cut_idx``:
        the first candidate of the device's order when it is already known to be no expander
        (it is skipped wherever the reference's order puts it)."""
        be = self._backend
        if not full_sets and hasattr(be, 'expanders_small_all'):
            # one round trip: the device lists the candidates in row order and tests ALL of
            # them; rows, widths and flags come back together (up to 4096 candidates)
            got = be.expanders_small_all(beta, self.fmin)
            if got is not None:
                rows, w_rows, flags = got
                order = w_rows.argsort()[::-1]         # (the reference's width[rows].argsort()[::-1])
                if 0 <= cut_idx < _I64_MAX:
                    order = order[rows[order] != cut_idx]
                is_exp = np.all(flags[order][:, active] != 0, axis=1)
                if is_exp.any():
                    be.mark_expanders(rows[order[int(np.argmax(is_exp))]][None])
                    self._argmax_cache = None
                return
        cand, width = be.candidate_widths()
        rows = np.flatnonzero(np.asarray(cand, dtype=bool))
        if full_sets:
            order = rows                                   # natural order, no early exit
        else:
            order = rows[np.asarray(width)[rows].argsort()[::-1]]
            if 0 <= cut_idx < _I64_MAX:
                order = order[order != cut_idx]
        for a in range(0, order.size, chunk):
            part = order[a:a + chunk]
            flags = be.expanders_small(beta, self.fmin, part)
            is_exp = np.all(flags[:, active] != 0, axis=1)
            if full_sets:
                be.mark_expanders(part[is_exp])
            elif is_exp.any():
                be.mark_expanders(part[int(np.argmax(is_exp))][None])
                self._argmax_cache = None
                return

    def _gather_shards(self, part, N):
        """Concatenate every rank's block of a per-row array."""
        if self._comm.world == 1:
            return part
        counts = [np.subtract(*shard_range(N, r, self._comm.world)[::-1])
                  for r in range(self._comm.world)]
        buf = np.zeros((max(counts),) + part.shape[1:], dtype=part.dtype)
        buf[:part.shape[0]] = part
        allp = self._comm.allgather(buf)
        return np.concatenate([allp[r][:c] for r, c in enumerate(counts)])

    def _settle_ties(self, beta, active, w_star, idx_star, n_tied=None):
        """Exact ties in the visiting order of the expander loop.

        The device visits candidates by (width descending, index descending) and
        has just found its first expander ``idx_star`` of width ``w_star``.  The
        reference visits them in the order of ``widths.argsort()[::-1]``
        (gp_opt.py:542-552), which is the same except among candidates whose
        widths are EQUAL bit for bit -- NumPy's (unstable) sort decides there.
        Only when such a tie exists: take the candidate widths to the host, run
        the very same ``argsort()[::-1]`` on the very same values, and walk the
        tied group in that order -- members above ``idx_star`` were already
        rejected on the device, ``idx_star`` is an expander, the others are tested
        now; the first expander in THAT order is the one the reference marks.
        Returns the global index that ends up in ``G``.
        """
        be = self._backend
        if not hasattr(be, 'candidate_widths'):
            return idx_star
        if n_tied is None:
            # is the next candidate in the device's order tied with this one?
            w_loc, i_loc = be.topk(0, w_star, idx_star, 1)
            wp = np.full(1, -np.inf)
            wp[:w_loc.size] = w_loc[:1]
            n_tied = 2 if np.max(self._comm.allgather(wp)) == w_star else 1
        if n_tied <= 1:
            return idx_star
        # the candidate rows of the whole grid and their widths, in row order: what the
        # reference's ``width[rows]`` is.  N ranks exchange the CANDIDATES only (12 bytes
        # each), not the masks and widths of all rows (9 N bytes per shard)
        cand_loc, w_loc = be.candidate_widths()
        rows_loc = np.flatnonzero(np.asarray(cand_loc, dtype=bool))
        wc_loc = np.asarray(w_loc, dtype=float)[rows_loc]
        rows_loc = rows_loc + self._shard[0]
        if self._comm.world > 1:
            counts = self._comm.allgather(np.array([rows_loc.size], dtype=np.int64))[:, 0]
            pad = int(counts.max())
            buf = np.zeros((pad, 2))
            buf[:rows_loc.size, 0], buf[:rows_loc.size, 1] = rows_loc, wc_loc   # (idx < 2^53)
            allp = self._comm.allgather(buf)
            rows = np.concatenate([allp[r][:c, 0] for r, c in enumerate(counts)]).astype(np.int64)
            wrows = np.concatenate([allp[r][:c, 1] for r, c in enumerate(counts)])
        else:
            rows, wrows = rows_loc, wc_loc
        order = wrows.argsort()[::-1]              # the reference's own expression
        winner = idx_star
        for k in order:
            idx = int(rows[k])
            if wrows[k] != w_star or idx > idx_star:
                continue                           # other width / rejected before
            if idx == idx_star:
                break
            own = be.owns(idx)
            xc = np.zeros((1, self.inputs.shape[1]))
            mu_c = np.zeros((1, len(self.gps)))
            u_c = np.zeros((1, len(self.gps)))
            if own:
                x_o, mean_o, _v, Q_o = be.gather_rows(np.array([idx], dtype=np.int64))
                xc[0], mu_c[0], u_c[0] = x_o[0], mean_o[0], Q_o[0, 1::2]
            if self._comm.world > 1:
                packed = self._comm.allgather(
                    np.concatenate([xc, mu_c, u_c], axis=1)).sum(axis=0)
                d = self.inputs.shape[1]
                xc, mu_c, u_c = (packed[:, :d], packed[:, d:d + len(self.gps)],
                                 packed[:, d + len(self.gps):])
            if self._expander_flags(beta, xc, mu_c, u_c, active, probe=False)[0]:
                winner = idx
                break
        if winner != idx_star:
            if be.owns(idx_star):
                be.unmark_expanders(np.array([idx_star], dtype=np.int64))
            if be.owns(winner):
                be.mark_expanders(np.array([winner], dtype=np.int64))
            self._argmax_cache = None
        return winner

    def _expander_flags(self, beta, xc, mu_c, u_c, active, probe):
        """Which of the candidates are expanders (all ranks agree).

        ``probe``: first scan only the rows strongly correlated with the
        candidate; a hit there already certifies it (``any`` over a subset).
        Only when the FIRST candidate in visiting order is not certified that
        way is the exact scan over all unsafe rows run.
        """
        be = self._backend

        def run(**kw):
            if self.use_lipschitz:
                f = be.lipschitz_check(self.fmin, self.liptschitz, xc, u_c)
            else:
                f = be.expander_check(beta, self.fmin, xc, mu_c, u_c, **kw)
            f = self._comm.allreduce_max(f.astype(np.float64)) > 0
            return np.all(f[:, active], axis=1)

        if probe and not self.use_lipschitz:
            hit = run(near_frac=0.5)
            if hit[0]:
                return hit
        return run()

    def get_new_query_point(self, ucb=False):
        """Next parameters to evaluate (first index wins among equals)."""
        self._flush_Q()
        self._flush_masks()
        if not self._any_safe:
            raise EnvironmentError('There are no safe points to evaluate.')
        mode = _hip.ARGMAX_UCB if ucb else _hip.ARGMAX_MG_WIDTH
        cached = getattr(self, '_argmax_cache', None)
        if not ucb and cached is not None:
            idx = cached[1]          # arg-max already done with the sets
        else:
            idx = self._global_argmax(mode)[1]
        x = self.inputs[idx, :]
        if self.num_contexts:
            return x[:-self.num_contexts]
        return x

    def _global_argmax(self, mode):
        v, i = self._backend.argmax(mode, self.scaling)
        if self._comm.world > 1:
            v, i = merge_argmax(self._comm.allgather(np.array([v])),
                                self._comm.allgather(np.array([i],
                                                              dtype=np.int64)))
        return v, int(i)

    def optimize(self, context=None, ucb=False):
        """One SafeOpt step: intervals -> sets -> next query point."""
        # common case on one GPU: sweep, set passes, probe of the first
        # candidate and arg-max are enqueued back to back, one read-back
        # (N ranks: the same deferral with the two scalar all-reduces in stream)
        one_trip = (not ucb and not self.use_lipschitz
                    and hasattr(self._backend, 'sets_fused')
                    and bool(np.any(self.fmin != -np.inf))
                    and (self._comm.world == 1
                         or getattr(self._comm, 'in_stream', False)))
        if (one_trip and self._comm.world == 1 and self.small_step
                and hasattr(self._backend, 'step_small')):
            devs = self._backend.small_step_devs()
            if devs is not None:
                # a small grid (the reference's own regime): the whole step is one launch
                self._small_step(context, devs)
                return self.get_new_query_point()
        self.update_confidence_intervals(context=context, _defer=one_trip)
        if ucb:
            self.compute_safe_set()
        else:
            self.compute_sets()
        return self.get_new_query_point(ucb=ucb)

    def _small_step(self, context, devs):
        """``update_confidence_intervals`` + ``compute_sets`` of a small grid in one launch
        and one read-back (``sgp_grid_step_small``: grids of at most 16384 rows, GPs with at
        most 48 observations -- gp_opt.py:651-675 as the reference's examples run it)."""
        beta = self.beta(self.t)
        if self.num_contexts:
            self.context = context
        thr = self.threshold
        key = (beta, thr if isinstance(thr, (int, float)) else tuple(np.ravel(thr)))
        if self._thr_beta_key != key:        # (constant for a constant beta: built once)
            self._thr_beta = np.broadcast_to(
                np.asarray(self.threshold, dtype=float) * beta, (len(self.gps),)).copy()
            self._thr_beta_key = key
        # element-wise edits of opt.S / M / G since the last pass are dropped, not uploaded:
        # the launch recomputes all three sets (gp_opt.py:478-481, 505-615), exactly as
        # _flush_masks(for_sets=True) does in front of compute_sets on the large-grid path
        self._masks_written = set()
        (out5, x_c, mu_c, q_c, flags, val, idx,
         max_l) = self._backend.step_small(devs, beta, self.fmin, self.scaling, self._thr_beta)
        self._stale.update(Q=True, S=True)
        self._q_written = False          # (the sweep overwrites the intervals)
        self._s_user = False
        self._ci_fresh = True
        self._argmax_cache = None
        if self._deferred_max_l(max_l):
            self._certify_first_candidate(beta, self.fmin != -np.inf,
                                          self._front_of(out5, x_c, mu_c, q_c,
                                                         (flags, val, idx)), exact=True)

    def get_maximum(self, context=None):
        """Best lower bound inside the safe set: ``(x, l)`` or ``None``."""
        self.update_confidence_intervals(context=context)
        self.compute_safe_set()
        if not self._any_safe:
            return None
        v, idx = self._global_argmax(_hip.ARGMAX_LCB)
        return (self.inputs[idx, :-self.num_contexts or None], v)


def _step_into_band(value_at, lo=0.94, hi=0.95, v_max=1000., tol=1e-5):
    """Bisection for a DECREASING ``value_at``: the midpoint ``v`` of the bracket
    ``[a, b]`` (``value_at(a) >= hi > value_at(b)``) at which the value first lies
    strictly inside ``(lo, hi)`` or the bracket is shorter than ``tol``."""
    a, b = 0., v_max
    while True:
        v = 0.5 * (a + b)
        c = value_at(v)
        if c >= hi:
            a = v
        else:
            b = v
        if lo < c < hi or b - a < tol:
            return v


class SafeOptSwarm(GaussianProcessOptimization):
    """SafeOpt for higher dimensions with adaptive swarm discretisation.

    Same constructor and behaviour as ``gp_opt.py:715-1192`` of the reference
    (no Lipschitz constant, no contexts).  The particle fitness -- the GP
    posterior of every GP at all particles plus the penalty / interest shaping
    -- is one fused HIP kernel per swarm iteration; the swarm bookkeeping and
    its NumPy global RNG stay on the host so runs are reproducible against the
    reference.
    """

    def __init__(self, gp, fmin, bounds, beta=2, scaling='auto', threshold=0,
                 swarm_size=20, pso='device'):
        if pso not in ('device', 'device-rng', 'host'):
            raise ValueError("pso must be 'device', 'device-rng' or 'host'")
        GaussianProcessOptimization.__init__(self, gp, fmin=fmin, beta=beta, num_contexts=0,
                                             threshold=threshold, scaling=scaling)
        # the safe set starts as the observed inputs; one (min, max) pair may
        # stand for every dimension
        safe_points = np.asarray(self.gps[0].X)
        self.S, self.greedy_point = safe_points, safe_points[0, :]
        self.swarm_size, self.max_iters = swarm_size, 100
        self.bounds = bounds if isinstance(bounds, list) else [bounds] * safe_points.shape[1]
        self.best_lower_bound = -np.inf
        self.optimal_velocities = self.optimize_particle_velocity()

        # pso='device' (default): whole swarm runs on the GPU with NumPy's
        # random stream (bit-identical to the host loop); 'device-rng': GPU
        # generator, no per-run random upload; 'host': the reference's loop
        # with one fitness call per iteration.
        if pso == 'host':
            self.swarms = {
                swarm_type: SwarmOptimization(
                    swarm_size, self.optimal_velocities,
                    partial(self._compute_particle_fitness, swarm_type),
                    bounds=self.bounds)
                for swarm_type in ['greedy', 'maximizers', 'expanders']}
        else:
            self.swarms = {
                swarm_type: DeviceSwarmOptimization(
                    swarm_size, self.optimal_velocities, self, swarm_type,
                    bounds=self.bounds,
                    rng='numpy' if pso == 'device' else 'device')
                for swarm_type in ['greedy', 'maximizers', 'expanders']}

    def optimize_particle_velocity(self):
        """Step length per input dimension at which a particle's prior correlation
        with its starting point has fallen into (0.94, 0.95), the smallest over the
        GPs, divided by sqrt(d) (``gp_opt.py:818-872``).  Stationary kernels only:
        the correlation is probed from the origin."""
        d = self.gp.input_dim
        origin = np.zeros((1, d))

        def correlation(gp, scale, axis):
            probe = np.zeros((1, d))

            def at(v):
                probe[0, axis] = v
                return float(np.squeeze(gp.kern.K(origin, probe))) / scale ** 2
            return at

        per_gp = [[_step_into_band(correlation(gp, self.scaling[i], j)) for j in range(d)]
                  for i, gp in enumerate(self.gps)]
        return np.min(np.asarray(per_gp, dtype=float), axis=0) / np.sqrt(d)

    def _compute_penalty(self, slack):
        """Penalty of a constraint violation (``gp_opt.py:874-899``): zero for a
        satisfied constraint, the (negative) slack times 2 / 5 / 10 on the bands
        (-0.001, 0), (-1, -0.1], ..., and ``-300 slack^2`` below -1 (a slack of
        exactly -1 keeps its own value).  Host helper; the device applies the same
        rule inside the fitness kernels (``csrc/fitness.h``)."""
        s = np.atleast_1d(np.asarray(slack, dtype=float))
        return np.select([s >= 0, s > -0.001, s > -0.1, s > -1, s == -1],
                         [np.zeros_like(s), 2 * s, 5 * s, 10 * s, s], default=-300 * (s * s))

    def _compute_particle_fitness(self, swarm_type, particles):
        """Fitness and safety of ``particles`` for one swarm type:
        ``'greedy' | 'maximizers' | 'expanders' | 'safe_set'``."""
        if swarm_type not in _hip.SWARM_TYPES:
            raise AssertionError("Invalid swarm type")
        beta = self.beta(self.t)
        particles = np.atleast_2d(particles)
        devs = [g._fitted() for g in self.gps]
        values, safe = _hip.swarm_fitness(
            devs[0].ctx, devs, swarm_type, particles, beta, self.fmin,
            self.scaling, self.best_lower_bound)
        return values, safe

    # -- one swarm run, in the steps of gp_opt.py:1015-1134 -------------------------
    def _recheck_safe_set(self):
        """The stored safe points under the CURRENT posterior: raise if none is
        safe any more, drop the unsafe ones unless that would leave fewer than
        one swarm's worth of points."""
        _, still_safe = self._compute_particle_fitness('safe_set', self.S)
        kept = int(still_safe.sum())
        if kept == 0:
            raise RuntimeError('The safe set is empty.')
        if self.swarm_size <= kept < len(still_safe):
            logging.warning("Warning: {} unsafe points removed. "
                            "Model might be violated"
                            .format(np.count_nonzero(~still_safe)))
            self.S = self.S[still_safe]

    def _initial_particles(self, swarm_type):
        """Start positions: uniform draws (with replacement, NumPy's global
        stream -- one ``randint`` call, as in the reference) from the safe set;
        the greedy swarm swaps three of them for the previous greedy point, the
        newest observation and the best observation."""
        fixed = []
        if swarm_type == 'greedy':
            fixed = [self.greedy_point, self.gp.X[-1, :],
                     self.gp.X[np.argmax(self.gp.Y)]]
        picks = np.random.randint(self.S.shape[0],
                                  size=self.swarm_size - len(fixed))
        return np.vstack([self.S[picks, :]] + fixed)

    def _grow_safe_set(self, swarm, swarm_type):
        """Append the swarm's personal bests that are at most 0.95-correlated
        (prior correlation of GP 0) with everything already in the safe set,
        in order.  The correlation filter runs on the device (``sgp_swarm_grow``);
        the m x (|S| + m) covariance matrix of gp_opt.py:1093 is never formed."""
        gp0 = self.gp._fitted()
        keep = _hip.swarm_grow(gp0.ctx, gp0, self.S, swarm.best_positions,
                               self.scaling[0] ** 2, 0.95)
        added = int(np.count_nonzero(keep))
        if added:
            self.S = np.vstack((self.S, swarm.best_positions[keep]))
        logging.debug("At the end of swarm {}, {} points were appended to"
                      " the safeset".format(swarm_type, added))

    def _lower_bound_at(self, x, beta):
        mean, var = self.gp.predict_noiseless(x[None, :])
        return mean.squeeze() - beta * np.sqrt(var.squeeze())

    def get_new_query_point(self, swarm_type):
        """Run one swarm (``'greedy' | 'maximizers' | 'expanders'``).

        Returns ``(global_best, value)`` for the greedy swarm and
        ``(global_best, std_devs)`` -- one posterior standard deviation per GP
        at the chosen point -- otherwise.
        """
        beta = self.beta(self.t)
        self._recheck_safe_set()

        swarm = self.swarms[swarm_type]
        swarm.init_swarm(self._initial_particles(swarm_type))
        swarm.run_swarm(self.max_iters)

        if swarm_type == 'greedy':
            # keep the better of the old and the new estimate of the optimum
            best_value = np.max(swarm.best_values)
            if self._lower_bound_at(self.greedy_point, beta) < best_value:
                self.greedy_point = swarm.global_best.copy()
            return swarm.global_best.copy(), best_value

        self._grow_safe_set(swarm, swarm_type)
        point = swarm.global_best
        std = np.sqrt([gp.predict_noiseless(point[None, :])[1].item()
                       for gp in self.gps])
        return point, std

    def optimize(self, ucb=False):
        """One SafeOptSwarm step; returns the next parameters to evaluate:
        the best maximiser, or the best expander when that one is at least as
        uncertain (``ucb=True``: always the maximiser)."""
        self.greedy, self.best_lower_bound = self.get_new_query_point('greedy')

        x_maxi, std_maxi = self.get_new_query_point('maximizers')
        if ucb:
            logging.info('Using ucb criterion.')
            return x_maxi
        x_exp, std_exp = self.get_new_query_point('expanders')

        # expander uncertainty: only GPs that carry a constraint and are above
        # the threshold count, each in units of its prior std
        counts = (std_exp >= self.threshold) & (self.fmin != -np.inf)
        std_exp = np.max(np.where(counts, std_exp, 0.) / self.scaling)
        std_maxi = std_maxi[0] / self.scaling[0]

        logging.info("The best maximizer has std. dev. %f" % std_maxi)
        logging.info("The best expander has std. dev. %f" % std_exp)
        logging.info("The greedy estimate of lower bound has value %f" %
                     self.best_lower_bound)
        return x_maxi if std_maxi > std_exp else x_exp

    def get_maximum(self):
        """Best observed point ``(x, y)``."""
        maxi = np.argmax(self.gp.Y)
        return self.gp.X[maxi, :], self.gp.Y[maxi]
