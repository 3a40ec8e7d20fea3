"""ctypes binding of libsafeopt_hip.so (include/safeopt_hip.h).

There is NO CPU fallback: every numerical entry point of the package goes
through this module, and a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SAFEOPT_HIP_LIB: another build of the same library, for same-box A/B runs)
LIB_PATH = os.environ.get("SAFEOPT_HIP_LIB") or os.path.join(_HERE, "libsafeopt_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_i32_p = C.POINTER(C.c_int32)
c_u32_p = C.POINTER(C.c_uint32)
c_i64_p = C.POINTER(C.c_int64)
c_u8_p = C.POINTER(C.c_uint8)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)
# host-side collectives a caller may plug in (sgp_comm_init_host)
HOST_ALLREDUCE_F64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
HOST_ALLREDUCE_I32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int)
HOST_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

RBF, MATERN32, MATERN52 = 0, 1, 2
Q, S, M, G, MEAN, VAR, CAND, WIDTH = 0, 1, 2, 3, 4, 5, 6, 7
ARGMAX_MG_WIDTH, ARGMAX_UCB, ARGMAX_LCB = 0, 1, 2
SWARM_TYPES = {"greedy": 0, "maximizers": 1, "expanders": 2, "safe_set": 3}
MAX_D, MAX_PARTS, MAX_GPS, TOPK = 8, 4, 8, 16

# name -> (restype, argtypes); mirrors include/safeopt_hip.h one to one
PROTOTYPES = {
    "sgp_device_count": (C.c_int, [c_int_p]),
    "sgp_create": (C.c_int, [C.c_int, vpp]),
    "sgp_destroy": (None, [vp]),
    "sgp_last_error": (C.c_char_p, [vp]),
    "sgp_sync": (C.c_int, [vp]),
    "sgp_gp_create": (C.c_int, [vp, C.c_int, C.c_int, c_int_p, c_double_p,
                                c_double_p, C.c_double, vpp]),
    "sgp_gp_destroy": (None, [vp]),
    "sgp_gp_set_data": (C.c_int, [vp, c_double_p, c_double_p, C.c_int64,
                                  c_int_p, c_double_p]),
    "sgp_gp_append": (C.c_int, [vp, c_double_p, C.c_double, c_int_p]),
    "sgp_gp_pop": (C.c_int, [vp]),
    "sgp_gp_predict": (C.c_int, [vp, c_double_p, C.c_int64, C.c_int64,
                                 C.c_int64, c_double_p, c_double_p]),
    "sgp_gp_get_factor": (C.c_int, [vp, c_double_p, c_double_p]),
    "sgp_kern_K": (C.c_int, [vp, C.c_int, C.c_int, c_int_p, c_double_p,
                             c_double_p, c_double_p, C.c_int64, c_double_p,
                             C.c_int64, c_double_p]),
    "sgp_grid_create": (C.c_int, [vp, c_double_p, C.c_int64, C.c_int,
                                  C.c_int64, C.c_int64, C.c_int, C.c_int64,
                                  vpp]),
    "sgp_grid_destroy": (None, [vp]),
    "sgp_grid_set_context": (C.c_int, [vp, c_double_p, C.c_int]),
    "sgp_grid_set_axes": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    c_double_p, c_int_p]),
    "sgp_grid_confidence": (C.c_int, [vp, vpp, C.c_int, C.c_double,
                                      c_double_p, c_double_p]),
    "sgp_grid_posterior": (C.c_int, [vp, vpp, C.c_int]),
    "sgp_grid_rank1_update": (C.c_int, [vp, vpp, C.c_int, c_int_p, C.c_double,
                                        c_double_p, c_double_p]),
    "sgp_grid_upload_Q": (C.c_int, [vp, c_double_p, c_double_p, c_double_p]),
    "sgp_grid_maximizers": (C.c_int, [vp, C.c_double, c_double_p]),
    "sgp_grid_candidates": (C.c_int, [vp, C.c_double, c_double_p, c_double_p,
                                      C.c_int, c_i64_p]),
    "sgp_grid_topk": (C.c_int, [vp, C.c_int, C.c_double, C.c_int64, C.c_int,
                                c_double_p, c_i64_p, c_int_p]),
    "sgp_grid_gather_rows": (C.c_int, [vp, c_i64_p, C.c_int, c_double_p,
                                       c_double_p, c_double_p, c_double_p]),
    "sgp_grid_expander_check": (C.c_int, [vp, vpp, C.c_int, C.c_double,
                                          c_double_p, C.c_int, c_double_p,
                                          c_double_p, c_double_p, C.c_double,
                                          c_i32_p]),
    "sgp_grid_lipschitz_check": (C.c_int, [vp, C.c_int, c_double_p,
                                           c_double_p, C.c_int, c_double_p,
                                           c_double_p, c_i32_p]),
    "sgp_grid_sets_front": (C.c_int, [vp, C.c_double, C.c_int, C.c_double,
                                      c_double_p, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p]),
    "sgp_grid_sets_front_comm": (C.c_int, [vp, c_double_p, c_double_p,
                                           c_double_p, c_double_p, c_double_p,
                                           c_double_p, c_double_p]),
    "sgp_grid_sets_fused": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p,
                                      C.c_double, c_double_p, c_double_p,
                                      C.c_double, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_i32_p,
                                      c_double_p, c_i64_p, c_double_p]),
    "sgp_grid_expander_batch": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p, C.c_int,
                                          C.c_double, C.c_int64, C.c_int, c_double_p, c_i64_p,
                                          c_int_p, c_i32_p]),
    "sgp_grid_expander_pass": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p, C.c_int,
                                         C.c_double, C.c_int64, C.c_double, C.c_double, C.c_int,
                                         c_double_p, c_double_p]),
    "sgp_grid_lipschitz_pass": (C.c_int, [vp, C.c_int, c_double_p, c_double_p, C.c_int, C.c_double,
                                          C.c_int64, C.c_double, C.c_double, C.c_int, c_double_p,
                                          c_double_p]),
    "sgp_grid_pass_lipschitz_test": (C.c_int, [vp, C.c_int, c_double_p, c_double_p, C.c_int,
                                               c_double_p, c_double_p, c_i32_p]),
    "sgp_grid_expanders_small_all": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p, C.c_int,
                                               C.POINTER(C.c_int), c_i64_p, c_double_p, c_i32_p]),
    "sgp_grid_pass_hist": (C.c_int, [vp, C.c_int, C.c_double, C.c_int64, C.c_double, C.c_double,
                                     c_u32_p]),
    "sgp_grid_pass_list": (C.c_int, [vp, C.c_int, C.c_double, C.c_int64, C.c_double, C.c_int,
                                     c_int_p, c_i64_p, c_double_p, c_double_p, c_double_p]),
    "sgp_grid_pass_test": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p, C.c_int,
                                     c_double_p, c_double_p, c_i32_p]),
    "sgp_grid_step_small": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p, c_double_p,
                                      c_double_p, c_i32_p, c_double_p, c_i64_p, c_double_p]),
    "sgp_grid_step_small_ok": (C.c_int, [vp, vpp, C.c_int]),
    "sgp_grid_expanders_small": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p, c_i64_p,
                                           C.c_int, c_i32_p]),
    "sgp_grid_sets_fused_comm": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p,
                                           c_double_p, c_double_p, C.c_double,
                                           c_double_p, c_double_p, c_double_p,
                                           c_double_p, c_i32_p, c_double_p, c_i64_p,
                                           c_double_p]),
    "sgp_grid_sets_back": (C.c_int, [vp, vpp, C.c_int, C.c_double, c_double_p,
                                     c_double_p, c_double_p, c_double_p,
                                     C.c_double, C.c_int64, C.c_int,
                                     c_double_p, c_i32_p, c_double_p,
                                     c_i64_p]),
    "sgp_grid_mark_expanders": (C.c_int, [vp, c_i64_p, C.c_int]),
    "sgp_grid_unmark_expanders": (C.c_int, [vp, c_i64_p, C.c_int]),
    "sgp_grid_argmax": (C.c_int, [vp, C.c_int, c_double_p, c_double_p,
                                  c_i64_p]),
    "sgp_grid_download": (C.c_int, [vp, C.c_int, vp]),
    "sgp_grid_upload_mask": (C.c_int, [vp, C.c_int, c_u8_p]),
    "sgp_swarm_fitness": (C.c_int, [vp, vpp, C.c_int, C.c_int, c_double_p,
                                    C.c_int64, C.c_double, c_double_p,
                                    c_double_p, C.c_double, c_double_p,
                                    c_u8_p]),
    "sgp_swarm_grow": (C.c_int, [vp, vp, c_double_p, C.c_int64, c_double_p,
                                 C.c_int64, C.c_double, C.c_double, c_u8_p]),
    "sgp_swarm_run": (C.c_int, [vp, vpp, C.c_int, C.c_int, C.c_double,
                                c_double_p, c_double_p, C.c_double, C.c_int64,
                                c_double_p, c_double_p, c_double_p, c_double_p,
                                c_double_p, c_double_p, c_double_p, C.c_int,
                                C.c_int, C.c_double, C.c_double, c_double_p,
                                C.c_uint64]),
    "sgp_comm_unique_id": (C.c_int, [vp]),
    "sgp_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "sgp_comm_init_host": (C.c_int, [vp, C.c_int, C.c_int, HOST_ALLREDUCE_F64,
                                     HOST_ALLREDUCE_I32, HOST_ALLGATHER, vp]),
    "sgp_comm_allreduce_max": (C.c_int, [vp, c_double_p, C.c_int]),
    "sgp_comm_allgather": (C.c_int, [vp, vp, vp, C.c_int64]),
    "sgp_comm_barrier": (C.c_int, [vp]),
    "sgp_timer_start": (C.c_int, [vp]),
    "sgp_timer_stop": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "sgp_profile_enable": (C.c_int, [vp, C.c_int]),
    "sgp_profile_read": (C.c_int, [vp, c_double_p, c_i64_p, c_double_p]),
    "sgp_ctx_alloc_count": (C.c_int64, [vp]),
    "sgp_comm_count": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "sgp_ctx_set_sweep": (C.c_int, [vp, C.c_int]),
    "sgp_ctx_last_sweep": (C.c_int, [vp]),
    "sgp_ctx_set_share": (C.c_int, [vp, C.c_int]),

}

_lib = None


class _stdout_to_stderr(object):
    """RCCL prints a version banner on stdout when a communicator is created;
    keep stdout clean (bench.py promises exactly one JSON line there)."""

    def __enter__(self):
        import sys
        try:
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        except OSError:
            self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            try:                      # RCCL uses buffered C stdio: flush it
                C.CDLL(None).fflush(None)   # while fd 1 still is stderr
            except Exception:
                pass
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


class HipError(RuntimeError):
    """A call into libsafeopt_hip.so failed (HIP / RCCL / argument error)."""


def lib():
    """Load the shared library once; fail loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(
                "libsafeopt_hip.so is not built (%s). Run "
                "`python -m safeopt_amd.build`; there is no CPU fallback."
                % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_count():
    n = C.c_int(0)
    lib().sgp_device_count(C.byref(n))
    return n.value


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context(object):
    """One device context (stream, scratch, optional RCCL communicator)."""

    _default = {}

    def __init__(self, device=0):
        L = lib()
        if device_count() <= device:
            raise HipError("no HIP device %d visible: %s -- the HIP path is "
                           "the only path" % (device, L.sgp_last_error(None).decode()))
        h = vp()
        rc = L.sgp_create(device, C.byref(h))
        if rc != 0:
            raise HipError(L.sgp_last_error(None).decode())
        self.h = h
        self.device = device
        self.rank, self.world = 0, 1
        self.sweep_forced = any(os.environ.get(k) for k in
                                ("SGP_SWEEP", "SGP_NO_TINY", "SGP_NO_STEP_SMALL"))

    @classmethod
    def default(cls, device=None):
        if device is None:
            device = int(os.environ.get("SAFEOPT_HIP_DEVICE",
                                        os.environ.get("LOCAL_RANK", "0")))
            n = device_count()
            world = int(os.environ.get("LOCAL_WORLD_SIZE",
                                       os.environ.get("WORLD_SIZE", "1")))
            if n > 0 and device >= n:
                shared = os.environ.get("SAFEOPT_COMM", "rccl") == "socket"
                if world > 1 and "SAFEOPT_HIP_DEVICE" not in os.environ and not shared:
                    # two ranks on one GPU: RCCL would fail or hang in
                    # ncclCommInitRank -- say what is wrong instead
                    raise HipError(
                        "LOCAL_RANK %d but only %d visible GPU(s): one process per "
                        "GPU (launch with --nproc-per-node <= %d, or set "
                        "SAFEOPT_HIP_DEVICE explicitly)" % (device, n, n))
                device %= n
        if device not in cls._default:
            cls._default[device] = Context(device)
        return cls._default[device]

    def check(self, rc):
        if rc != 0:
            raise HipError("libsafeopt_hip (rc=%d): %s"
                           % (rc, lib().sgp_last_error(self.h).decode()))

    def sync(self):
        self.check(lib().sgp_sync(self.h))

    # -- measurement
    def timer_start(self):
        self.check(lib().sgp_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float(0)
        self.check(lib().sgp_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self.check(lib().sgp_profile_enable(self.h, int(bool(on))))

    def profile_read(self):
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        self.check(lib().sgp_profile_read(self.h, C.byref(ms), C.byref(n),
                                          C.byref(fl)))
        return ms.value, n.value, fl.value

    def alloc_count(self):
        """Device allocations made so far (a warm loop must not add any)."""
        return int(lib().sgp_ctx_alloc_count(self.h))

    def comm_count(self):
        """Ranks of the RCCL communicator (ncclCommCount); 1 without one."""
        n = C.c_int(0)
        self.check(lib().sgp_comm_count(self.h, C.byref(n)))
        return n.value

    def set_share(self, on):
        """GPs with identical inputs / kernel / noise share the variance contraction
        of the sweep (default on); returns the previous setting."""
        return bool(lib().sgp_ctx_set_share(self.h, int(bool(on))))

    def set_sweep(self, which):
        """Posterior-sweep kernel: 'auto' | 'classic' (4 waves) | 'pair' (paired
        waves) | 'mid' (= auto: the resident-factor kernel where it applies, which
        'classic' and 'pair' switch off), '-nosplit' appended: remainder tiles are not cut into runs of
        chunks, '-notables': no factor tables on tensor grids (set_axes), '-streamed':
        small factors go through the 4-wave kernel's double buffer instead of staying
        in LDS for the launch, '-unmerged': the paired kernel runs one j-block per stage
        (the schedule until round 5; merged stages: csrc/sweep_pair.hip); or the integer
        of sgp_ctx_set_sweep.  Returns the
        previous setting (a name)."""
        names = ("auto", "classic", "pair", "mid", "auto-nosplit", "classic-nosplit",
                 "pair-nosplit", None)
        names = names + tuple(n + "-notables" if n else None for n in names)
        names = names + tuple(n + "-streamed" if n else None for n in names)
        names = names + tuple(n + "-unmerged" if n else None for n in names)

        def code(w):          # a name, or the integer of sgp_ctx_set_sweep
            return int(w) if isinstance(w, (int, np.integer)) else names.index(w)

        old = names[int(lib().sgp_ctx_set_sweep(self.h, code(which)))]
        #: a sweep kernel is forced (A/B runs, tests): no one-launch step of small grids
        # (3 = 'mid' counts too: step_small_eligible in csrc/step_small.hip refuses every
        # non-zero choice, the driver must not pick the one-launch step then)
        self.sweep_forced = (code(which) & 3) != 0
        return old

    def last_sweep(self):
        """Kernel of the last posterior sweep: 'classic' | 'pair' | 'tiny' | 'few-points' |
        'step-small' | 'mid'."""
        return (None, "classic", "pair", "tiny", "few-points",
                "step-small", "mid")[int(lib().sgp_ctx_last_sweep(self.h))]

    # -- RCCL
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        with _stdout_to_stderr():
            rc = lib().sgp_comm_unique_id(buf)
        if rc != 0:
            raise HipError(lib().sgp_last_error(None).decode())
        return buf.raw

    def comm_init(self, uid, rank, world):
        buf = C.create_string_buffer(uid, 128)
        with _stdout_to_stderr():
            rc = lib().sgp_comm_init(self.h, buf, rank, world)
        self.check(rc)
        self.rank, self.world = rank, world

    def comm_init_host(self, comm):
        """The caller's collectives as this context's transport (``sgp_comm_init_host``):
        ``comm`` has ``rank``, ``world``, ``allreduce_max(array)`` and ``allgather(array)``
        on host arrays (``dist.SocketComm``).  The N-rank entry points then stage their
        device operands through the host around these calls."""
        def _guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except Exception:           # noqa -- nothing may unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            return run

        def ar_f64(_user, buf, n):
            a = np.ctypeslib.as_array(buf, shape=(n,))
            a[:] = comm.allreduce_max(a.copy())

        def ar_i32(_user, buf, n):
            a = np.ctypeslib.as_array(buf, shape=(n,))
            # (flags / small counts: exact in float64)
            a[:] = comm.allreduce_max(a.astype(np.float64)).astype(np.int32)

        def ag(_user, send, recv, nbytes):
            raw = C.string_at(send, nbytes)
            parts = comm.allgather(np.frombuffer(raw, dtype=np.uint8))
            C.memmove(recv, np.ascontiguousarray(parts).ctypes.data, nbytes * comm.world)

        # (the ctypes thunks must outlive the context's use of them)
        self._host_cbs = (HOST_ALLREDUCE_F64(_guard(ar_f64)), HOST_ALLREDUCE_I32(_guard(ar_i32)),
                          HOST_ALLGATHER(_guard(ag)))
        self.check(lib().sgp_comm_init_host(self.h, int(comm.rank), int(comm.world),
                                            *self._host_cbs, None))
        self.rank, self.world = int(comm.rank), int(comm.world)

    def allreduce_max(self, a):
        a = f64(a).copy()
        self.check(lib().sgp_comm_allreduce_max(self.h, dptr(a), a.size))
        return a

    def allgather_bytes(self, b):
        out = C.create_string_buffer(len(b) * self.world)
        src = C.create_string_buffer(bytes(b), len(b))
        self.check(lib().sgp_comm_allgather(self.h, src, out, len(b)))
        return out.raw

    def barrier(self):
        self.check(lib().sgp_comm_barrier(self.h))

    def kern_K(self, kdesc, X1, X2):
        d, kinds, variances, inv_ls = kdesc
        X1 = f64(X1).reshape(-1, d)
        X2 = f64(X2).reshape(-1, d)
        out = np.empty((X1.shape[0], X2.shape[0]))
        if out.size:
            self.check(lib().sgp_kern_K(
                self.h, d, len(kinds), kinds.ctypes.data_as(c_int_p),
                dptr(variances), dptr(inv_ls), dptr(X1), X1.shape[0],
                dptr(X2), X2.shape[0], dptr(out)))
        return out


def _gp_array(gps):
    arr = (vp * len(gps))(*[g.h for g in gps])
    return C.cast(arr, vpp)


class DeviceGP(object):
    """Device-resident GP state (training data, packed L^-1, alpha)."""

    _serials = itertools.count(1)      # process-wide, never reused

    def __init__(self, ctx, kdesc, noise_var):
        d, kinds, variances, inv_ls = kdesc
        self.ctx = ctx
        self.d = d
        h = vp()
        ctx.check(lib().sgp_gp_create(
            ctx.h, d, len(kinds), kinds.ctypes.data_as(c_int_p),
            dptr(variances), dptr(inv_ls), float(noise_var), C.byref(h)))
        self.h = h
        self.n = 0
        self.jitter = 0.0
        # data version: bumped by every change; `appended` = the last change
        # was a one-row append whose rank-1 record is on the device
        self.version = 0
        self.appended = False
        self.serial = next(DeviceGP._serials)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().sgp_gp_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_data(self, X, Y):
        X = f64(X).reshape(-1, self.d)
        Y = f64(Y).reshape(-1)
        info, jit = C.c_int(0), C.c_double(0)
        rc = lib().sgp_gp_set_data(self.h, dptr(X), dptr(Y), X.shape[0],
                                   C.byref(info), C.byref(jit))
        if rc > 0 or info.value != 0:
            raise np.linalg.LinAlgError(
                lib().sgp_last_error(self.ctx.h).decode())
        self.ctx.check(rc)
        self.n = X.shape[0]
        self.jitter = jit.value
        self.version += 1
        self.appended = False

    def append(self, x, y):
        """One more observation by a bordered update; False = not possible
        (no capacity / pivot not positive): the caller refits with set_data."""
        x = f64(x).reshape(self.d)
        info = C.c_int(0)
        self.ctx.check(lib().sgp_gp_append(self.h, dptr(x), float(y),
                                           C.byref(info)))
        if info.value != 0:
            return False
        self.n += 1
        self.version += 1
        self.appended = True
        return True

    def pop(self):
        """Drop the last observation (O(n^2))."""
        self.ctx.check(lib().sgp_gp_pop(self.h))
        self.n -= 1
        self.version += 1
        self.appended = False

    def predict(self, Xnew):
        Xnew = np.asarray(Xnew, dtype=np.float64)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.d:
            Xnew = np.atleast_2d(Xnew).reshape(-1, self.d)
        N = Xnew.shape[0]
        it = Xnew.itemsize
        if Xnew.strides[0] % it or Xnew.strides[1] % it or \
                min(Xnew.strides) < 0:
            Xnew = np.ascontiguousarray(Xnew)
        mean = np.empty((N, 1))
        var = np.empty((N, 1))
        if N:
            self.ctx.check(lib().sgp_gp_predict(
                self.h, dptr(Xnew), N, Xnew.strides[0] // it,
                Xnew.strides[1] // it, dptr(mean), dptr(var)))
        return mean, var

    def factor(self):
        Linv = np.empty((self.n, self.n))
        alpha = np.empty(self.n)
        self.ctx.check(lib().sgp_gp_get_factor(self.h, dptr(Linv), dptr(alpha)))
        return Linv, alpha


def tensor_grid_axes(points):
    """``(counts, strides, axis values)`` of an (N, d) array whose rows are a tensor
    grid -- row ``i`` has column ``k`` equal to ``values[k][(i // strides[k]) % counts[k]]``,
    as ``linearly_spaced_combinations`` builds it (safeopt/utilities.py:21-54), constant
    columns (contexts) having one point -- or None.  Necessary conditions only (run
    lengths, periods, the chain of the strides): the device compares every row
    (``DeviceGrid.set_axes``)."""
    points = np.asarray(points)
    if points.ndim != 2 or points.shape[0] < 1:
        return None
    N, d = points.shape
    counts, strides, values = [], [], []
    for k in range(d):
        col = points[:, k]
        change = np.flatnonzero(col[1:] != col[:-1])
        if change.size == 0:
            counts.append(1); strides.append(1); values.append(col[:1].copy())
            continue
        run = int(change[0]) + 1
        firsts = col[::run]
        again = np.flatnonzero(firsts[1:] == firsts[0])
        cnt = int(again[0]) + 1 if again.size else int(firsts.size)
        counts.append(cnt); strides.append(run); values.append(firsts[:cnt].copy())
    total, expect = 1, 1
    for k in sorted((k for k in range(d) if counts[k] > 1), key=lambda k: strides[k]):
        if strides[k] != expect:
            return None
        expect *= counts[k]
        total *= counts[k]
    if total != N or N >= 2 ** 31:
        return None
    return counts, strides, values


class DeviceGrid(object):
    """This rank's shard of SafeOpt.inputs resident in HBM, with Q/S/M/G."""

    def __init__(self, ctx, inputs, G, global_offset=0):
        inputs = np.asarray(inputs, dtype=np.float64)
        assert inputs.ndim == 2
        if min(inputs.strides) < 0:
            inputs = np.ascontiguousarray(inputs)
        self.ctx = ctx
        self.N, self.d = inputs.shape
        self.G = G
        self.goff = int(global_offset)
        h = vp()
        ctx.check(lib().sgp_grid_create(
            ctx.h, dptr(inputs), self.N, self.d, inputs.strides[0],
            inputs.strides[1], G, self.goff, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().sgp_grid_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_context(self, c):
        c = f64(c).reshape(-1)
        self.ctx.check(lib().sgp_grid_set_context(self.h, dptr(c), c.size))

    def set_axes(self, axes):
        """Declare the rows a tensor grid (``axes`` from ``tensor_grid_axes`` of the
        WHOLE grid, this shard's rows being its rows ``goff .. goff + N``); the device
        checks the declaration against the resident rows.  True when it holds: RBF
        kernels are then swept through per-axis factor tables."""
        if axes is None:
            return False
        counts, strides, values = axes
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        stv = np.ascontiguousarray(strides, dtype=np.int64)
        vals = f64(np.concatenate([np.asarray(v, dtype=np.float64) for v in values]))
        ok = C.c_int(0)
        self.ctx.check(lib().sgp_grid_set_axes(
            self.h, int(cnt.size), cnt.ctypes.data_as(C.POINTER(C.c_int64)),
            stv.ctypes.data_as(C.POINTER(C.c_int64)), dptr(vals), C.byref(ok)))
        return bool(ok.value)

    def upload_mask(self, what, mask):
        """``S`` / ``M`` / ``G`` of this shard from the host (user code wrote into the mirror)."""
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape == (self.N,)
        self.ctx.check(lib().sgp_grid_upload_mask(self.h, int(what), m.ctypes.data_as(c_u8_p)))

    def clear_axes(self):
        """Forget the tensor-grid declaration (the sweeps evaluate covariances again)."""
        one = np.ones(self.d, dtype=np.int64)
        ok = C.c_int(0)
        self.ctx.check(lib().sgp_grid_set_axes(
            self.h, self.d, one.ctypes.data_as(C.POINTER(C.c_int64)),
            one.ctypes.data_as(C.POINTER(C.c_int64)), dptr(np.zeros(self.d)), C.byref(ok)))
        assert not ok.value

    def confidence(self, gps, beta, fmin, defer=False):
        """``defer``: enqueue only; ``max l0[S]`` stays on the device and comes
        back with the next ``sets_fused`` call (returns ``(None, None)``)."""
        fmin = f64(fmin)
        out = np.empty(2)
        self.ctx.check(lib().sgp_grid_confidence(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin),
            None if defer else dptr(out)))
        return (None, None) if defer else (out[0], bool(out[1]))

    def posterior(self, gps):
        """Resident mean / var of every GP from a full sweep; Q, S untouched."""
        self.ctx.check(lib().sgp_grid_posterior(self.h, _gp_array(gps), len(gps)))

    def rank1_update(self, gps, which, beta, fmin, defer=False):
        fmin = f64(fmin)
        w = np.ascontiguousarray(which, dtype=np.int32)
        out = np.empty(2)
        self.ctx.check(lib().sgp_grid_rank1_update(
            self.h, _gp_array(gps), len(gps), w.ctypes.data_as(c_int_p),
            float(beta), dptr(fmin), None if defer else dptr(out)))
        return (None, None) if defer else (out[0], bool(out[1]))

    def upload_Q(self, Qh, fmin):
        Qh = f64(Qh).reshape(self.N, 2 * self.G)
        fmin = f64(fmin)
        out = np.empty(2)
        self.ctx.check(lib().sgp_grid_upload_Q(self.h, dptr(Qh), dptr(fmin),
                                               dptr(out)))
        return out[0], bool(out[1])

    def maximizers(self, max_l):
        out = C.c_double(0)
        self.ctx.check(lib().sgp_grid_maximizers(self.h, float(max_l),
                                                 C.byref(out)))
        return out.value

    def candidates(self, max_var, scaling, thr_beta, full_sets=False):
        scaling = f64(scaling)
        thr_beta = f64(thr_beta)
        counts = np.zeros(2, dtype=np.int64)
        self.ctx.check(lib().sgp_grid_candidates(
            self.h, float(max_var), dptr(scaling), dptr(thr_beta),
            int(bool(full_sets)), counts.ctypes.data_as(c_i64_p)))
        return int(counts[0]), int(counts[1])

    def topk(self, mode, cut_w, cut_idx, k=TOPK):
        w = np.empty(k)
        idx = np.empty(k, dtype=np.int64)
        n = C.c_int(0)
        self.ctx.check(lib().sgp_grid_topk(
            self.h, int(mode), float(cut_w), int(cut_idx), k, dptr(w),
            idx.ctypes.data_as(c_i64_p), C.byref(n)))
        return w[:n.value], idx[:n.value]

    def gather_rows(self, gidx):
        gidx = np.ascontiguousarray(gidx, dtype=np.int64)
        m = gidx.size
        x = np.empty((m, self.d))
        mean = np.empty((m, self.G))
        var = np.empty((m, self.G))
        Qr = np.empty((m, 2 * self.G))
        if m:
            self.ctx.check(lib().sgp_grid_gather_rows(
                self.h, gidx.ctypes.data_as(c_i64_p), m, dptr(x), dptr(mean),
                dptr(var), dptr(Qr)))
        return x, mean, var, Qr

    def expander_check(self, gps, beta, fmin, xc, mu_c, u_c, near_frac=0.0):
        fmin = f64(fmin)
        xc = f64(xc).reshape(-1, self.d)
        m = xc.shape[0]
        mu_c = f64(mu_c).reshape(m, self.G)
        u_c = f64(u_c).reshape(m, self.G)
        flags = np.zeros((m, self.G), dtype=np.int32)
        self.ctx.check(lib().sgp_grid_expander_check(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin), m,
            dptr(xc), dptr(mu_c), dptr(u_c), float(near_frac),
            flags.ctypes.data_as(c_i32_p)))
        return flags

    def expander_batch(self, gps, beta, fmin, mode, cut_w, cut_idx, k):
        """The next ``k`` candidates behind the cut and whether they are expanders (exact
        scan), one device round trip: ``(widths, global indices, flags (m, G))``."""
        fmin = f64(fmin)
        w = np.empty(k)
        idx = np.empty(k, dtype=np.int64)
        n = C.c_int(0)
        flags = np.zeros((k, self.G), dtype=np.int32)
        self.ctx.check(lib().sgp_grid_expander_batch(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin), int(mode),
            float(cut_w), int(cut_idx), int(k), dptr(w), idx.ctypes.data_as(c_i64_p),
            C.byref(n), flags.ctypes.data_as(c_i32_p)))
        return w[:n.value], idx[:n.value], flags[:n.value]

    def expanders_small_all(self, gps, beta, fmin, cap=4096):
        """Every candidate of a small grid tested in one round trip
        (``sgp_grid_expanders_small_all``): ``(global rows in row order, widths, flags (m, G))``,
        or None when there are more than ``cap`` candidates."""
        fmin = f64(fmin)
        cap = int(min(cap, self.N))
        gidx = np.empty(cap, dtype=np.int64)
        width = np.empty(cap)
        flags = np.empty((cap, len(gps)), dtype=np.int32)
        n = C.c_int(0)
        self.ctx.check(lib().sgp_grid_expanders_small_all(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin), cap, C.byref(n),
            gidx.ctypes.data_as(c_i64_p), dptr(width), flags.ctypes.data_as(c_i32_p)))
        if n.value > cap:
            return None
        return gidx[:n.value], width[:n.value], flags[:n.value]

    def expander_pass(self, gps, beta, fmin, mode, cut_w, cut_idx, key_lo, key_hi, want,
                      scaling=None):
        """About ``want`` candidates behind the cut, all tested in one scan of the unsafe rows
        (``sgp_grid_expander_pass``): ``(tested, hits, key, row of the first expander in
        visiting order, key below which candidates are left or -inf, row of the arg-max of the
        step taken behind the test or -1 -- with ``scaling``, ``mode`` 0)``."""
        fmin = f64(fmin)
        sc = None if scaling is None else f64(scaling)
        out = np.zeros(6)
        self.ctx.check(lib().sgp_grid_expander_pass(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin), int(mode), float(cut_w),
            int(cut_idx), float(key_lo), float(key_hi), int(want),
            None if sc is None else dptr(sc), dptr(out)))
        return (int(out[0]), int(out[1]), float(out[2]), int(out[3]), float(out[4]), int(out[5]))

    def lipschitz_pass(self, fmin, lipschitz, mode, cut_w, cut_idx, key_lo, key_hi, want,
                       scaling=None):
        """The same pass with Lipschitz certificates (``sgp_grid_lipschitz_pass``); result as
        ``expander_pass``."""
        fmin, lipschitz = f64(fmin), f64(lipschitz)
        sc = None if scaling is None else f64(scaling)
        out = np.zeros(6)
        self.ctx.check(lib().sgp_grid_lipschitz_pass(
            self.h, len(fmin), dptr(fmin), dptr(lipschitz), int(mode), float(cut_w), int(cut_idx),
            float(key_lo), float(key_hi), int(want), None if sc is None else dptr(sc), dptr(out)))
        return (int(out[0]), int(out[1]), float(out[2]), int(out[3]), float(out[4]), int(out[5]))

    def pass_hist(self, mode, cut_w, cut_idx, key_lo, key_hi):
        """This shard's histogram (4096 bins over [key_lo, key_hi]) of the keys of its
        candidates behind the cut (``sgp_grid_pass_hist``)."""
        hist = np.zeros(4096, dtype=np.uint32)
        self.ctx.check(lib().sgp_grid_pass_hist(
            self.h, int(mode), float(cut_w), int(cut_idx), float(key_lo), float(key_hi),
            hist.ctypes.data_as(c_u32_p)))
        return hist

    def pass_list(self, mode, cut_w, cut_idx, thr, cap):
        """This shard's candidates behind the cut with key >= thr: ``(global rows, keys, rows
        (m, d), u - mu (m, G))`` (``sgp_grid_pass_list``)."""
        cap = max(int(cap), 1)
        gidx = np.empty(cap, dtype=np.int64)
        key = np.empty(cap)
        x = np.empty((cap, self.d))
        resid = np.empty((cap, self.G))
        n = C.c_int(0)
        self.ctx.check(lib().sgp_grid_pass_list(
            self.h, int(mode), float(cut_w), int(cut_idx), float(thr), cap, C.byref(n),
            gidx.ctypes.data_as(c_i64_p), dptr(key), dptr(x), dptr(resid)))
        m = n.value
        return gidx[:m], key[:m], x[:m], resid[:m]

    def pass_test(self, gps, beta, fmin, xc, resid):
        """Flags (K, G): candidate c lifts one of this shard's unsafe rows above fmin_i
        (``sgp_grid_pass_test``)."""
        fmin = f64(fmin)
        xc = f64(xc).reshape(-1, self.d)
        K = xc.shape[0]
        resid = f64(resid).reshape(K, self.G)
        flags = np.zeros((K, self.G), dtype=np.int32)
        if K:
            self.ctx.check(lib().sgp_grid_pass_test(
                self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin), K, dptr(xc),
                dptr(resid), flags.ctypes.data_as(c_i32_p)))
        return flags

    def pass_lipschitz_test(self, fmin, lipschitz, xc, u_c):
        """Flags (K, G) of the Lipschitz test of K gathered candidates against this shard's
        unsafe rows (``sgp_grid_pass_lipschitz_test``)."""
        fmin, lipschitz = f64(fmin), f64(lipschitz)
        xc = f64(xc).reshape(-1, self.d)
        K = xc.shape[0]
        u_c = f64(u_c).reshape(K, self.G)
        flags = np.zeros((K, self.G), dtype=np.int32)
        if K:
            self.ctx.check(lib().sgp_grid_pass_lipschitz_test(
                self.h, self.G, dptr(fmin), dptr(lipschitz), K, dptr(xc), dptr(u_c),
                flags.ctypes.data_as(c_i32_p)))
        return flags

    def lipschitz_check(self, fmin, lipschitz, xc, u_c):
        fmin = f64(fmin)
        lipschitz = f64(lipschitz)
        xc = f64(xc).reshape(-1, self.d)
        m = xc.shape[0]
        u_c = f64(u_c).reshape(m, self.G)
        flags = np.zeros((m, self.G), dtype=np.int32)
        self.ctx.check(lib().sgp_grid_lipschitz_check(
            self.h, self.G, dptr(fmin), dptr(lipschitz), m, dptr(xc),
            dptr(u_c), flags.ctypes.data_as(c_i32_p)))
        return flags

    def sets_front(self, max_l, max_var, scaling, thr_beta):
        scaling = f64(scaling)
        thr_beta = f64(thr_beta)
        out5 = np.empty(6)           # [5] = candidates tied with the first one
        x = np.empty(self.d)
        mean = np.empty(self.G)
        q = np.empty(2 * self.G)
        self.ctx.check(lib().sgp_grid_sets_front(
            self.h, float(max_l), int(max_var is not None),
            0.0 if max_var is None else float(max_var), dptr(scaling),
            dptr(thr_beta), dptr(out5), dptr(x), dptr(mean), dptr(q)))
        return out5, x, mean, q

    def sets_front_comm(self, scaling, thr_beta):
        scaling, thr_beta = f64(scaling), f64(thr_beta)
        out5 = np.empty(6)
        x = np.empty(self.d)
        mean = np.empty(self.G)
        q = np.empty(2 * self.G)
        ml = C.c_double(0)
        self.ctx.check(lib().sgp_grid_sets_front_comm(
            self.h, dptr(scaling), dptr(thr_beta), dptr(out5), dptr(x),
            dptr(mean), dptr(q), C.byref(ml)))
        return out5, x, mean, q, ml.value

    def sets_back(self, gps, beta, fmin, xc, mu_c, u_c, near_frac, gidx_c,
                  scaling, mark=True):
        fmin = f64(fmin)
        scaling = f64(scaling)
        xc, mu_c, u_c = f64(xc), f64(mu_c), f64(u_c)
        flags = np.zeros(self.G, dtype=np.int32)
        v = C.c_double(0)
        i = C.c_int64(0)
        self.ctx.check(lib().sgp_grid_sets_back(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin),
            dptr(xc), dptr(mu_c), dptr(u_c), float(near_frac), int(gidx_c),
            int(bool(mark)), dptr(scaling), flags.ctypes.data_as(c_i32_p),
            C.byref(v), C.byref(i)))
        return flags, v.value, i.value

    def sets_fused(self, gps, beta, fmin, max_l, scaling, thr_beta, near_frac):
        fmin, scaling, thr_beta = f64(fmin), f64(scaling), f64(thr_beta)
        out5 = np.empty(6)           # [5] = candidates tied with the first one
        x = np.empty(self.d)
        mean = np.empty(self.G)
        q = np.empty(2 * self.G)
        flags = np.zeros(self.G, dtype=np.int32)
        v = C.c_double(0)
        i = C.c_int64(0)
        ml = C.c_double(0)
        self.ctx.check(lib().sgp_grid_sets_fused(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin),
            float('nan') if max_l is None else float(max_l), dptr(scaling),
            dptr(thr_beta), float(near_frac), dptr(out5), dptr(x), dptr(mean),
            dptr(q), flags.ctypes.data_as(c_i32_p), C.byref(v), C.byref(i),
            C.byref(ml)))
        return out5, x, mean, q, flags, v.value, i.value, ml.value

    def expanders_small(self, gps, beta, fmin, gidx):
        """Which of the candidate rows ``gidx`` are expanders, per GP: ``(m, G)`` flags in
        one round trip (small grids: ``sgp_grid_expanders_small``)."""
        gidx = np.ascontiguousarray(gidx, dtype=np.int64)
        flags = np.zeros((gidx.size, self.G), dtype=np.int32)
        if gidx.size:
            self.ctx.check(lib().sgp_grid_expanders_small(
                self.h, _gp_array(gps), len(gps), float(beta), dptr(f64(fmin)),
                gidx.ctypes.data_as(c_i64_p), int(gidx.size), flags.ctypes.data_as(c_i32_p)))
        return flags

    def step_small_ok(self, gps):
        """Does ``step_small`` serve this grid with these (fitted) GPs?"""
        return bool(lib().sgp_grid_step_small_ok(self.h, _gp_array(gps), len(gps)))

    def step_small(self, gps, beta, fmin, scaling, thr_beta):
        """A whole ``SafeOpt.optimize()`` of a small grid in one launch and one read-back
        (``sgp_grid_step_small``); returns what ``sets_fused`` returns.  The output
        buffers and their pointers are built once (this is the 30-microsecond path)."""
        ss = self.__dict__.get("_ss")
        if ss is None:
            out5, x, mean, q = np.empty(6), np.empty(self.d), np.empty(self.G), np.empty(2 * self.G)
            flags = np.zeros(self.G, dtype=np.int32)
            v, i, ml = C.c_double(0), C.c_int64(0), C.c_double(0)
            ss = self._ss = dict(
                out=(out5, x, mean, q, flags, v, i, ml),
                ptr=(dptr(out5), dptr(x), dptr(mean), dptr(q), flags.ctypes.data_as(c_i32_p),
                     C.byref(v), C.byref(i), C.byref(ml)),
                fn=lib().sgp_grid_step_small, gps=None, gp_arr=None)
        key = tuple(g.h.value for g in gps)
        if ss["gps"] != key:
            ss["gps"], ss["gp_arr"] = key, _gp_array(gps)
        # (the three input arrays are the optimiser's own attributes: their pointers are
        # looked up once per array object -- the cache holds the arrays, so an id is not reused)
        ins = ss.get("ins")
        if ins is None or ins[0] is not fmin or ins[1] is not scaling or ins[2] is not thr_beta:
            # (f64: the same object back when it already is a contiguous float64 array)
            conv = (f64(fmin), f64(scaling), f64(thr_beta))
            ins = ss["ins"] = (fmin, scaling, thr_beta) + conv + tuple(dptr(a) for a in conv)
        elif ins[3] is not fmin or ins[4] is not scaling:
            # converted copies (an integer fmin, say): their VALUES may have been edited
            ins[3][...], ins[4][...], ins[5][...] = fmin, scaling, thr_beta
        rc = ss["fn"](self.h, ss["gp_arr"], len(gps), float(beta), ins[6], ins[7], ins[8],
                      *ss["ptr"])
        if rc != 0:
            self.ctx.check(rc)
        out5, x, mean, q, flags, v, i, ml = ss["out"]
        return out5, x, mean, q, flags, v.value, i.value, ml.value

    def sets_fused_comm(self, gps, beta, fmin, scaling, thr_beta, near_frac):
        """N-rank certified step in one round trip (merges behind in-stream collectives)."""
        fmin, scaling, thr_beta = f64(fmin), f64(scaling), f64(thr_beta)
        out5 = np.empty(6)
        x = np.empty(self.d)
        mean = np.empty(self.G)
        q = np.empty(2 * self.G)
        flags = np.zeros(self.G, dtype=np.int32)
        v = C.c_double(0)
        i = C.c_int64(0)
        ml = C.c_double(0)
        self.ctx.check(lib().sgp_grid_sets_fused_comm(
            self.h, _gp_array(gps), len(gps), float(beta), dptr(fmin), dptr(scaling),
            dptr(thr_beta), float(near_frac), dptr(out5), dptr(x), dptr(mean),
            dptr(q), flags.ctypes.data_as(c_i32_p), C.byref(v), C.byref(i),
            C.byref(ml)))
        return out5, x, mean, q, flags, v.value, i.value, ml.value

    def mark_expanders(self, gidx):
        gidx = np.ascontiguousarray(gidx, dtype=np.int64)
        if gidx.size:
            self.ctx.check(lib().sgp_grid_mark_expanders(
                self.h, gidx.ctypes.data_as(c_i64_p), gidx.size))

    def unmark_expanders(self, gidx):
        gidx = np.ascontiguousarray(gidx, dtype=np.int64)
        if gidx.size:
            self.ctx.check(lib().sgp_grid_unmark_expanders(
                self.h, gidx.ctypes.data_as(c_i64_p), gidx.size))

    def argmax(self, mode, scaling):
        scaling = f64(scaling)
        v = C.c_double(0)
        i = C.c_int64(0)
        self.ctx.check(lib().sgp_grid_argmax(self.h, int(mode), dptr(scaling),
                                             C.byref(v), C.byref(i)))
        return v.value, i.value

    def download(self, what, out=None):
        if what == Q:
            shape, dt = (self.N, 2 * self.G), np.float64
        elif what in (S, M, G, CAND):
            shape, dt = (self.N,), np.uint8
        elif what == WIDTH:
            shape, dt = (self.N,), np.float64
        else:
            shape, dt = (self.G, self.N), np.float64
        buf = np.empty(shape, dtype=dt)
        self.ctx.check(lib().sgp_grid_download(self.h, what,
                                               buf.ctypes.data_as(vp)))
        if dt == np.uint8:
            buf = buf.view(np.bool_)
        if out is not None:
            np.copyto(out, buf)
            return out
        return buf


def swarm_grow(ctx, gp, S, B, scale2, thr=0.95):
    """Rows of ``B`` to append to the safe set ``S`` (bool mask, in order)."""
    d = gp.d
    S = f64(S).reshape(-1, d)
    B = f64(B).reshape(-1, d)
    accept = np.zeros(B.shape[0], dtype=np.uint8)
    if B.shape[0]:
        ctx.check(lib().sgp_swarm_grow(
            ctx.h, gp.h, dptr(S), S.shape[0], dptr(B), B.shape[0],
            float(scale2), float(thr), accept.ctypes.data_as(c_u8_p)))
    return accept.view(np.bool_)


def swarm_run(ctx, gps, swarm_type, beta, fmin, scaling, best_lower_bound,
              positions, velocities, best_positions, best_values, global_best,
              velocity_scale, bounds, init, iters, inertia0, step, rand, seed=0):
    """Whole PSO run on the device; the state arrays are updated in place."""
    P = positions.shape[0]
    for a in (positions, velocities, best_positions, best_values, global_best):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    bnd = None if bounds is None else f64(bounds)
    rnd = None if rand is None else f64(rand).ravel()
    ctx.check(lib().sgp_swarm_run(
        ctx.h, _gp_array(gps), len(gps), SWARM_TYPES[swarm_type], float(beta),
        dptr(f64(fmin)), dptr(f64(scaling)), float(best_lower_bound), P,
        dptr(positions), dptr(velocities), dptr(best_positions),
        dptr(best_values), dptr(global_best), dptr(f64(velocity_scale)),
        None if bnd is None else dptr(bnd), int(bool(init)), int(iters),
        float(inertia0), float(step), None if rnd is None else dptr(rnd),
        int(seed)))


def swarm_fitness(ctx, gps, swarm_type, particles, beta, fmin, scaling,
                  best_lower_bound):
    d = gps[0].d
    particles = f64(particles).reshape(-1, d)
    P = particles.shape[0]
    fmin = f64(fmin)
    scaling = f64(scaling)
    values = np.empty(P)
    safe = np.empty(P, dtype=np.uint8)
    if P:
        ctx.check(lib().sgp_swarm_fitness(
            ctx.h, _gp_array(gps), len(gps), SWARM_TYPES[swarm_type],
            dptr(particles), P, float(beta), dptr(fmin), dptr(scaling),
            float(best_lower_bound), dptr(values),
            safe.ctypes.data_as(c_u8_p)))
    return values, safe.view(np.bool_)
