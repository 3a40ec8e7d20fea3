"""Row-sharding of the candidate grid over the GPUs of one node.

One process per GPU.  Candidate rows are independent for the posterior sweep
and every masked pass; the only cross-rank traffic of SafeOpt's path is a few
scalars per iteration:

  * ``max(l[S])`` / ``any(S)``, ``max_var``, candidate / unsafe counts,
  * the merge of each rank's next-k expander candidates,
  * the any-flags of the expander test,
  * the final ``(value, index)`` arg-max (lowest global index wins).

``RcclComm`` runs them as RCCL collectives over xGMI on the context's HIP
stream.  The rendezvous (broadcast of the 128-byte ncclUniqueId) needs no
third-party runtime: rank 0 serves it on MASTER_ADDR:MASTER_PORT.
"""
from __future__ import annotations

import os
import socket
import struct
import time

import numpy as np

__all__ = ["shard_range", "LocalComm", "RcclComm", "init_from_env",
           "merge_topk", "merge_argmax"]


def shard_range(N, rank, world):
    """Contiguous block of the flat grid index owned by ``rank``.

    Blocks differ by at most one row; global index = offset + local row, so
    "lowest global index wins" stays trivial after sharding.
    """
    base, rem = divmod(int(N), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class LocalComm(object):
    """world_size == 1: every collective is the identity."""
    rank, world = 0, 1

    def allreduce_max(self, a):
        return np.array(a, dtype=np.float64, copy=True)

    def allgather(self, a):
        return np.asarray(a)[None, ...]

    def barrier(self):
        pass


class RcclComm(object):
    """RCCL collectives (through libsafeopt_hip.so) on small host arrays.

    ``in_stream``: the library may also run scalar all-reduces on device-
    resident operands inside a fused call (``sgp_grid_sets_front_comm``)."""
    in_stream = True

    def __init__(self, ctx):
        self.ctx = ctx
        self.rank, self.world = ctx.rank, ctx.world

    def allreduce_max(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return self.ctx.allreduce_max(a.ravel()).reshape(a.shape)

    def allgather(self, a):
        a = np.ascontiguousarray(a)
        raw = self.ctx.allgather_bytes(a.tobytes())
        return np.frombuffer(raw, dtype=a.dtype).reshape((self.world,) + a.shape)

    def barrier(self):
        self.ctx.barrier()


def _serve_uid(addr, port, uid, world, timeout):
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind((addr, port))
    srv.listen(world)
    srv.settimeout(timeout)
    conns = []
    try:
        for _ in range(world - 1):
            c, _a = srv.accept()
            conns.append(c)
        for c in conns:
            c.sendall(struct.pack("<I", len(uid)) + uid)
    finally:
        for c in conns:
            c.close()
        srv.close()


def _fetch_uid(addr, port, timeout):
    deadline = time.time() + timeout
    while True:
        try:
            s = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    with s:
        s.settimeout(timeout)
        hdr = b""
        while len(hdr) < 4:
            hdr += s.recv(4 - len(hdr))
        n = struct.unpack("<I", hdr)[0]
        buf = b""
        while len(buf) < n:
            chunk = s.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("rendezvous connection closed early")
            buf += chunk
    return buf


def _file_rendezvous(rank, world, port, timeout):
    """Single-node exchange of the ncclUniqueId through /tmp.

    All workers of one ``torch.distributed.run`` launch share the agent as
    parent process, so (parent pid, MASTER_PORT) names the launch.
    """
    from . import _hip
    path = os.path.join(os.environ.get("SAFEOPT_RDZV_DIR", "/tmp"),
                        "safeopt_rdzv_%d_%d.bin" % (os.getppid(), port))
    if rank == 0:
        uid = _hip.Context.comm_unique_id()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)                  # atomic publish
        return uid, path
    deadline = time.time() + timeout
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == 128:
                return uid, None
        except OSError:
            pass
        if time.time() > deadline:
            raise RuntimeError("rendezvous file %s did not appear" % path)
        time.sleep(0.02)


def init_from_env(ctx=None, timeout=300.0):
    """Create the communicator from the torchrun-style environment
    (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT).

    Returns ``(ctx, comm)``; with ``WORLD_SIZE`` unset or 1 it is a
    ``LocalComm`` and no RCCL call is made (``SAFEOPT_FORCE_RCCL=1`` builds a
    one-rank RCCL communicator instead, to exercise that path).  The
    ncclUniqueId travels through a file in /tmp when MASTER_ADDR is local
    (one node -- the supported topology), otherwise over a TCP socket on
    MASTER_PORT+17 (``SAFEOPT_RDZV=tcp|file`` overrides).
    """
    from . import _hip
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if ctx is None:
        ctx = _hip.Context.default()
    force = os.environ.get("SAFEOPT_FORCE_RCCL", "0") == "1"
    if world <= 1 and not force:
        return ctx, LocalComm()
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    mport = int(os.environ.get("MASTER_PORT", "29500"))
    mode = os.environ.get("SAFEOPT_RDZV",
                          "file" if addr in ("127.0.0.1", "localhost", "::1")
                          else "tcp")
    cleanup = None
    if mode == "file":
        uid, cleanup = _file_rendezvous(rank, world, mport, timeout)
    else:
        # a port next to the launcher's, so a c10d store on MASTER_PORT can coexist
        port = int(os.environ.get("SAFEOPT_RDZV_PORT", mport + 17))
        if rank == 0:
            uid = _hip.Context.comm_unique_id()
            if world > 1:
                _serve_uid(addr, port, uid, world, timeout)
        else:
            uid = _fetch_uid(addr, port, timeout)
    ctx.comm_init(uid, rank, world)            # returns once every rank joined
    comm = RcclComm(ctx)
    comm.barrier()
    if cleanup:
        try:
            os.remove(cleanup)
        except OSError:
            pass
    return ctx, comm


# ---------------------------------------------------------------------------
# rank-independent merges (pure functions; identical result on every rank)
# ---------------------------------------------------------------------------
def merge_topk(ws, idxs, k, by_index=False):
    """Merge per-rank candidate lists into the next ``k`` in visiting order.

    Visiting order of the reference (``gp_opt.py:542-552``): descending
    interval width; exact ties resolve to the higher global index first (what
    a stable ascending sort, reversed, gives).  ``by_index``: ascending global
    index (``full_sets=True``, ``gp_opt.py:553-555``).  Entries with index < 0
    are padding.
    """
    w = np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in ws])
    i = np.concatenate([np.asarray(x, dtype=np.int64).ravel() for x in idxs])
    keep = i >= 0
    w, i = w[keep], i[keep]
    if by_index:
        order = np.argsort(i, kind="stable")
    else:
        order = np.lexsort((-i, -w))       # primary: w desc, then index desc
    order = order[:k]
    return w[order], i[order]


def merge_argmax(values, idxs):
    """Global arg-max from per-rank ``(value, global index)`` pairs; equal
    values resolve to the lowest global index (``np.argmax`` on the unsharded
    array, ``gp_opt.py:635, 644, 710``).  Index < 0 = empty shard."""
    values = np.asarray(values, dtype=np.float64).ravel()
    idxs = np.asarray(idxs, dtype=np.int64).ravel()
    keep = idxs >= 0
    if not keep.any():
        return -np.inf, -1
    values, idxs = values[keep], idxs[keep]
    best = np.lexsort((idxs, -values))[0]
    return float(values[best]), int(idxs[best])
