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

__all__ = ["shard_range", "LocalComm", "RcclComm", "SocketComm", "init_from_env",
           "merge_topk", "merge_argmax", "launch_check"]


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
        # SAFEOPT_RCCL_IN_STREAM=0: keep the collectives on the host side of the step (the
        # variant of the N-rank step that SocketComm takes: three round trips instead of
        # one) -- a switch for the operator, the default is the in-stream step
        if os.environ.get("SAFEOPT_RCCL_IN_STREAM", "1") == "0":
            self.in_stream = False

    def allreduce_max(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return self.ctx.allreduce_max(a.ravel()).reshape(a.shape)

    def allgather(self, a):
        a = np.ascontiguousarray(a)
        raw = self.ctx.allgather_bytes(a.tobytes())
        return np.frombuffer(raw, dtype=a.dtype).reshape((self.world,) + a.shape)

    def barrier(self):
        self.ctx.barrier()


class SocketComm(object):
    """The collectives of ``RcclComm`` over TCP, through rank 0 (a star).

    RCCL wants one GPU per rank; this communicator does not care where the ranks'
    kernels run, so N processes -- each with its own HIP context and its TRUE shard
    of the grid -- can share one GPU (``SAFEOPT_COMM=socket``; how the N-rank product
    is tested on a one-GPU box, tests/test_gpu_nrank.py), or sit on machines without
    xGMI.  The payloads of the path are a few dozen bytes per collective
    (SURVEY.md section 8e), latency is all that matters.

    With a device context the communicator registers itself as that context's transport
    (``sgp_comm_init_host``): the library's one-round-trip N-rank step
    (``sgp_grid_sets_fused_comm``: first-candidate merge, flag all-reduce and arg-max
    merge on the device) then runs over it as it does over RCCL, the collectives staged
    through the host at the points where RCCL's are enqueued in stream (``in_stream``
    True).  ``SAFEOPT_SOCKET_IN_STREAM=0`` keeps the collectives on the host side of the
    step instead (``sets_front`` / ``sets_back``: three round trips).
    """
    in_stream = False

    def __init__(self, rank, world, addr="127.0.0.1", port=29517, timeout=300.0, ctx=None):
        self.rank, self.world, self.ctx = int(rank), int(world), ctx
        self._peers = []          # rank 0: sockets of ranks 1 .. world - 1
        self._up = None           # other ranks: socket to rank 0
        if (ctx is not None and hasattr(ctx, "comm_init_host")
                and os.environ.get("SAFEOPT_SOCKET_IN_STREAM", "1") != "0"):
            ctx.comm_init_host(self)
            self.in_stream = True
        if self.world <= 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            peers = {}
            try:
                while len(peers) < self.world - 1:
                    c, _a = srv.accept()
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(timeout)
                    r = struct.unpack("<I", self._recv_exact(c, 4))[0]
                    if not 0 < r < self.world or r in peers:
                        raise RuntimeError("rendezvous: unexpected rank %d" % r)
                    peers[r] = c
            finally:
                srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout)
            c.sendall(struct.pack("<I", self.rank))
            self._up = c

    @staticmethod
    def _recv_exact(c, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("peer closed the connection")
            buf += chunk
        return bytes(buf)

    def _gather_bytes(self, raw):
        """Every rank's bytes, in rank order, on every rank."""
        if self.world <= 1:
            return [raw]
        if self.rank == 0:
            parts = [raw]
            for c in self._peers:
                n = struct.unpack("<Q", self._recv_exact(c, 8))[0]
                parts.append(self._recv_exact(c, n))
            blob = b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for c in self._peers:
                c.sendall(struct.pack("<Q", len(blob)) + blob)
            return parts
        self._up.sendall(struct.pack("<Q", len(raw)) + raw)
        n = struct.unpack("<Q", self._recv_exact(self._up, 8))[0]
        blob = self._recv_exact(self._up, n)
        parts, at = [], 0
        for _ in range(self.world):
            m = struct.unpack_from("<Q", blob, at)[0]
            parts.append(blob[at + 8:at + 8 + m])
            at += 8 + m
        return parts

    def allreduce_max(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        parts = self._gather_bytes(a.tobytes())
        out = np.frombuffer(parts[0], dtype=np.float64).copy()
        for x in parts[1:]:
            # (max as RCCL's ncclMax: NaN-free operands on this path)
            np.maximum(out, np.frombuffer(x, dtype=np.float64), out=out)
        return out.reshape(a.shape)

    def allgather(self, a):
        a = np.ascontiguousarray(a)
        parts = self._gather_bytes(a.tobytes())
        return np.stack([np.frombuffer(x, dtype=a.dtype).reshape(a.shape) for x in parts])

    def barrier(self):
        self._gather_bytes(b"")

    def close(self):
        for c in self._peers + ([self._up] if self._up else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._up = [], None


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
            chunk = s.recv(4 - len(hdr))
            if not chunk:
                raise RuntimeError("rendezvous connection closed early")
            hdr += chunk
        n = struct.unpack("<I", hdr)[0]
        buf = b""
        while len(buf) < n:
            chunk = s.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("rendezvous connection closed early")
            buf += chunk
    return buf


def _launch_tag(port):
    """16 bytes naming THIS launch: every rank of one launch derives the same
    tag (launcher pid, MASTER_PORT, the launcher's run id / nonce and its restart
    count), a file left behind by another launch -- or by the workers the same
    launcher started before a restart (``torchrun --max-restarts``) -- does not
    carry it."""
    import hashlib
    nonce = os.environ.get("SAFEOPT_RDZV_NONCE",
                           os.environ.get("TORCHELASTIC_RUN_ID", ""))
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    return hashlib.md5(("%d:%d:%s:%s" % (os.getppid(), port, nonce, restart)).encode()).digest()


def _rdzv_dir():
    """A directory only this user can write (0700, ownership checked)."""
    d = os.environ.get("SAFEOPT_RDZV_DIR")
    if d is None:
        d = os.path.join("/tmp", "safeopt-rdzv-%d" % os.geteuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.geteuid() or (st.st_mode & 0o022):
        raise RuntimeError("rendezvous directory %s is not private to this user" % d)
    return d


def _launcher_start():
    """Start time (epoch seconds) of the launching process; 0 if /proc does not
    tell.  From the ``starttime`` field of /proc/<pid>/stat and the boot time --
    the timestamps of the /proc/<pid> directory itself are those of the first
    lookup, not of the process."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            fields = f.read().rsplit(")", 1)[1].split()
        ticks = float(fields[19])              # field 22 of the full line
        with open("/proc/stat") as f:
            btime = next(float(l.split()[1]) for l in f if l.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, IndexError, ValueError, StopIteration):
        return 0.0


def _file_rendezvous(rank, world, port, timeout):
    """Single-node exchange of the ncclUniqueId through a private directory.

    All workers of one launch (``torch.distributed.run`` or
    ``bench.py --gpus N``) share the launcher as parent process, so
    (parent pid, MASTER_PORT, run id) names the launch.  Rank 0 removes any
    left-over file first and publishes ``uid + tag`` atomically (O_EXCL
    temporary, 0600, rename); readers only accept a file that carries this
    launch's tag and is not older than the launcher itself.
    """
    from . import _hip
    tag = _launch_tag(port)
    path = os.path.join(_rdzv_dir(),
                        "rdzv_%d_%d_%s.bin" % (os.getppid(), port, tag.hex()[:8]))
    if rank == 0:
        try:
            os.unlink(path)                    # a crashed launch's left-over
        except OSError:
            pass
        uid = _hip.Context.comm_unique_id()
        tmp = path + ".tmp%d" % os.getpid()
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid + tag)
        os.replace(tmp, path)                  # atomic publish
        return uid, path
    deadline = time.time() + timeout
    born = _launcher_start() - 2.0         # (file times may be coarser than ours)
    while True:
        try:
            if os.stat(path).st_mtime >= born:
                with open(path, "rb") as f:
                    blob = f.read()
                if len(blob) == 144 and blob[128:] == tag:
                    return blob[:128], None
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
    one-rank RCCL communicator instead, to exercise that path);
    ``SAFEOPT_COMM=socket`` takes host-side collectives over TCP (``SocketComm``:
    ranks that share a GPU, or no xGMI between them) instead of RCCL.  The
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
    if os.environ.get("SAFEOPT_COMM", "rccl") == "socket":
        # host-side collectives: the ranks need not have a GPU each
        comm = SocketComm(rank, world, addr, int(os.environ.get("SAFEOPT_RDZV_PORT", mport + 17)),
                          timeout, ctx=ctx)
        comm.barrier()
        return ctx, comm
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


def launch_check(timeout=60.0):
    """Rendezvous of a launch WITHOUT a device (``bench.py --launch-check``):
    rank 0 serves a 128-byte token over the TCP rendezvous, every other rank
    fetches it; returns ``(rank, world, token_ok)``.  Exercises the launcher's
    environment (RANK / WORLD_SIZE / MASTER_*) and the socket path on a CPU box.
    """
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + 17
    token = _launch_tag(port) * 8
    if world <= 1:
        return rank, world, True
    if rank == 0:
        _serve_uid(addr, port, token, world, timeout)
        return rank, world, True
    return rank, world, _fetch_uid(addr, port, timeout) == token


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
