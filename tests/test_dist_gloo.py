"""N > 1 path on CPU: two processes over gloo drive the product's sharded
SafeOpt host logic (shard ranges, phase driver, top-k / arg-max merges) with the
NumPy oracle standing in for the per-rank HIP kernels.  Result must equal the
reference's golden vectors, i.e. the unsharded run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class TorchComm(object):
    """Same interface as safeopt_amd.dist.RcclComm, on torch.distributed/gloo."""

    def __init__(self):
        import torch.distributed as td
        self.td = td
        self.rank, self.world = td.get_rank(), td.get_world_size()

    def allreduce_max(self, a):
        import torch
        t = torch.from_numpy(np.array(a, dtype=np.float64, copy=True))
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return t.numpy()

    def allgather(self, a):
        import torch
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.td.all_gather(outs, t)
        return np.stack([o.numpy().view(a.dtype).reshape(a.shape) for o in outs])

    def barrier(self):
        self.td.barrier()


def _worker(rank, world, port, names, q):
    try:
        sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
        import torch.distributed as td
        td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                              rank=rank, world_size=world)
        import safeopt_amd
        from oracle import gp_numpy as gpn
        from _golden import load, make_kernel
        from _oracle_backend import OracleGridBackend, use_oracle_backend
        use_oracle_backend()
        comm = TorchComm()
        report = []
        for name in names:
            z, meta = load(name)
            if "recorded" in meta:
                its = meta["recorded"]
            else:
                its = [None]
            for t in its:
                pre = "" if t is None else "it%d_" % t
                gps = [gpn.GPRegression(z[pre + "X%d" % i], z[pre + "Y%d" % i],
                                        make_kernel(gpn, spec), noise_var=meta["noise_vars"][i])
                       for i, spec in enumerate(meta["kernels"])]
                lip = meta.get("lipschitz")
                if lip is not None and len(lip) == 1:
                    lip = lip[0]
                beta = meta["beta"] if t is None else float(z["beta_all"][t])
                G = len(gps)
                opt = safeopt_amd.SafeOpt(
                    gps if G > 1 else gps[0], z["parameter_set"],
                    meta["fmin"] if G > 1 else meta["fmin"][0], lipschitz=lip, beta=beta,
                    threshold=meta["threshold"], num_contexts=meta.get("num_contexts", 0),
                    comm=comm)
                lo, hi = opt._shard
                assert hi - lo < z["parameter_set"].shape[0]        # really sharded
                ctx = z[pre + "context"] if meta.get("num_contexts") else None
                x = opt.optimize(context=ctx, ucb=meta.get("ucb", False))
                ok = (np.array_equal(x, z[pre + "x_next"]) and
                      np.array_equal(opt.S, z[pre + "S"]) and
                      np.allclose(opt.Q, z[pre + "Q"], atol=1e-10, rtol=0))
                if not meta.get("ucb", False):
                    ok = ok and np.array_equal(opt.M, z[pre + "M"]) and \
                        np.array_equal(opt.G, z[pre + "G"])
                    if t is not None:
                        mx = opt.get_maximum(context=ctx)
                        ok = ok and np.array_equal(mx[0], z[pre + "max_x"])
                report.append((name, t, bool(ok)))
        td.destroy_process_group()
        q.put((rank, report, None))
    except Exception as e:                      # surface the traceback in rank order
        import traceback
        q.put((rank, [], traceback.format_exc()))


@pytest.mark.timeout(600)
def test_two_ranks_reproduce_unsharded_golden():
    pytest.importorskip("torch")
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    names = ["safeopt_2d_rbf", "safeopt_1d_multi", "safeopt_1d_lipschitz", "safeopt_context",
             "safeopt_2d_ucb", "sets_1d_seed7", "sets_1d_g2_seed0", "sets_2d_seed3"]
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, names, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=550) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, report, err in sorted(results):
        assert err is None, "rank %d failed:\n%s" % (rank, err)
        assert report and all(ok for _, _, ok in report), report


def _pass_worker(rank, world, port, q):
    try:
        sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
        import torch.distributed as td
        td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                              rank=rank, world_size=world)
        import safeopt_amd
        import _scenarios as sc
        from oracle import gp_numpy as gpn
        from oracle import safeopt_numpy as son
        from _oracle_backend import OracleGridBackend, use_oracle_backend
        use_oracle_backend()
        calls = {"gp": 0, "lip": 0}
        for name, key in (("pass_test", "gp"), ("pass_lipschitz_test", "lip")):
            orig = getattr(OracleGridBackend, name)
            setattr(OracleGridBackend, name,
                    lambda self, *a, _o=orig, _k=key: (calls.__setitem__(_k, calls[_k] + 1), _o(self, *a))[1])
        comm = TorchComm()
        # a converged state: 48 candidates, no expander (tests/test_gpu_expander_passes.py at CPU size)
        data = sc.rim_data(40, ls=0.8, rings=4, dring=0.4, dmid=1.0, dtop=0.5, r0=2.0, dout=2.0,
                           plateau=0.6)
        grid = data["grid"]
        report = []
        for lip in (None, 1.0, 0.7):
            go = sc.make_gp(gpn, data)
            opt = safeopt_amd.SafeOpt(sc.make_gp(gpn, data), grid, 0.0, lipschitz=lip, threshold=0.1,
                                      comm=comm)
            opt.pass_sizes = (8, 16)
            before = dict(calls)
            x = opt.optimize()
            idx, Q, S, M, G = son.optimize_grid([go], grid, [0.0], opt.scaling, 0.1, 2.0,
                                                lipschitz=None if lip is None else [lip])
            ok = (np.array_equal(x, grid[idx]) and np.array_equal(opt.S, S)
                  and np.array_equal(opt.M, M) and np.array_equal(opt.G, G))
            ran = calls["gp" if lip is None else "lip"] - before["gp" if lip is None else "lip"]
            report.append(("lipschitz=%s: |G| = %d, %d N-rank passes" % (lip, int(G.sum()), ran),
                           bool(ok) and (ran >= 2 or G.any())))
        td.destroy_process_group()
        q.put((rank, report, None))
    except Exception:
        import traceback
        q.put((rank, [], traceback.format_exc()))


@pytest.mark.timeout(600)
def test_two_ranks_big_passes_of_the_expander_loop():
    """``SafeOpt._visit_in_big_passes_nrank`` over gloo with the oracle's arithmetic behind the
    three calls of a pass (histograms summed, candidates gathered, flags or-ed): a converged
    state -- every candidate visited in passes of 8 and 16, none marked -- with GP and with
    Lipschitz certificates, equal to the unsharded oracle."""
    pytest.importorskip("torch")
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_pass_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=550) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, report, err in sorted(results):
        assert err is None, "rank %d failed:\n%s" % (rank, err)
        assert report and all(ok for _, ok in report), report


def test_three_way_shard_single_process():
    """Same driver, world = 3 emulated in one process (each 'rank' in turn with
    a communicator that replays the other ranks): covers uneven shards."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import safeopt_amd
    from oracle import gp_numpy as gpn
    from _golden import load, make_kernel
    from _oracle_backend import OracleGridBackend, use_oracle_backend
    use_oracle_backend()
    import threading

    class ThreadComm(object):
        """world ranks as threads of this process, rendezvous on a barrier."""
        def __init__(self, rank, world, shared):
            self.rank, self.world, self.sh = rank, world, shared

        def _exchange(self, a):
            self.sh["slots"][self.rank] = np.array(a, copy=True)
            self.sh["bar"].wait()
            out = [np.array(s, copy=True) for s in self.sh["slots"]]
            self.sh["bar"].wait()
            return out

        def allreduce_max(self, a):
            return np.max(np.stack(self._exchange(np.asarray(a, dtype=float))), axis=0)

        def allgather(self, a):
            return np.stack(self._exchange(np.asarray(a)))

        def barrier(self):
            self.sh["bar"].wait()

    z, meta = load("sets_1d_seed7")
    world = 3
    shared = dict(slots=[None] * world, bar=threading.Barrier(world))
    out = [None] * world

    def run(rank):
        gp = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                              noise_var=meta["noise_vars"][0])
        opt = safeopt_amd.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"],
                                  comm=ThreadComm(rank, world, shared))
        x = opt.optimize()
        out[rank] = (x, opt.S.copy(), opt.M.copy(), opt.G.copy())

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ths]; [t.join(120) for t in ths]
    for r in range(world):
        x, S, M, G = out[r]
        assert np.array_equal(x, z["x_next"]) and np.array_equal(S, z["S"])
        assert np.array_equal(M, z["M"]) and np.array_equal(G, z["G"])


def test_socket_comm_three_ranks_collectives_and_driver():
    """dist.SocketComm (the interface of RcclComm over a TCP star through rank 0 -- what the
    N-rank product runs on when several ranks share one GPU, tests/test_gpu_nrank.py):
    the collectives themselves on three ranks, then the host phase driver on uneven
    shards against the unsharded golden vectors.  (Ranks as threads of this process.)"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import threading
    import safeopt_amd
    from safeopt_amd import dist
    from oracle import gp_numpy as gpn
    from _golden import load, make_kernel
    from _oracle_backend import use_oracle_backend
    use_oracle_backend()
    z, meta = load("sets_1d_seed7")
    world, port = 3, _free_port()
    out, errs = [None] * world, []

    def run(rank):
        try:
            comm = dist.SocketComm(rank, world, "127.0.0.1", port, timeout=60.0)
            assert not comm.in_stream
            m = comm.allreduce_max(np.array([float(rank), -float(rank), 7.0]))
            assert np.array_equal(m, [2.0, 0.0, 7.0])
            g = comm.allgather(np.array([[rank, 10 * rank]], dtype=np.int64))
            assert g.shape == (3, 1, 2) and np.array_equal(g[:, 0, 1], [0, 10, 20])
            assert comm.allgather(np.zeros(0)).shape == (3, 0)
            comm.barrier()
            gp = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                                  noise_var=meta["noise_vars"][0])
            opt = safeopt_amd.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"],
                                      comm=comm)
            x = opt.optimize()
            out[rank] = (x, opt.S.copy(), opt.M.copy(), opt.G.copy(), opt._shard)
            comm.barrier()
            comm.close()
        except Exception:
            import traceback
            errs.append(traceback.format_exc())

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ths]; [t.join(120) for t in ths]
    assert not errs, errs[0]
    assert len({o[4] for o in out}) == world          # three different shards
    for x, S, M, G, _ in out:
        assert np.array_equal(x, z["x_next"]) and np.array_equal(S, z["S"])
        assert np.array_equal(M, z["M"]) and np.array_equal(G, z["G"])


def test_q_written_in_place_on_three_ranks():
    """SPMD: every rank makes the same element-wise writes into ``opt.Q`` (the
    reference mutates Q in place); the write-back uploads each rank's shard before the
    next pass and the sets / query point equal the unsharded optimiser's."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import safeopt_amd
    from oracle import gp_numpy as gpn
    from _golden import load, make_kernel
    from _oracle_backend import OracleGridBackend, use_oracle_backend
    use_oracle_backend()
    import threading

    class ThreadComm(object):
        def __init__(self, rank, world, shared):
            self.rank, self.world, self.sh = rank, world, shared

        def _exchange(self, a):
            self.sh["slots"][self.rank] = np.array(a, copy=True)
            self.sh["bar"].wait()
            out = [np.array(s, copy=True) for s in self.sh["slots"]]
            self.sh["bar"].wait()
            return out

        def allreduce_max(self, a):
            return np.max(np.stack(self._exchange(np.asarray(a, dtype=float))), axis=0)

        def allgather(self, a):
            return np.stack(self._exchange(np.asarray(a)))

        def barrier(self):
            self.sh["bar"].wait()

    z, meta = load("sets_1d_seed7")

    def scenario(comm):
        gp = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                              noise_var=meta["noise_vars"][0])
        opt = safeopt_amd.SafeOpt(gp, z["parameter_set"], 0., threshold=meta["threshold"],
                                  comm=comm)
        opt.update_confidence_intervals()
        n = z["parameter_set"].shape[0]
        opt.Q[n // 5:n // 2, 0] -= 0.2          # spans shard boundaries
        opt.Q[[1, n - 2], 1] += 0.5
        opt.compute_sets()
        x = opt.get_new_query_point()
        return x, opt.Q.copy(), opt.S.copy(), opt.M.copy(), opt.G.copy()

    ref = scenario(None)
    world = 3
    shared = dict(slots=[None] * world, bar=threading.Barrier(world))
    out = [None] * world
    err = []

    def run(rank):
        try:
            out[rank] = scenario(ThreadComm(rank, world, shared))
        except Exception:                       # noqa: BLE001
            import traceback
            err.append(traceback.format_exc())
            shared["bar"].abort()

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ths]; [t.join(120) for t in ths]
    assert not err, err[0]
    for r in range(world):
        for a, b in zip(out[r], ref):
            assert np.array_equal(a, b)


# ---------------------------------------------------------------------------
# the launcher: `python bench.py --gpus N` from a bare shell
def test_bench_spawns_its_own_ranks():
    """bench.py --gpus 2 with no torchrun environment starts two ranks itself
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), they rendezvous over the
    product's TCP helpers and rank 0 alone prints the JSON line."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2",
                          "--launch-check"], capture_output=True, text=True, env=env,
                         timeout=120, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j == {"launch_check": True, "n_gpus": 2, "local_rank": 0}


def test_file_rendezvous_rejects_foreign_and_stale_files(tmp_path, monkeypatch):
    """Readers of the single-node rendezvous only accept a file that carries
    this launch's tag; rank 0 replaces a left-over file."""
    sys.path.insert(0, REPO)
    from safeopt_amd import dist
    monkeypatch.setenv("SAFEOPT_RDZV_DIR", str(tmp_path))
    os.chmod(str(tmp_path), 0o700)
    monkeypatch.setenv("SAFEOPT_RDZV_NONCE", "launch-A")
    tag_a = dist._launch_tag(4711)
    path = os.path.join(str(tmp_path), "rdzv_%d_%d_%s.bin" % (os.getppid(), 4711, tag_a.hex()[:8]))
    monkeypatch.setenv("SAFEOPT_RDZV_NONCE", "launch-B")
    assert dist._launch_tag(4711) != tag_a
    # a file of another launch under the name a reader of launch A polls
    with open(path, "wb") as f:
        f.write(b"x" * 128 + dist._launch_tag(4711))
    monkeypatch.setenv("SAFEOPT_RDZV_NONCE", "launch-A")
    with pytest.raises(RuntimeError, match="did not appear"):
        dist._file_rendezvous(1, 2, 4711, timeout=0.3)
    # the right tag is accepted
    with open(path, "wb") as f:
        f.write(b"u" * 128 + tag_a)
    uid, _ = dist._file_rendezvous(1, 2, 4711, timeout=2.0)
    assert uid == b"u" * 128
    # truncated connection on the TCP path is an error, not a spin
    import socket, threading
    srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(1)
    port = srv.getsockname()[1]
    t = threading.Thread(target=lambda: srv.accept()[0].close()); t.start()
    with pytest.raises(RuntimeError, match="closed early"):
        dist._fetch_uid("127.0.0.1", port, timeout=5.0)
    t.join(); srv.close()


def test_file_rendezvous_three_processes(tmp_path):
    """The single-node exchange of the ncclUniqueId for real: three worker
    processes of one launcher (this test), the readers polling BEFORE rank 0
    publishes.  (No device: the unique id is a stand-in token.)"""
    code = r"""
import os, sys, time
sys.path.insert(0, %r)
from safeopt_amd import dist, _hip
rank = int(sys.argv[1])
_hip.Context.comm_unique_id = staticmethod(lambda: bytes(range(128)))
if rank == 0:
    time.sleep(1.0)
uid, path = dist._file_rendezvous(rank, 3, 29871, timeout=30.0)
print(uid.hex())
""" % REPO
    env = dict(os.environ, SAFEOPT_RDZV_DIR=str(tmp_path), SAFEOPT_RDZV_NONCE="t3")
    os.chmod(str(tmp_path), 0o700)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (1, 2, 0)]
    outs = [p.communicate(timeout=60) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err
        assert out.strip() == bytes(range(128)).hex()


def test_two_ranks_on_one_gpu_fail_loudly(monkeypatch):
    """LOCAL_RANK beyond the visible devices used to wrap around (two ranks on one
    GPU: RCCL then fails or hangs in ncclCommInitRank); now the context refuses."""
    from safeopt_amd import _hip
    monkeypatch.setattr(_hip, "device_count", lambda: 1)
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("SAFEOPT_HIP_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    with pytest.raises(_hip.HipError, match="one process per GPU"):
        _hip.Context.default()


def _transport_worker(rank, world, port, cmd, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), SAFEOPT_BENCH_PROBE_TIMEOUT="2",
                          SAFEOPT_BENCH_PROBE_CMD="\x1f".join(cmd))
        for k in ("SAFEOPT_COMM", "SAFEOPT_RCCL_IN_STREAM"):
            os.environ.pop(k, None)
        import bench
        rep = bench.choose_transport(rank, world)
        q.put((rank, rep, os.environ.get("SAFEOPT_COMM"), None))
    except Exception:
        import traceback
        q.put((rank, None, None, traceback.format_exc()))


@pytest.mark.timeout(120)
@pytest.mark.parametrize("probe", ["hangs", "fails on one rank", "passes"])
def test_bench_transport_chain_under_the_watchdog(probe):
    """bench.py --gpus N decides its transport under a watchdog (RCCL in stream -> RCCL with
    host-side collectives -> TCP): a probe that HANGS (ncclCommInitRank, an in-stream
    collective) or fails on any rank must take every rank to the next variant, and the line
    says which one ran."""
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    port = _free_port()
    q = mpc.Queue()
    cmd = {"hangs": [sys.executable, "-c", "import time; time.sleep(600)"],
           "fails on one rank": [sys.executable, "-c",
                                 "import os, sys; sys.exit(int(os.environ['RANK'] == '1'))"],
           "passes": [sys.executable, "-c", "pass"]}[probe]
    procs = [mpc.Process(target=_transport_worker, args=(r, 2, port, cmd, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    for rank, rep, comm_env, err in res:
        assert err is None, err
        if probe == "passes":
            assert rep["chosen"] == "rccl-in-stream" and comm_env is None
            assert [c["ok"] for c in rep["chain"]] == [True]
        else:
            assert rep["chosen"] == "socket (fallback)" and comm_env == "socket"
            assert [c["transport"] for c in rep["chain"]] == ["rccl-in-stream", "rccl-host"]
            assert not any(c["ok"] for c in rep["chain"])
            if probe == "hangs":
                assert all(c["rc_this_rank"] == -9 for c in rep["chain"])


@pytest.mark.timeout(120)
def test_bench_refuses_a_fallback_when_rccl_is_required():
    """``SAFEOPT_REQUIRE_RCCL=1 bench.py --gpus N``: when RCCL with the step in stream does not
    come up on every rank, the run ends NON-ZERO with an error line on rank 0 -- no number
    measured over TCP under the name of the product path."""
    port = _free_port()
    bad = "\x1f".join([sys.executable, "-c", "import sys; sys.exit(1)"])
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SAFEOPT_REQUIRE_RCCL="1",
                   SAFEOPT_BENCH_PROBE_TIMEOUT="5", SAFEOPT_BENCH_PROBE_CMD=bad)
        for k in ("SAFEOPT_COMM", "SAFEOPT_RCCL_IN_STREAM"):
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=100) for p in procs]
    assert [p.returncode for p in procs] == [3, 3], [o[1][-400:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["value"] is None and "SAFEOPT_REQUIRE_RCCL" in line["error"]
    assert line["transport"]["chosen"] == "socket (fallback)"
    assert [c["ok"] for c in line["transport"]["chain"]] == [False, False]
    assert outs[1][0].strip() == ""
