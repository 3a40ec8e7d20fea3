"""the N-rank control flow on ONE GPU -- pretended worlds, the in-stream step with a one-rank
communicator, the device-side merges, a torchrun launch (real processes on true shards: test_gpu_nrank.py)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal
from _golden import load, make_kernel

from _gpu_common import (  # noqa: F401
    MEAN_TOL, VAR_TOL, mods, smooth, kernels, check_posterior, product_kernel, GOLD, build_opt, _swarm_problem, _grow_reference, kernels_from, _PretendWorld, _PretendWorldPadded, _dev_script)

pytestmark = pytest.mark.gpu


def test_rccl_world1_collectives(mods):
    from safeopt_amd import _hip
    ctx = _hip.Context(0)
    uid = _hip.Context.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 0, 1)
    assert_array_equal(ctx.allreduce_max(np.array([1.5, -2.0])), [1.5, -2.0])
    ctx.barrier()


# ---------------------------------------------------------------------------


def test_multirank_control_flow_on_one_gpu(mods):
    """The N-rank host driver with in-stream RCCL scalars on ONE GPU: rank 0 of
    a pretended world of 2 owns the first half of the grid, so every iteration
    must equal a plain single-GPU SafeOpt on that half."""
    safeopt_amd, gpy, _, _ = mods
    from safeopt_amd import _hip, dist
    from bench import make_config, build_gps, _bumps
    ctx = _hip.Context.default()
    if not getattr(ctx, "_one_rank_comm", False):
        ctx.comm_init(_hip.Context.comm_unique_id(), 0, 1)
        ctx._one_rank_comm = True
    comm = _PretendWorld(dist.RcclComm(ctx), 2)
    cfg = make_config(3, side=90)                 # 3 GPs, Matern-5/2
    half = cfg["grid"][:cfg["grid"].shape[0] // 2]

    def make(grid, comm):
        gps = build_gps(cfg, gpy)
        return safeopt_amd.SafeOpt(gps, grid, cfg["fmin"], threshold=cfg["threshold"], comm=comm)
    a, b = make(cfg["grid"], comm), make(half, None)
    assert a._shard == (0, half.shape[0])
    # the certified step of the N-rank driver is ONE device round trip: first-candidate
    # merge, probe flags and arg-max merge on the device behind in-stream collectives
    calls = {"fused_comm": 0, "host_gathers": 0}
    inner = a._backend.sets_fused_comm

    def counted(*args, **kw):
        calls["fused_comm"] += 1
        return inner(*args, **kw)
    a._backend.sets_fused_comm = counted
    gather = comm.allgather

    def counted_gather(x):
        calls["host_gathers"] += 1
        return gather(x)
    comm.allgather = counted_gather
    for it in range(4):
        before = dict(calls)
        xa, xb = a.optimize(), b.optimize()
        assert calls["fused_comm"] == before["fused_comm"] + 1
        # (host collectives only when the probe does not certify the first candidate
        # or exact ties have to be settled)
        if a._argmax_cache is not None and calls["host_gathers"] != before["host_gathers"]:
            assert calls["host_gathers"] - before["host_gathers"] <= 2
        assert_array_equal(xa, xb)
        n = half.shape[0]
        assert_array_equal(a._backend.download(_hip.Q), b.Q)
        for what, ref in ((_hip.S, b.S), (_hip.M, b.M), (_hip.G, b.G)):
            assert_array_equal(a._backend.download(what)[:n], ref)
        y = np.array([[_bumps(np.atleast_2d(xa), 102 + g)[0] + 1.0 for g in range(3)]])
        a.add_new_data_point(xa, y)
        b.add_new_data_point(xb, y)


def test_fused_comm_step_equals_fused_step(mods):
    """sgp_grid_sets_fused_comm (front half, merges behind the -- here one-rank --
    in-stream collectives, probe, mark, arg-max) returns what sgp_grid_sets_fused
    returns on the same grid, with and without a communicator in the context, and
    leaves the same M / G."""
    safeopt_amd, gpy, _, _ = mods
    from safeopt_amd import _hip
    from bench import make_config, build_gps
    ctx = _hip.Context.default()
    cfg = make_config(3, side=70)
    cfg["X"], cfg["Y"] = cfg["X"][:20], cfg["Y"][:20]      # (wide intervals: expanders exist)
    gps = build_gps(cfg, gpy)
    devs = [g._fitted() for g in gps]
    G = cfg["G"]
    fmin = np.array(cfg["fmin"], dtype=float)
    scaling = np.full(G, 2.0 ** 0.5)
    thr = np.full(G, 0.05)
    grid = _hip.DeviceGrid(ctx, cfg["grid"], G)
    out = {}
    for name in ("fused", "comm"):
        grid.confidence(devs, 2.0, fmin, defer=True)
        if name == "fused":
            r = grid.sets_fused(devs, 2.0, fmin, None, scaling, thr, 0.5)
        else:
            r = grid.sets_fused_comm(devs, 2.0, fmin, scaling, thr, 0.5)
        out[name] = r + (grid.download(_hip.M), grid.download(_hip.G))
    for x, y in zip(out["fused"], out["comm"]):
        assert_array_equal(np.asarray(x), np.asarray(y))
    assert out["comm"][0][4] >= 0          # (a candidate was found: the test is not void)


def test_device_merges_of_the_n_rank_step(mods):
    """k_merge_front / k_merge_argmax on gathered blocks of 1..8 ranks (the harness
    tests/native/merge_check.hip feeds them what the in-stream all-gathers of
    sgp_grid_sets_fused_comm would deliver) against the NumPy merges of
    safeopt_amd/dist.py that the gloo tests pin to unsharded runs: first candidate in
    visiting order with forced width ties across ranks, shards without a candidate,
    total counts, tie counts, staged expander operand, first-index arg-max."""
    import os, struct, subprocess
    from safeopt_amd import dist
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "merge_check")
    if not os.path.exists(exe):
        from safeopt_amd import build as _build      # (hipcc is on the GPU box as well)
        _build.build()
    assert os.path.exists(exe), "python -m safeopt_amd.build builds tests/native/merge_check"
    rng = np.random.default_rng(77)
    for trial in range(40):
        world = int(rng.integers(1, 9))
        d, G = int(rng.integers(1, 9)), int(rng.integers(1, 5))
        nfront = 6 + d + 3 * G
        blocks = np.zeros((world, nfront))
        raw = blocks.view(np.uint8).reshape(world, nfront * 8)
        found = rng.random(world) < (0.0 if trial == 0 else 0.75)
        widths = rng.choice([0.5, 1.25, 1.25, 3.0], size=world)      # ties across ranks
        idx = rng.permutation(10 ** 6)[:world].astype(np.int64)
        ntied = rng.integers(1, 5, size=world).astype(np.int32)
        counts = rng.integers(0, 2 ** 40, size=(world, 2)).astype(np.uint64)
        blocks[:, 0] = 0.875
        blocks[:, 6:] = rng.normal(size=(world, nfront - 6))
        for r in range(world):
            raw[r, 8:24] = counts[r].view(np.uint8)
            blocks[r, 3] = widths[r] if found[r] else -np.inf
            raw[r, 32:40] = np.array([idx[r] if found[r] else -1], dtype=np.int64).view(np.uint8)
            raw[r, 40:48] = np.array([int(found[r]), ntied[r] if found[r] else 0],
                                     dtype=np.int32).view(np.uint8)
        vals = rng.choice([-np.inf, 0.1, 0.7, 0.7], size=world)
        aidx = rng.permutation(10 ** 6)[:world].astype(np.int64)
        aidx[vals == -np.inf] = -1
        pairs = np.zeros((world, 2))
        pairs[:, 0] = vals
        pairs.view(np.int64)[:, 1] = aidx
        out = subprocess.run([exe], input=struct.pack("4i", world, nfront, d, G) +
                             blocks.tobytes() + pairs.tobytes(),
                             capture_output=True, timeout=120)
        assert out.returncode == 0, out.stderr.decode()
        got = np.frombuffer(out.stdout, dtype=np.float64)
        res, xc, resid = got[:nfront], got[nfront:nfront + d], got[nfront + d:nfront + d + G]
        v_got = got[nfront + d + G]
        i_got = int(got[nfront + d + G + 1:].view(np.int64)[0])
        # ---- expectation from the NumPy merges
        w_b, i_b = dist.merge_topk(np.where(found, widths, -np.inf),
                                   np.where(found, idx, -1), 1)
        assert res[0] == 0.875
        assert_array_equal(res[1:3].view(np.uint64), counts.sum(axis=0))
        head = res[5:6].view(np.int32)
        if i_b.size == 0:
            assert head[0] == 0 and head[1] == 0 and res[4:5].view(np.int64)[0] == -1
        else:
            r = int(np.flatnonzero(found & (idx == i_b[0]))[0])
            assert res[3] == w_b[0] and res[4:5].view(np.int64)[0] == i_b[0]
            assert head[0] == 1
            assert head[1] == int(ntied[found & (widths == w_b[0])].sum())
            assert_array_equal(res[6:], blocks[r, 6:])
            assert_array_equal(xc, blocks[r, 6:6 + d])
            assert_array_equal(resid, blocks[r, 6 + d + G + 1::2][:G] - blocks[r, 6 + d:6 + d + G])
        v_e, i_e = dist.merge_argmax(vals, aidx)
        assert i_got == int(i_e)
        if i_e >= 0:
            assert v_got == v_e


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_tied_widths_on_two_pretended_ranks(mods, seed):
    """The forced-tie fixtures through the N-rank driver (sets_front_comm, the tie
    count travelling with each rank's first candidate, _settle_ties over the
    gathered widths): rank 0 of a pretended world of 2 owns the first half of the
    grid and must produce what a single-GPU SafeOpt produces on that half."""
    safeopt_amd, gpy, gpn, son = mods
    from safeopt_amd import _hip, dist
    z, meta = load("ties_1d_seed%d" % seed)
    ctx = _hip.Context.default()
    if not getattr(ctx, "_one_rank_comm", False):
        ctx.comm_init(_hip.Context.comm_unique_id(), 0, 1)
        ctx._one_rank_comm = True
    comm = _PretendWorldPadded(dist.RcclComm(ctx), 2)
    grid = z["parameter_set"]
    n = grid.shape[0] // 2

    def make(g, comm):
        gp = gpy.models.GPRegression(z["X0"], z["Y0"], make_kernel(gpy.kern, meta["kernels"][0]),
                                     noise_var=meta["noise_vars"][0])
        return safeopt_amd.SafeOpt(gp, g, 0., threshold=meta["threshold"], comm=comm)
    a, b = make(grid, comm), make(grid[:n], None)
    assert a._shard == (0, n)
    a.Q = z["Q"]; b.Q = z["Q"][:n]
    a.compute_sets(); b.compute_sets()
    for what, ref in ((_hip.S, b.S), (_hip.M, b.M), (_hip.G, b.G)):
        assert_array_equal(a._backend.download(what)[:n], ref)
    assert_array_equal(a.get_new_query_point(), b.get_new_query_point())
    # ... and the oracle on that half agrees
    go = gpn.GPRegression(z["X0"], z["Y0"], make_kernel(gpn, meta["kernels"][0]),
                          noise_var=meta["noise_vars"][0])
    So, Mo, Go = son.compute_sets([go], grid[:n], z["Q"][:n], meta["fmin"], meta["scaling"],
                                  meta["threshold"], meta["beta"])
    assert_array_equal(b.S, So); assert_array_equal(b.M, Mo); assert_array_equal(b.G, Go)


def test_torchrun_launch_with_rccl(mods, tmp_path):
    """The driver's launch line (torch.distributed.run, one rank) with the RCCL
    communicator forced on: rendezvous file, comm init, collectives, bench JSON."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAFEOPT_FORCE_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(repo, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--side", "200", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=repo)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["roofline"]["achieved"] > 0


# ---------------------------------------------------------------------------
# one-row updates
