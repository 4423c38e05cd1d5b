"""The N > 1 path on CPU: two processes, `gloo` backend, the same ShardedHessian logic the GPU
ranks run (shard rows -> per-rank statistics -> ONE sum all-reduce -> replicated assembly), with
an oracle-backed engine standing in for the device engine.  The result must equal the
single-process Hessian of the full data set."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class OracleEngine(object):
    """Host engine with the device engine's interface and stats layout
    ([value | d f_data / d beta | tile-packed X^T diag(w loss'') X])."""

    def __init__(self, model):
        self.model = model

    def partial(self, theta):
        from lrvb_amd.distributed import pack_tiles
        from oracle import models as om
        m = self.model
        th = theta.numpy()
        eta = m.layout.constrain(th)
        z = m.x @ eta[m.glm_off:m.glm_off + m.P]
        l0, l1, l2 = om.loss_terms(m.loss, m.y, z, m.lik_info)
        stats = np.concatenate([[np.sum(m.w * l0)], m.x.T @ (m.w * l1), pack_tiles(m.x.T @ ((m.w * l2)[:, None] * m.x))])
        return torch.from_numpy(stats)

    def finish(self, theta, stats):
        from lrvb_amd.distributed import unpack_tiles
        from oracle import packing as opk
        m = self.model
        th = theta.numpy()
        st = stats.numpy()
        P, V = m.P, m.layout.V
        eta = m.layout.constrain(th)
        g = np.zeros(V); g[m.glm_off:m.glm_off + P] = st[1:1 + P]
        H = np.zeros((V, V)); H[m.glm_off:m.glm_off + P, m.glm_off:m.glm_off + P] = unpack_tiles(st[1 + P:], P)
        if m.quad_A is not None:                     # N-independent term: added once, after the reduction
            g += m.quad_scale * (m._A_apply(eta - m.quad_m) + m.quad_b)
            H += m.quad_scale * m._A_dense()
        return torch.from_numpy(opk.convert_vector_to_free_hessian(m.layout, th, g, H))


def _make_problem():
    from oracle import packing as opk, models as om
    rng = np.random.default_rng(77)
    N, P = 1003, 150                 # odd N (uneven shards), P spans two 128-column tiles
    x = rng.normal(size=(N, P)) / np.sqrt(P)
    y = rng.poisson(1.0, size=N).astype(np.float64)
    w = rng.uniform(0.5, 1.5, size=N)
    theta = rng.normal(size=P) * 0.2
    def model(rows):
        lay = opk.Layout([opk.box_block(100), opk.box_block(50, lb=0.0)])
        return om.DeclaredModel(lay, loss=om.POISSON, x=x[rows], y=y[rows], w=w[rows], quad_A=np.full(P, 0.7))
    return N, theta, model


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lrvb_amd.distributed import ShardedHessian, shard_rows
    N, theta, model = _make_problem()
    r0, r1 = shard_rows(N, rank, world)
    H = ShardedHessian(OracleEngine(model(slice(r0, r1)))).build(torch.from_numpy(theta))
    # every rank holds the same matrix
    gathered = [torch.empty_like(H) for _ in range(world)]
    dist.all_gather(gathered, H)
    if rank == 0:
        assert all(torch.equal(gathered[0], g) for g in gathered)
        np.save(out_path, H.numpy())
    dist.destroy_process_group()


def test_shard_rows_partition():
    from lrvb_amd.distributed import shard_rows
    for n, w in ((10, 3), (1000000, 8), (7, 8), (1003, 2)):
        cuts = [shard_rows(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def test_tile_packing_roundtrip():
    from lrvb_amd.distributed import pack_tiles, unpack_tiles, stats_layout
    rng = np.random.default_rng(0)
    for P in (5, 128, 150, 300):
        a = rng.normal(size=(P, P)); S = a + a.T
        flat = pack_tiles(S)
        assert flat.size == stats_layout(P)[3] - 1 - P
        np.testing.assert_array_equal(unpack_tiles(flat, P), S)


def test_two_rank_gloo_build_equals_single_process(tmp_path):
    world = 2
    port = _free_port()
    out_path = str(tmp_path / 'H.npy')
    mp.spawn(_worker, args=(world, port, out_path), nprocs=world, join=True)
    H = np.load(out_path)
    N, theta, model = _make_problem()
    want = model(slice(0, N)).hessian(theta)
    assert np.max(np.abs(H - want)) < 1e-11 * np.max(np.abs(want))


def _lmm_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lrvb_amd.distributed import shard_rows, allreduce_stats
    from test_lmm_host_math import make_par, shell, random_eta, host_stats
    rng = np.random.default_rng(11)
    N, p, G = 501, 2, 7
    x = rng.normal(size=(N, p)); y = rng.normal(size=N); w = rng.uniform(0.5, 1.5, N)
    gid = rng.integers(0, G, size=N); gid[:G] = np.arange(G)
    eta = random_eta(rng, p, G)
    r0, r1 = shard_rows(N, rank, world)
    # whole groups are NOT required per rank: per-group statistics are summed too
    local = host_stats(x[r0:r1], y[r0:r1], gid[r0:r1], G, w[r0:r1])
    total = allreduce_stats(local)
    priors = (np.zeros(p), np.eye(p) * 0.7, 0.2, 0.5, (2.0, 1.5), (1.5, 0.8))
    f = shell(make_par(p, G), p, G, priors)
    f._external_stats = None
    f.set_reduced_stats(total)
    g, H = f._dense_vec(eta)
    if rank == 0:
        f.set_reduced_stats(host_stats(x, y, gid, G, w))
        g1, H1 = f._dense_vec(eta)
        np.save(out_path, np.array([np.max(np.abs(H - H1)) / np.max(np.abs(H1)), np.max(np.abs(g - g1)) / np.max(np.abs(g1))]))
    dist.destroy_process_group()


def test_two_rank_sufficient_statistics_allreduce(tmp_path):
    """Config 4's exchange step: per-rank Gram + per-group sums, one sum all-reduce, identical
    arrow Hessian on every rank."""
    out_path = str(tmp_path / 'err.npy')
    mp.spawn(_lmm_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    err = np.load(out_path)
    assert err[0] < 1e-12 and err[1] < 1e-12


def _mixture_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lrvb_amd.distributed import shard_rows, allreduce_stats
    from test_mixture_host_math import make_par, shell, near_optimum_problem, oracle_stats
    N, V, K = 90, 4, 3
    x, w, theta = near_optimum_problem(N, V, K, seed=5)
    ng = K + V * K
    fz = theta[ng:].reshape(N, K - 1)
    r0, r1 = shard_rows(N, rank, world)
    # each rank owns a slice of the observations AND the simplex rows that go with them
    f_loc = shell(make_par(r1 - r0, V, K), x[r0:r1], K, 1.5, 0.8)
    th_loc = np.concatenate([theta[:ng], fz[r0:r1].ravel()])
    local, _, _ = oracle_stats(f_loc, x[r0:r1], w[r0:r1], th_loc)
    total = allreduce_stats(local)
    f_loc.set_reduced_stats(total)
    HS = f_loc.global_hessian(th_loc)
    val = f_loc.value(th_loc)
    if rank == 0:
        f = shell(make_par(N, V, K), x, K, 1.5, 0.8)
        f.set_reduced_stats(oracle_stats(f, x, w, theta)[0])
        HS1 = f.global_hessian(theta)
        np.save(out_path, np.array([np.max(np.abs(HS - HS1)) / np.max(np.abs(HS1)), abs(val - f.value(theta)) / abs(val)]))
    dist.destroy_process_group()


def test_two_rank_mixture_statistics_allreduce(tmp_path):
    """Config 3's exchange step: per-rank [value | S64 | Schur operand R], one sum all-reduce, identical
    Schur complement of the Dirichlet block on every rank."""
    out_path = str(tmp_path / 'err.npy')
    mp.spawn(_mixture_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    err = np.load(out_path)
    assert err[0] < 1e-11 and err[1] < 1e-12


def _mvn_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lrvb_amd.distributed import shard_rows, allreduce_stats
    from test_mvn_regression_host_math import _shell
    rng = np.random.default_rng(4)
    N, k = 333, 4
    x = rng.normal(size=(N, k)); y = rng.normal(size=N); w = rng.uniform(0.5, 1.5, N)
    a = rng.normal(size=(k, k)); lam0 = a @ a.T / k + np.eye(k)
    c = rng.normal(size=(k, k)); lam = c @ c.T + np.eye(k)
    eta = np.concatenate([rng.normal(size=k), lam[np.tril_indices(k)], [3.1, 1.7]])
    f = _shell(k, rng.normal(size=k), lam0, 2.5, 1.3)
    z = np.hstack([x, y[:, None]])
    r0, r1 = shard_rows(N, rank, world)
    local = np.concatenate([(z[r0:r1].T @ (w[r0:r1, None] * z[r0:r1])).ravel(), [w[r0:r1].sum()]])
    f.set_reduced_stats(allreduce_stats(local))
    S, W = f._stats()
    val, g, H = f._terms(eta, S, W)
    if rank == 0:
        v1, g1, H1 = f._terms(eta, z.T @ (w[:, None] * z), float(w.sum()))
        np.save(out_path, np.array([abs(val - v1) / abs(v1), np.max(np.abs(g - g1)) / np.max(np.abs(g1)),
                                    np.max(np.abs(H - H1)) / np.max(np.abs(H1))]))
    dist.destroy_process_group()


def test_two_rank_quadratic_data_statistics_allreduce(tmp_path):
    """Configs 1, 2, 5: S = sum w z z^T and W are the whole exchange."""
    out_path = str(tmp_path / 'err.npy')
    mp.spawn(_mvn_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    assert np.all(np.load(out_path) < 1e-12)


class OracleShardCtx(object):
    """The slice of DeviceContext that ShardedObjective uses, with the oracle's arithmetic (no GPU in this suite)."""

    def __init__(self, model):
        self.model = model
        self.D = model.layout.D

    @property
    def quad_scale(self):
        return self.model.quad_scale

    def set_quad_scale(self, s):
        self.model.quad_scale = s

    def value(self, x):
        return self.model.value(x)

    def grad(self, x):
        return self.model.grad(x)

    def hvp(self, x, v):
        return self.model.hvp(x, v)


def _objective_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lrvb_amd.distributed import ShardedObjective, ShardedHessian, shard_rows
    N, theta, model = _make_problem()
    r0, r1 = shard_rows(N, rank, world)
    local = model(slice(r0, r1))
    obj = ShardedObjective(OracleShardCtx(local))
    rng = np.random.default_rng(5)                    # same stream on every rank
    v, b = rng.normal(size=theta.size), rng.normal(size=theta.size)
    val, g, hv = obj.value(theta), obj.grad(theta), obj.hvp(theta, v)
    sol, info = obj.cg_solve(theta, b, tol=1e-10)
    fit = obj.minimize_trust_ncg(theta, gtol=1e-7, maxiter=100)
    # fit first, THEN the sharded Hessian build on the same shard object: the objective calls must leave nothing
    # rescaled behind (the N-independent term enters the finished Hessian once, at full weight)
    assert local.quad_scale == 1.0
    H = ShardedHessian(OracleEngine(local)).build(torch.from_numpy(theta)).numpy()
    flat = np.concatenate([[val], g, hv, sol, [float(info)], fit.x, [float(fit.nit), float(fit.status)], H.ravel()])
    t = torch.from_numpy(flat.copy())
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    if rank == 0:
        assert all(torch.equal(gathered[0], x) for x in gathered)       # bitwise the same on every rank
        np.save(out_path, flat)
    dist.destroy_process_group()


def test_two_rank_sharded_value_gradient_hvp_cg_and_fit(tmp_path):
    """the HVP / CG path over sharded observations, one D-vector all-reduce per product."""
    import scipy.optimize
    world = 2
    port = _free_port()
    out_path = str(tmp_path / 'obj.npy')
    mp.spawn(_objective_worker, args=(world, port, out_path), nprocs=world, join=True)
    flat = np.load(out_path)
    N, theta, model = _make_problem()
    full = model(slice(0, N))
    D = theta.size
    rng = np.random.default_rng(5)
    v, b = rng.normal(size=D), rng.normal(size=D)
    o = 0
    assert abs(flat[o] - full.value(theta)) < 1e-12 * abs(full.value(theta)); o += 1
    np.testing.assert_allclose(flat[o:o + D], full.grad(theta), rtol=1e-11, atol=1e-12); o += D
    H = full.hessian(theta)
    np.testing.assert_allclose(flat[o:o + D], H @ v, rtol=1e-11, atol=1e-12); o += D
    np.testing.assert_allclose(flat[o:o + D], np.linalg.solve(H, b), rtol=1e-7, atol=1e-9); o += D
    assert flat[o] == 0.0; o += 1
    fit = scipy.optimize.minimize(full.value, theta, jac=full.grad, hessp=lambda x, p: full.hvp(x, p), method='trust-ncg',
                                  options={'maxiter': 100, 'gtol': 1e-7})
    # the lower-bounded coordinates are nearly flat in free coordinates (exp map), so two runs that both meet gtol may
    # sit 1e-4 apart there; the objective value and the stationarity are the sharp statements
    x_sharded = flat[o:o + D]; o += D
    np.testing.assert_allclose(x_sharded, fit.x, rtol=0, atol=1e-3)
    assert abs(full.value(x_sharded) - fit.fun) < 1e-10 * abs(fit.fun)
    assert abs(flat[o] - fit.nit) <= 1 and flat[o + 1] == fit.status == 0
    assert np.linalg.norm(full.grad(x_sharded)) < 1e-6
    o += 2
    H_sharded = flat[o:o + D * D].reshape(D, D)
    assert np.max(np.abs(H_sharded - H)) < 1e-11 * np.max(np.abs(H))


class _StubCommContext(object):
    """Stands where a DeviceContext stands in `native_comm_init`, with the two RCCL calls stubbed at the ctypes boundary:
    `comm_unique_id` returns 128 bytes that differ per process (as ncclGetUniqueId does), `comm_init` records what it got."""
    seen = None

    @staticmethod
    def comm_unique_id():
        return bytes([os.getpid() % 251] * 64 + list(os.urandom(64)))

    def comm_init(self, world, rank, comm_id):
        self.seen = (world, rank, bytes(comm_id))


def _comm_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lrvb_amd.distributed import native_comm_init
    ctx = _StubCommContext()
    got = native_comm_init(ctx)
    own = _StubCommContext.comm_unique_id()                      # what this process WOULD have drawn: never used on rank > 0
    t = torch.tensor(list(ctx.seen[2]) + [ctx.seen[0], ctx.seen[1], got[0], got[1]], dtype=torch.int64)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    if rank == 0:
        np.save(out_path, torch.stack(gathered).numpy())
    assert len(ctx.seen[2]) == 128 and own[:64] != b'' 
    dist.destroy_process_group()


def test_native_communicator_id_is_the_same_on_every_rank(tmp_path):
    """`distributed.native_comm_init` (the id exchange in front of lrvb_comm_init, untestable with RCCL on a one-GPU box):
    rank 0 draws the 128-byte id, the process group carries it, every rank hands THE SAME bytes and its own (world, rank) to
    the library.  Two gloo ranks, the RCCL calls stubbed."""
    out_path = str(tmp_path / 'comm.npy')
    mp.spawn(_comm_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    rows = np.load(out_path)
    assert rows.shape == (2, 132)
    assert np.array_equal(rows[0, :128], rows[1, :128])          # one id
    assert rows[0, 128:].tolist() == [2, 0, 2, 0] and rows[1, 128:].tolist() == [2, 1, 2, 1]
