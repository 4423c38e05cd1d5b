"""SURVEY.md section 8(e) on the device: observations sharded over ranks, one in-place all-reduce per sum over
observations (`lrvb_set_reduce_hook`), everything else replicated.

A one-GPU box cannot run RCCL with two ranks (it refuses two ranks on one device), so the multi-rank runs here put
TWO PROCESSES on GPU 0 and exchange over gloo: every line of the sharded code path runs -- the hook inside
value / gradient / HVP / device CG / blocked CG / device trust-ncg / Hessian build, `ShardedHessian`, `bench.py`'s
self-launch -- only the transport differs from the 8-GPU run (RCCL over xGMI), which the driver performs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import models as om, packing as opk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    rng = np.random.default_rng(404)
    N, P = 3001, 160                      # odd N: uneven shards; P spans two 128-column tiles
    x = rng.normal(size=(N, P)) / np.sqrt(P)
    beta = rng.normal(size=P)
    y = (rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-x @ beta))).astype(np.float64)
    w = rng.uniform(0.5, 1.5, size=N)
    theta = rng.normal(size=P) * 0.2
    v = rng.normal(size=P)
    B = rng.normal(size=(3, P))
    return N, P, x, y, w, theta, v, B


def _layout(P):
    return opk.Layout([opk.box_block(100), opk.box_block(P - 100, lb=0.0)])


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import torch
import torch.distributed as dist
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[3]
dist.init_process_group('gloo', rank=rank, world_size=world)
import lrvb_amd as vb
from lrvb_amd.distributed import ShardedObjective, ShardedHessian, DeviceEngine, shard_rows
from test_gpu_sharded import _problem
N, P, x, y, w, theta, v, B = _problem()
r0, r1 = shard_rows(N, rank, world)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
par = vb.ModelParamsDict('par')
par.push_param(vb.VectorParam('u', 100)); par.push_param(vb.VectorParam('pos', P - 100, lb=0.0))
fun = vb.DeviceObjective(par, x=x[r0:r1], y=y[r0:r1], loss='logistic', quad_A=np.full(P, 0.9), weights=w[r0:r1])
fun._push_state()
ctx = fun.ctx
so = ShardedObjective(ctx, dev)
assert so.on_device and so.world == world
val, g, hv = so.value(theta), so.grad(theta), so.hvp(theta, v)
sol, info = so.cg_solve(theta, B[0], tol=1e-10)                     # device loop, one all-reduce per iteration
Xm, infos, its = ctx.cg_solve_multi(theta, B, tol=1e-10)           # blocked CG: one block all-reduce per iteration
fit = so.minimize_trust_ncg(theta, gtol=1e-7, maxiter=100, on_device=True)
fit_host = so.minimize_trust_ncg(theta, gtol=1e-7, maxiter=100)    # scipy drives; products reduced in the library
# fit first, THEN the Hessian on the same context, three ways: the hook inside lrvb_hessian, the explicit
# partial -> all-reduce -> finish of ShardedHessian with the hook removed, and G^T G
H_hook = ctx.hessian(theta)
G_hook = ctx.gram(theta)
so.close()
th = torch.from_numpy(theta).to(dev)
eng = DeviceEngine(ctx, dev)
H_sh = ShardedHessian(eng).build(th).cpu().numpy()
flat = np.concatenate([[val], g, hv, sol, [float(info)], Xm.ravel(), infos.astype(float), fit.x,
                       [float(fit.nit), float(fit.status)], fit_host.x, [float(fit_host.nit)],
                       H_hook.ravel(), H_sh.ravel(), G_hook.ravel()])
t = torch.from_numpy(flat.copy())
gathered = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
if rank == 0:
    assert all(torch.equal(gathered[0], q) for q in gathered), 'ranks disagree'
    np.save(sys.argv[4], flat)
dist.destroy_process_group()
'''


def test_two_ranks_on_one_gpu_every_sharded_route(tmp_path):
    world = 2
    port = _free_port()
    out_path = str(tmp_path / 'sharded.npy')
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), out_path], env=env)
             for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world
    flat = np.load(out_path)

    N, P, x, y, w, theta, v, B = _problem()
    full = om.DeclaredModel(_layout(P), loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.9))
    H = full.hessian(theta)
    o = 0
    assert abs(flat[o] - full.value(theta)) < 1e-12 * abs(full.value(theta)); o += 1
    np.testing.assert_allclose(flat[o:o + P], full.grad(theta), rtol=1e-11, atol=1e-12); o += P
    np.testing.assert_allclose(flat[o:o + P], H @ v, rtol=1e-11, atol=1e-12); o += P
    np.testing.assert_allclose(flat[o:o + P], np.linalg.solve(H, B[0]), rtol=1e-7, atol=1e-9); o += P
    assert flat[o] == 0.0; o += 1
    np.testing.assert_allclose(flat[o:o + 3 * P].reshape(3, P), np.linalg.solve(H, B.T).T, rtol=1e-7, atol=1e-9); o += 3 * P
    assert np.all(flat[o:o + 3] == 0.0); o += 3
    x_dev = flat[o:o + P]; o += P
    nit_dev, status_dev = flat[o], flat[o + 1]; o += 2
    x_host = flat[o:o + P]; o += P
    nit_host = flat[o]; o += 1
    assert status_dev == 0 and np.linalg.norm(full.grad(x_dev)) < 1e-6 and np.linalg.norm(full.grad(x_host)) < 1e-6
    # same iterates on both routes; compared in constrained coordinates (a lower-bounded coefficient near its bound
    # is flat in its free coordinate: exp(-20))
    lay = _layout(P)
    assert nit_dev == nit_host and np.max(np.abs(lay.constrain(x_dev) - lay.constrain(x_host))) < 1e-8
    for name in ('hook', 'sharded'):
        Hs = flat[o:o + P * P].reshape(P, P); o += P * P
        assert np.max(np.abs(Hs - H)) < 1e-11 * np.max(np.abs(H)), name
    G = flat[o:o + P * P].reshape(P, P); o += P * P
    want = full.gram(theta)
    assert np.max(np.abs(G - want)) < 1e-11 * np.max(np.abs(want))
    assert o == flat.size


def test_hook_over_a_one_rank_group_and_hook_errors(tmp_path):
    """In-process: the hook path with a one-rank gloo group reproduces the unsharded answers, an exception raised by a
    hook surfaces from the library call that triggered it, and removing the hook restores the plain context."""
    import torch
    import torch.distributed as dist
    import lrvb_amd as vb
    from lrvb_amd.distributed import ShardedObjective
    N, P, x, y, w, theta, v, B = _problem()
    par = vb.ModelParamsDict('par')
    par.push_param(vb.VectorParam('u', 100)); par.push_param(vb.VectorParam('pos', P - 100, lb=0.0))
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 0.9), weights=w)
    fun._push_state()
    full = om.DeclaredModel(_layout(P), loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.9))
    started = False
    if not dist.is_initialized():
        os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(_free_port())
        dist.init_process_group('gloo', rank=0, world_size=1)
        started = True
    try:
        calls = []
        so = ShardedObjective(fun.ctx, torch.device('cuda', 0))
        inner = fun.ctx._hook_cb
        assert so.on_device and inner is not None
        assert abs(so.value(theta) - full.value(theta)) < 1e-12 * abs(full.value(theta))
        np.testing.assert_allclose(so.hvp(theta, v), full.hessian(theta) @ v, rtol=1e-11, atol=1e-12)
        H = fun.ctx.hessian(theta)
        assert np.max(np.abs(H - full.hessian(theta))) < 1e-11 * np.max(np.abs(H))
        # a counting hook: one reduction per value / gradient call, one per product, ONE per Hessian build
        fun.ctx.set_reduce_hook(lambda ptr, n, stream: calls.append(n))
        fun.ctx.value(theta); fun.ctx.hvp(theta + 1e-3, v); fun.ctx.hessian(theta)
        assert calls == [1 + P, 1 + P, P, fun.ctx.stats_size()]
        # a failing hook fails the call, with the original exception
        def boom(ptr, n, stream):
            raise KeyError('exchange lost')
        fun.ctx.set_reduce_hook(boom)
        with pytest.raises(KeyError):
            fun.ctx.grad(theta)
        so.close()
        assert fun.ctx._hook_cb is None
        np.testing.assert_allclose(fun.ctx.grad(theta), full.grad(theta), rtol=1e-11, atol=1e-12)
    finally:
        if started:
            dist.destroy_process_group()


def test_in_library_rccl_communicator_one_rank():
    """lrvb_comm_unique_id / lrvb_comm_init / lrvb_allreduce_hessian / lrvb_comm_destroy: the RCCL communicator inside the
    library (librccl dlopen'ed, no torch tensor in the data path).  RCCL refuses two ranks on one device, so this box can
    only run a ONE-rank communicator: every collective really is launched on the context's stream (ncclAllReduce in place,
    identity for one rank), results equal the unsharded oracle, and the error paths hold."""
    import torch
    import lrvb_amd as vb
    N, P, x, y, w, theta, v, B = _problem()
    par = vb.ModelParamsDict('par')
    par.push_param(vb.VectorParam('u', 100)); par.push_param(vb.VectorParam('pos', P - 100, lb=0.0))
    fun = vb.DeviceObjective(par, x=x, y=y, loss='logistic', quad_A=np.full(P, 0.9), weights=w)
    fun._push_state()
    ctx = fun.ctx
    full = om.DeclaredModel(_layout(P), loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.9))
    dev = torch.device('cuda', 0)
    probe = torch.zeros(4, dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        ctx.allreduce_hessian(0, 1)                           # null buffer
    with pytest.raises(RuntimeError):                         # no communicator yet
        ctx.allreduce_hessian(probe.data_ptr(), 4)
    cid = vb.DeviceContext.comm_unique_id()
    assert len(cid) == 128 and any(cid)
    ctx.comm_init(1, 0, cid)
    with pytest.raises(RuntimeError):
        ctx.comm_init(1, 0, cid)                              # one communicator per context
    with pytest.raises(ValueError):
        ctx.comm_init(2, 5, cid)
    H = full.hessian(theta)
    assert abs(ctx.value(theta) - full.value(theta)) < 1e-12 * abs(full.value(theta))
    np.testing.assert_allclose(ctx.hvp(theta, v), H @ v, rtol=1e-11, atol=1e-12)
    assert np.max(np.abs(ctx.hessian(theta) - H)) < 1e-11 * np.max(np.abs(H))
    sol, info, _ = ctx.cg_solve(theta, B[0], tol=1e-10)
    assert info == 0
    np.testing.assert_allclose(sol, np.linalg.solve(H, B[0]), rtol=1e-7, atol=1e-9)
    # the explicit form on a statistics buffer: partial -> lrvb_allreduce_hessian -> finish
    th = torch.tensor(theta, device=dev)
    stats = torch.empty(ctx.stats_size(), dtype=torch.float64, device=dev)
    Hd = torch.empty((P, P), dtype=torch.float64, device=dev)
    # th / stats / Hd live on torch's current stream, the context on its own: explicit event hand-offs both ways
    # (include/lrvb_hip.h "stream ordering", rule 2; round 2's version of this test cloned `stats` with no ordering at
    # all and read the buffer before the tiles had landed)
    ts = torch.cuda.current_stream(dev).cuda_stream
    ctx.wait_stream(ts)
    ctx.hessian_partial_dev(th.data_ptr(), stats.data_ptr())
    ctx.stream_wait(ts)
    before = stats.clone()
    ctx.wait_stream(ts)                                       # the clone must have read `stats` before the collective rewrites it
    ctx.allreduce_hessian(stats.data_ptr(), stats.numel())
    ctx.hessian_finish_dev(th.data_ptr(), stats.data_ptr(), Hd.data_ptr(), P)
    ctx.sync()
    assert torch.equal(stats, before)                         # one rank: the sum over ranks is the buffer itself
    assert np.max(np.abs(Hd.cpu().numpy() - H)) < 1e-11 * np.max(np.abs(H))
    ctx.comm_destroy()
    ctx.comm_destroy()                                        # idempotent
    np.testing.assert_allclose(ctx.grad(theta), full.grad(theta), rtol=1e-11, atol=1e-12)


def _bench_line(extra_args, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra_args, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_and_builds_the_same_matrix():
    """`python bench.py --gpus 2` starts its own two ranks (rehearsal transport on this one-GPU box) and the built
    Hessian does not depend on the world size: the fingerprint equals the one-rank line's."""
    common = ['--steps', '2', '--warmup', '1', '--n-obs', '200000', '--n-free', '256', '--no-cpu-baseline']
    one = _bench_line(['--gpus', '1'] + common, {})
    two = _bench_line(['--gpus', '2'] + common, {'LRVB_BENCH_REHEARSE_ONE_GPU': '1'})
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2
    assert two['config']['ranks_seen'] == 2 and two['config']['backend'] == 'gloo'
    assert two['config']['n_obs_per_gpu'] == 100000 and one['config']['n_obs_per_gpu'] == 200000
    for key, a in one['config']['hessian_fingerprint'].items():
        b = two['config']['hessian_fingerprint'][key]
        assert abs(a - b) <= 1e-11 * max(abs(a), 1.0), key
    assert 'N=200000 obs x D=256' in two['metric'] and two['scaling'] == 'strong'


def test_bench_gpus_2_without_a_second_gpu_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this node has a second GPU')
    env = dict(os.environ)
    env.pop('LRVB_BENCH_REHEARSE_ONE_GPU', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                          '--n-obs', '20000', '--n-free', '128', '--no-cpu-baseline'], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode != 0 and out.stdout.strip() == ''
    assert 'needs GPU 1' in out.stderr


# ---- the statistics entry points of the other model families (configs 2-5): reduced ON THE DEVICE, once per call ----------
def _stat_problems():
    """Small seeded instances of every model family whose O(N) work is a statistics call."""
    from test_mixture_host_math import near_optimum_problem
    rng = np.random.default_rng(77)
    N, p, G = 3001, 5, 40                                    # q = 6: the fused one-pass kernel; uneven shards
    x = rng.normal(size=(N, p)); gid = rng.integers(0, G, size=N).astype(np.int32); gid[gid == 7] = 8      # group 7 is empty
    y = x @ rng.normal(size=p) + rng.normal(size=G)[gid] * 0.7 + rng.normal(size=N) * 0.5
    w = rng.uniform(0.5, 1.5, N)
    d = 3
    yy = rng.normal(size=(N, d))
    xm, wm, theta_m = near_optimum_problem(240, 5, 4, seed=28)
    P = 6
    xl = rng.normal(size=(N, P)) / np.sqrt(P)
    yl = (rng.uniform(size=N) < 0.5).astype(np.float64)
    return dict(N=N, p=p, G=G, x=x, y=y, gid=gid, w=w, d=d, yy=yy, xm=xm, wm=wm, theta_m=theta_m, P=P, xl=xl, yl=yl)


def _stat_objects(vb, pr, r0, r1, m0, m1):
    """The five objects over the row range [r0, r1) (mixture: [m0, m1)) and the points they are evaluated at."""
    from test_lmm_host_math import make_par as lmm_par, random_eta
    from test_gpu_lmm import _layout as lmm_layout
    rng = np.random.default_rng(5)
    p, G, d, P = pr['p'], pr['G'], pr['d'], pr['P']
    lmm = vb.LMMObjective(lmm_par(p, G), pr['x'][r0:r1], pr['y'][r0:r1], pr['gid'][r0:r1], G, weights=pr['w'][r0:r1])
    th_lmm = lmm_layout(p, G).unconstrain(random_eta(rng, p, G))
    par2 = vb.ModelParamsDict('p'); par2.push_param(vb.MVNParam('beta', dim=p)); par2.push_param(vb.GammaParam('tau'))
    reg = vb.MVNRegressionObjective(par2, pr['x'][r0:r1], pr['y'][r0:r1], prior_mean=np.zeros(p), prior_info=np.eye(p), prior_shape=2.0,
                                    prior_rate=2.0, weights=pr['w'][r0:r1])
    th_reg = par2.get_free() + 0.1 * rng.normal(size=par2.free_size())
    par5 = vb.ModelParamsDict('p'); par5.push_param(vb.MVNParam('mu', dim=d)); par5.push_param(vb.WishartParam('lambda', size=d))
    wish = vb.WishartMVNObjective(par5, pr['yy'][r0:r1])
    par5['lambda']['df'].set(d + 4.0)
    th_w = par5.get_free() + 0.05 * rng.normal(size=par5.free_size())
    K, V = 4, 5
    par3 = vb.ModelParamsDict('params')
    par3.push_param(vb.DirichletParamArray('pi', shape=(K,))); par3.push_param(vb.DirichletParamArray('phi', shape=(V, K)))
    par3.push_param(vb.SimplexParam('z', shape=(m1 - m0, K)))
    mix = vb.MixtureObjective(par3, pr['xm'][m0:m1], pi_prior=1.5, phi_prior=0.8, weights=pr['wm'][m0:m1])
    ng = mix.n_global
    th_mix = np.concatenate([pr['theta_m'][:ng], pr['theta_m'][ng:].reshape(240, K - 1)[m0:m1].ravel()])
    parl = vb.ModelParamsDict('p'); parl.push_param(vb.UVNParamVector('beta', length=P))
    lgt = vb.LogitNormalRegressionObjective(parl, pr['xl'][r0:r1], pr['yl'][r0:r1], weights=pr['w'][r0:r1])
    eta_l = np.concatenate([rng.normal(size=P) * 0.3, rng.uniform(0.5, 2.0, P)])
    return (lmm, th_lmm), (reg, th_reg), (wish, th_w), (mix, th_mix), (lgt, eta_l)


def _stat_results(objs):
    (lmm, th_lmm), (reg, th_reg), (wish, th_w), (mix, th_mix), (lgt, eta_l) = objs
    S, gs = lmm.ctx.grouped_stats(want_S=True, want_gs=True)
    out = [S.ravel(), gs.ravel(), lmm.ctx.group_sums().ravel(), lmm.global_hessian(th_lmm).ravel()]
    Sw, W = reg.ctx.weighted_gram(with_sum=True)
    out += [Sw.ravel(), [W], reg.hessian(th_reg, True).ravel(), [reg.value(th_reg, True)]]
    out += [wish.gram(th_w).ravel(), wish.hessian(th_w, True).ravel()]
    out += [mix.global_hessian(th_mix).ravel(), [mix.value(th_mix)]]
    val, g, Hb = lgt.ctx.logitnormal_terms(eta_l[:lgt.P], 1.0 / eta_l[lgt.P:], lgt.gh_x, lgt.gh_w)
    out += [[val], g, Hb[0].ravel(), Hb[1].ravel(), Hb[2].ravel(), lgt.hessian(eta_l, False).ravel()]
    return [np.asarray(o, dtype=np.float64).ravel() for o in out]


_STAT_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests')); sys.path.insert(0, os.path.join({root!r}, 'tools'))
import torch
import torch.distributed as dist
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[3]
dist.init_process_group('gloo', rank=rank, world_size=world)
import lrvb_amd as vb
from lrvb_amd.distributed import shard_rows, torch_reduce_hook
from test_gpu_sharded import _stat_problems, _stat_objects, _stat_results
pr = _stat_problems()
r0, r1 = shard_rows(pr['N'], rank, world)
m0, m1 = shard_rows(240, rank, world)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
objs = _stat_objects(vb, pr, r0, r1, m0, m1)
counts = []
for f, _ in objs:                 # observations are sharded: every statistics call of these contexts returns the sum over ranks
    f.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    inner = torch_reduce_hook(dev)
    f.ctx.set_reduce_hook(lambda ptr, n, stream, inner=inner: (counts.append(n), inner(ptr, n, stream)))
res = _stat_results(objs)
flat = np.concatenate(res)
t = torch.from_numpy(flat.copy())
gathered = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
if rank == 0:
    assert all(torch.equal(gathered[0], q) for q in gathered), 'ranks disagree'
    np.savez(sys.argv[4], flat=flat, sizes=np.array([r.size for r in res]), counts=np.array(counts))
dist.destroy_process_group()
'''


def test_two_ranks_statistics_calls_of_every_model_family(tmp_path):
    """lrvb_grouped_stats / lrvb_group_sums / lrvb_lmm_group_terms (config 4), lrvb_weighted_gram_sum (config 2),
    lrvb_quadform_gram (config 5), lrvb_mixture_stats + the device Schur assembly (config 3) and lrvb_logitnormal_terms with
    the observations sharded over two processes on GPU 0: every call reduces its device buffer through the hook ONCE, the
    ranks agree bitwise, and the results equal those of one unsharded context over all rows (which the other GPU tests pin
    to the oracles)."""
    import lrvb_amd as vb
    world = 2
    port = _free_port()
    out_path = str(tmp_path / 'stats.npz')
    script = tmp_path / 'stat_worker.py'
    script.write_text(_STAT_WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), out_path], env=env) for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world
    rec = np.load(out_path)
    pr = _stat_problems()
    want = _stat_results(_stat_objects(vb, pr, 0, pr['N'], 0, 240))
    assert [w.size for w in want] == list(rec['sizes'])
    o = 0
    for i, w in enumerate(want):
        got = rec['flat'][o:o + w.size]; o += w.size
        scale = max(np.max(np.abs(w)), 1e-300)
        assert np.max(np.abs(got - w)) < 1e-10 * scale, (i, np.max(np.abs(got - w)) / scale)
    # ONE reduction per statistics call, of exactly the buffer the header documents
    q, G, p = pr['p'] + 1, pr['G'], pr['p']
    d = pr['d']; qw = d + 1
    P = pr['P']
    nbk = (qw * (qw + 1) // 2 + 127) // 128                  # tile rows of the Kronecker SYRK over the packed triangle of z z^T
    kron = nbk * (nbk + 1) // 2 * 128 * 128 + qw * qw + 1
    mixn = 4100 + (21 + 21 % 2) * (10 + 10 % 2)             # [S64 | val2 | bad | pad | packed R]: q = 6 -> 21, K = 4 -> 10 packed columns
    assert list(rec['counts']) == [q * q + G * (q + 1), G * (q + 1), q * q + G * (q + 1),      # grouped_stats, group_sums, global_hessian
                                   q * q + 1, q * q + 1, q * q + 1, kron, qw * qw + 1,          # weighted_gram_sum, the one-call Hessian's own statistics (round 4: formed
                                                                                                # inside the call, not cached on the host), value's (cached after), gram, Wishart stats
                                   mixn, 4100, 3 * P * P + 2 * P + 1, 3 * P * P + 2 * P + 1]


def test_statistics_calls_reduce_once_and_fail_together():
    """In process, a counting hook over a one-rank group: each statistics entry point hands exactly one buffer to the hook;
    an indefinite simplex block is reported after the reduction, by the reduced count."""
    import lrvb_amd as vb
    pr = _stat_problems()
    objs = _stat_objects(vb, pr, 0, pr['N'], 0, 240)
    (lmm, th_lmm), (reg, th_reg), (wish, th_w), (mix, th_mix), (lgt, eta_l) = objs
    calls = []
    for f, _ in objs:
        f.ctx.set_reduce_hook(lambda ptr, n, stream: calls.append(n))
        assert f.ctx.has_reduce_hook
    q, G = pr['p'] + 1, pr['G']
    lmm.ctx.grouped_stats(); assert calls == [q * q + G * (q + 1)]; calls.clear()
    lmm.ctx.weighted_gram(); lmm.ctx.weighted_gram(with_sum=True); assert calls == [q * q, q * q + 1]; calls.clear()
    mix.ctx.mixture_stats(4, th_mix[mix.n_global:], mix._lam(np.exp(th_mix[:mix.n_global]))[2], want_schur=False)
    assert calls == [4100]; calls.clear()
    bad = th_mix.copy(); bad[mix.n_global:] = -bad[mix.n_global:] * 3.0 + 4.0
    with pytest.raises(np.linalg.LinAlgError):
        mix.global_hessian(bad)
    assert len(calls) == 1                                   # the reduction ran before the verdict: no rank stops early
    for f, _ in objs:
        f.ctx.set_reduce_hook(None)
        assert not f.ctx.has_reduce_hook


_WORKER_CG = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import torch
import torch.distributed as dist
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[3]
dist.init_process_group('gloo', rank=rank, world_size=world)
import lrvb_amd as vb
from lrvb_amd.distributed import torch_reduce_hook, shard_rows
rng = np.random.default_rng(5)
N, P, Q = 67, 8, 5                                  # tiny: a product takes about as long as the status copy
x = rng.normal(size=(N, P)); y = rng.normal(size=N); theta = rng.normal(size=P) * 0.1
r0, r1 = shard_rows(N, rank, world)
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
par = vb.ModelParamsDict('par'); par.push_param(vb.VectorParam('b', P))
fun = vb.DeviceObjective(par, x=x[r0:r1], y=y[r0:r1], loss='gaussian', quad_A=np.full(P, 0.5))
fun._push_state()
ctx = fun.ctx
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
inner = torch_reduce_hook(dev, None)
calls = [0]
def counting(buf, n, stream):
    calls[0] += 1
    inner(buf, n, stream)
ctx.set_reduce_hook(counting)
sols, counts = [], []
for rep in range(40):
    B = rng.normal(size=(Q, P)) * np.array([1.0, 1e-3, 10.0, 0.0, 1.0])[:, None]      # systems stop at different iterations; one is zero
    before = calls[0]
    X, info, its = ctx.cg_solve_multi(theta, B, tol=10.0 ** -(4 + rep % 7))
    counts.append(calls[0] - before)
    sols.append(np.concatenate([X.ravel(), info.astype(float), its.astype(float)]))
ctx.set_reduce_hook(None)
t = torch.from_numpy(np.concatenate([np.array(counts, dtype=float)] + sols))
gathered = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
if rank == 0:
    assert all(torch.equal(gathered[0], q) for q in gathered), 'ranks disagree (reduction counts or solutions)'
    np.save(sys.argv[4], gathered[0].numpy())
dist.destroy_process_group()
'''


def test_blocked_cg_queues_the_same_reductions_on_every_rank(tmp_path):
    """Advisor finding, round 3: the host of the fused blocked CG runs one iteration ahead of its convergence test and used
    to read the LIVE flags, which the next iteration's head kernel rewrites in place -- ranks could then stop at different
    iterations and queue different numbers of all-reduces.  The stop decision now reads a per-iteration snapshot.  Two ranks,
    a shape so small that a product takes about as long as the status copy, 40 solves with systems that stop at different
    iterations: the number of hook calls per solve and the solutions are identical on both ranks, and the solutions are right."""
    world = 2
    port = _free_port()
    out_path = str(tmp_path / 'cg.npy')
    script = tmp_path / 'worker_cg.py'
    script.write_text(_WORKER_CG.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), out_path], env=env) for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world
    flat = np.load(out_path)
    rng = np.random.default_rng(5)
    N, P, Q = 67, 8, 5
    x = rng.normal(size=(N, P)); y = rng.normal(size=N); theta = rng.normal(size=P) * 0.1
    H = x.T @ x + 0.5 * np.eye(P)
    counts, rest = flat[:40], flat[40:].reshape(40, Q * P + 2 * Q)
    assert np.all(counts >= 2) and np.all(counts <= P + 3)              # [value | gradient] state + one block product per iteration
    for rep in range(40):
        B = rng.normal(size=(Q, P)) * np.array([1.0, 1e-3, 10.0, 0.0, 1.0])[:, None]
        X = rest[rep, :Q * P].reshape(Q, P)
        tol = 10.0 ** -(4 + rep % 7)
        for q in range(Q):
            assert np.linalg.norm(H @ X[q] - B[q]) <= 10 * tol * np.linalg.norm(B[q]) + 1e-300
        assert np.all(rest[rep, Q * P:Q * P + Q] == 0)
