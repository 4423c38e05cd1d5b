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
