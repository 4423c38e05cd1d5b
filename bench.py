#!/usr/bin/env python3
"""Headline benchmark: ELBO-Hessian builds/sec (+ LRVB-covariance solve time) at
N = 1e6 observations x D = 1024 free parameters (BASELINE.json `metric`), synthetic data.

    python bench.py --gpus N --steps K --warmup W

One "step" = one Hessian build: (theta, w) resident in HBM -> dense free-coordinate Hessian
(both triangles) in HBM, including the pass over all observations (weights are an input, so
nothing is hoisted): constrain -> fused value/gradient/curvature pass over X -> fp64-MFMA
weighted SYRK X^T diag(w loss'') X -> [N > 1: sum all-reduce of the packed statistics over
RCCL] -> J^T (.) J + third-order + prior assembly.

The total N is FIXED as --gpus grows (observations shard over ranks): "scaling": "strong".
Rank 0 prints ONE JSON line with the contract keys plus `roofline` (dominant kernel = the
weighted SYRK, timed with HIP events on the context's stream inside the timed region) and,
at N = 1, `cpu_baseline` (the numpy oracle timed on this box's host cores on a bounded
sample) and `lrvb_solve_ms`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix peak (vendor figure)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-obs', type=float, default=1e6)
    ap.add_argument('--n-free', type=int, default=1024)
    ap.add_argument('--loss', default='gaussian', choices=['gaussian', 'logistic', 'poisson'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-rows', type=int, default=16384)
    ap.add_argument('--n-splits', type=int, default=0)
    return ap.parse_args()


def cpu_baseline(x_sample, y_sample, n_total, D, n_pos, loss, lik_info, prior_info, theta):
    """Reference-faithful port: what autograd.hessian executes (LRVB/SparseObjectives.py:103) --
    one gradient + D Hessian-vector products, each a full pass over the observations -- timed on a
    row sample with numpy (BLAS threads = all host cores) and extrapolated linearly in N.  The
    strong numpy path (one dsyrk-like X^T diag(c) X) is timed beside it."""
    import numpy as np
    from oracle import packing as opk, models as om
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    loss_id = {'gaussian': om.GAUSSIAN, 'logistic': om.LOGISTIC, 'poisson': om.POISSON}[loss]
    layout = opk.Layout([opk.box_block(D - n_pos), opk.box_block(n_pos, lb=0.0)])
    model = om.DeclaredModel(layout, loss=loss_id, x=x_sample, y=y_sample, lik_info=lik_info,
                             quad_A=np.full(D, prior_info))
    ns = x_sample.shape[0]
    # time a prefix of the D columns if the full sweep would exceed ~25 s
    t0 = time.time()
    model.hessian_by_hvps(theta, n_columns=16)
    per_col = (time.time() - t0) / 16
    ncol = int(min(D, max(16, 25.0 / max(per_col, 1e-9))))
    t0 = time.time()
    model.grad(theta)
    model.hessian_by_hvps(theta, n_columns=ncol)
    t_sample = time.time() - t0
    t_full = t_sample * (D / ncol) * (n_total / ns)
    # strong path on the same sample
    t0 = time.time()
    model.hessian(theta)
    t_strong = (time.time() - t0) * (n_total / ns)
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = None
    return {
        'value': 1.0 / t_full, 'unit': 'hessian_builds/s', 'cores': int(threads), 'kind': 'port',
        'cpu_affinity': affinity,
        'sample': '{} of {} rows x {} of {} HVP columns, numpy oracle (gradient + D Hessian-vector '
                  'products = the passes autograd.hessian makes), extrapolated linearly in rows and '
                  'columns'.format(ns, int(n_total), ncol, D),
        'seconds_on_sample': t_sample,
        'strong_numpy_value': 1.0 / t_strong,
        'strong_numpy_note': 'closed-form X^T diag(c) X through BLAS on the same row sample, extrapolated in rows',
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import lrvb_amd as vb
    from lrvb_amd.distributed import ShardedHessian, DeviceEngine, shard_rows

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # LRVB_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (process group, stats all-reduce,
    # finish) even with one rank -- used to rehearse the N > 1 path on a one-GPU box
    force_sharded = os.environ.get('LRVB_BENCH_FORCE_SHARDED', '0') == '1'
    use_dist = world > 1 or force_sharded
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # LRVB_BENCH_REHEARSE_ONE_GPU=1: every rank uses GPU 0 and the exchange goes over gloo -- a
        # correctness rehearsal of the multi-rank path on a one-GPU box (RCCL refuses two ranks on one
        # device); the numbers it prints are not a measurement
        rehearse = os.environ.get('LRVB_BENCH_REHEARSE_ONE_GPU', '0') == '1'
        dist.init_process_group('gloo' if rehearse else 'nccl', rank=rank, world_size=world)
        if rehearse:
            local_rank = 0
    if args.gpus != world and rank == 0:
        print('warning: --gpus {} but WORLD_SIZE {}'.format(args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    N_total, D = int(args.n_obs), int(args.n_free)
    n_pos = D // 4                       # box constraint (lb = 0) on the last quarter
    lik_info, prior_info = 2.0, 1.0
    r0, r1 = shard_rows(N_total, rank, world)
    n_local = r1 - r0

    # synthetic shard, generated on the device in 65,536-row chunks seeded by GLOBAL chunk index,
    # so every world size sees the same N_total rows
    X = torch.empty((n_local, D), dtype=torch.float64, device=dev)
    y = torch.empty((n_local,), dtype=torch.float64, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20241)
    beta_true = torch.randn((D,), dtype=torch.float64, device=dev, generator=gen) / D ** 0.5
    chunk = 65536
    c0 = (r0 // chunk) * chunk
    while c0 < r1:
        gen.manual_seed(20241 + 1 + c0 // chunk)
        rows = min(chunk, N_total - c0)
        xc = torch.randn((rows, D), dtype=torch.float64, device=dev, generator=gen)
        nc = torch.randn((rows,), dtype=torch.float64, device=dev, generator=gen)
        a, b = max(c0, r0), min(c0 + rows, r1)
        if b > a:
            xs = xc[a - c0:b - c0]
            X[a - r0:b - r0] = xs
            z = xs @ beta_true
            if args.loss == 'gaussian':
                y[a - r0:b - r0] = z + nc[a - c0:b - c0] / lik_info ** 0.5
            elif args.loss == 'logistic':
                y[a - r0:b - r0] = (torch.sigmoid(z) > torch.rand_like(z)).double()
            else:
                y[a - r0:b - r0] = torch.poisson(torch.exp(z))
        c0 += chunk
        del xc, nc
    w = torch.ones((n_local,), dtype=torch.float64, device=dev)
    gen.manual_seed(777)
    theta = 0.05 * torch.randn((D,), dtype=torch.float64, device=dev, generator=gen)
    torch.cuda.synchronize()

    blocks = [dict(kind=0, free_size=D - n_pos, vec_size=D - n_pos, dim0=D - n_pos, dim1=0, lb=-np.inf, ub=np.inf),
              dict(kind=0, free_size=n_pos, vec_size=n_pos, dim0=n_pos, dim1=0, lb=0.0, ub=np.inf)]
    ctx = vb.DeviceContext(blocks, loss=args.loss, n_obs=n_local, n_cols=D, lik_info=lik_info,
                           quad_kind=vb._hip.QUAD_DIAG, device=local_rank)
    ctx.set_data_dev(vb._hip.SLOT_X, X.data_ptr(), n_local, D)
    ctx.set_data_dev(vb._hip.SLOT_Y, y.data_ptr(), n_local, 1)
    ctx.set_weights_dev(w.data_ptr(), n_local)
    ctx.set_data(vb._hip.SLOT_QUAD_A, np.full(D, prior_info))
    if args.n_splits:
        ctx.set_tuning(args.n_splits)

    H = torch.empty((D, D), dtype=torch.float64, device=dev)
    engine = DeviceEngine(ctx, dev) if use_dist else None
    sharded = ShardedHessian(engine) if use_dist else None

    def step():
        if not use_dist:
            ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), D)
            return H
        return sharded.build(theta)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Hout = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed

    # dominant kernel: weighted SYRK of this rank's shard (algorithmic flops / HIP-event time)
    ws_ms = prof['wsyrk_ms'] / max(prof['wsyrk_calls'], 1)
    ws_flops = float(n_local) * D * (D + 1)
    achieved = ws_flops / (ws_ms * 1e-3) / 1e12 if ws_ms > 0 else 0.0
    traffic = None
    tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tfile) and world == 1 and N_total == 1000000 and D == 1024:
        try:
            traffic = json.load(open(tfile)).get('wsyrk_hbm_bytes_per_launch')
        except Exception:
            traffic = None
    out = {
        'metric': 'ELBO-Hessian builds/sec, N=1e6 obs x D=1024 free params',
        'value': value, 'unit': 'hessian_builds/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'headline dense-design {} GLM-type ELBO term: N={} observations x D={} free '
                               'parameters (last {} box-constrained lb=0), Gaussian prior; one step = one dense '
                               'Hessian build'.format(args.loss, N_total, D, n_pos),
                   'n_obs_total': N_total, 'n_obs_per_gpu': n_local, 'n_free': D,
                   'parallelism': 'observation shards x{} + 1 sum all-reduce per build'.format(world)},
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / PEAK_FP64_MFMA_TFLOPS, 'traffic': traffic,
                     'kernel': 'wsyrk_glds_kernel (v_mfma_f64_16x16x4_f64)', 'kernel_ms': ws_ms,
                     'flops_per_launch': ws_flops,
                     'pass_kernel_ms': prof['pass_ms'] / max(prof['pass_calls'], 1),
                     'pass_kernel_GBs': (8.0 * n_local * (D + 3)) / (prof['pass_ms'] / max(prof['pass_calls'], 1) * 1e-3) / 1e9
                     if prof['pass_ms'] > 0 else None},
    }

    # the same data for every world size (chunks are seeded by their global index), so the built matrix must not depend on
    # it: a fingerprint of H that can be compared across the N = 1, 2, 4, 8 lines (agreement to ~1e-12 relative)
    torch.cuda.synchronize()
    out['config']['hessian_fingerprint'] = {'trace': float(torch.trace(Hout).item()),
                                           'sum_abs_64x64': float(Hout[:64, :64].abs().sum().item()),
                                           'last_row_sum': float(Hout[D - 1].sum().item())}

    if world == 1:
        # LRVB-covariance solve time: cho_factor(H) + M H^-1 M^T with Q = D moments (worst case)
        M = torch.eye(D, dtype=torch.float64, device=dev)
        cov = torch.empty((D, D), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.chol_factor_dev(Hout.data_ptr(), D, D)
        ctx.lrvb_cov_dev(M.data_ptr(), D, D, cov.data_ptr())
        ctx.sync()
        t0 = time.perf_counter()
        ctx.chol_factor_dev(Hout.data_ptr(), D, D)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.lrvb_cov_dev(M.data_ptr(), D, D, cov.data_ptr())
        ctx.sync()
        t2 = time.perf_counter()
        out['lrvb_solve_ms'] = {'cho_factor': (t1 - t0) * 1e3, 'cov_Q_eq_D': (t2 - t1) * 1e3}
        # conjugate-gradient route (ConjugateGradientSolver, tol 1e-8) for Q = 16 right-hand sides
        theta_h = theta.cpu().numpy()
        rng = np.random.default_rng(5)
        rhs = rng.normal(size=(16, D))
        ctx.cg_solve(theta_h, rhs[0])
        t3 = time.perf_counter()
        iters = []
        for q in range(16):
            _, info, it = ctx.cg_solve(theta_h, rhs[q], tol=1e-8)
            iters.append(it if info == 0 else -1)
        t4 = time.perf_counter()
        out['lrvb_solve_ms']['cg_16_rhs_one_by_one'] = (t4 - t3) * 1e3
        # the same 16 systems in lockstep: one pair of passes over X per iteration for all of them
        ctx.cg_solve_multi(theta_h, rhs[:2], tol=1e-8)
        t5 = time.perf_counter()
        _, infos, its = ctx.cg_solve_multi(theta_h, rhs, tol=1e-8)
        t6 = time.perf_counter()
        out['lrvb_solve_ms']['cg_16_rhs_tol1e-8'] = (t6 - t5) * 1e3
        out['lrvb_solve_ms']['cg_iterations'] = [int(i) if f == 0 else -1 for i, f in zip(its, infos)]
        out['lrvb_solve_ms']['cg_iterations_one_by_one'] = iters
        if not args.no_cpu_baseline and rank == 0:
            ns = min(args.cpu_sample_rows, n_local)
            xs = X[:ns].cpu().numpy()
            ys = y[:ns].cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(xs, ys, N_total, D, n_pos, args.loss, lik_info, prior_info,
                                               theta.cpu().numpy())
            # parity of the timed result on the sample's leading block (cheap sanity, not the test suite)
    if use_dist:
        dist.destroy_process_group()
    return out if rank == 0 else None


def run_with_clean_stdout():
    """The contract is ONE JSON line on stdout.  Libraries underneath write there too (RCCL prints a five-line
    version banner to stdout when the first communicator comes up, on every rank), so file descriptor 1 points at
    stderr while the bench runs and is restored only for the result line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    out = main()
    sys.stdout.flush()
    if out is not None:                      # rank 0; the other ranks keep writing to stderr until they exit
        os.dup2(saved, 1)
        os.close(saved)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    run_with_clean_stdout()
