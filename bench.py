#!/usr/bin/env python3
"""Headline benchmark: ELBO-Hessian builds/sec (+ LRVB-covariance solve time) at
N = 1e6 observations x D = 1024 free parameters (BASELINE.json `metric`), synthetic data.

    python bench.py --gpus N --steps K --warmup W

One "step" = one Hessian build: (theta, w) resident in HBM -> dense free-coordinate Hessian
(both triangles) in HBM, including the pass over all observations (weights are an input, so
nothing is hoisted): constrain -> [logistic / Poisson: fused value/gradient/curvature pass over X]
-> fp64-MFMA weighted SYRK X^T diag(w loss'') X (Gaussian loss: its diagonal tiles also form
X^T (c o y), from which the gradient of the data term follows without a separate pass) ->
[N > 1: sum all-reduce of the packed statistics over RCCL] -> J^T (.) J + third-order + prior assembly.

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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n-obs', type=float, default=1e6)
    ap.add_argument('--n-free', type=int, default=1024)
    ap.add_argument('--loss', default='gaussian', choices=['gaussian', 'logistic', 'poisson'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget-s', type=float, default=30.0,
                    help='rough bound on the CPU work of the cpu_baseline leg (seconds)')
    ap.add_argument('--n-splits', type=int, default=0)
    ap.add_argument('--no-configs', action='store_true',
                    help='headline run only: skip the short embedded runs of configurations c2..c5 (the `configs` object)')
    ap.add_argument('--config', default='h', choices=['h', 'c2', 'c3', 'c4', 'c5'],
                    help='h = the headline metric (default, what the driver runs; at N = 1 its line also carries short runs of '
                         'c2..c5 under `configs`); c2..c5 = the other BASELINE.json configurations, one line each with its own '
                         'roofline, observations sharded over --gpus ranks')
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n_ranks, argv, port):
    """The child command that runs this script on n_ranks ranks of ONE node, one rank per GPU (the form the driver
    itself uses for N > 1): torchrun with a loopback rendezvous."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node={}'.format(int(n_ranks)),
            '--master-addr', '127.0.0.1', '--master-port', str(int(port)), os.path.abspath(__file__)] + list(argv)


def needs_launch(args, environ):
    """`python bench.py --gpus N` with N > 1 and no rank environment: this process is not a rank, it starts them."""
    return args.gpus > 1 and 'RANK' not in environ and 'WORLD_SIZE' not in environ


def launch_ranks(args, argv):
    """Start the N ranks as a fresh child process tree and relay rank 0's JSON line.  This parent has not touched
    the GPU (no torch import, no HIP call) and never execs: it waits for the child and exits with its code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = launch_command(args.gpus, argv, _free_port())
    print('bench: starting {} ranks: {}'.format(args.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode('utf-8', 'replace').splitlines() if ln.startswith('{')]
    if proc.returncode != 0 or not lines:
        print('bench: the {}-rank run failed (exit code {}, {} result lines)'.format(args.gpus, proc.returncode, len(lines)),
              file=sys.stderr, flush=True)
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def _blas_threads_scan(np, candidates):
    """Pick the BLAS thread count that runs a 2048^3 dgemm fastest on this host (the box may expose more logical CPUs
    than its share of the machine).  Returns (threads, {threads: GFLOP/s})."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        return None, {}
    a = np.ones((2048, 2048)) * 0.5
    rates = {}
    for t in candidates:
        with threadpool_limits(limits=t):
            a @ a
            t0 = time.perf_counter()
            a @ a
            rates[t] = 2.0 * 2048 ** 3 / (time.perf_counter() - t0) / 1e9
    return max(rates, key=rates.get), rates


def cpu_baseline(fetch_rows, n_total, D, n_pos, loss, lik_info, prior_info, theta, budget_s=30.0):
    """The two CPU baselines of SURVEY.md section 8(d), timed on this host with numpy / BLAS.

    fetch_rows(a, b) -> (x[a:b], y[a:b]) host arrays of the bench's own synthetic rows.

    1. `strong_numpy`: the closed-form build the oracle's `hessian` makes -- curvature pass, X^T diag(c) X through BLAS,
       packing assembly -- MEASURED over all n_total rows in 65,536-row chunks (only the compute of each chunk is
       timed, not fetching it), the N-independent assembly timed once and added.  Its GFLOP/s is printed so the
       figure can be sanity-checked.
    2. `port` (the `value`): the cost structure of `autograd.hessian` (LRVB/SparseObjectives.py:103) -- one gradient,
       then D Hessian-vector products against the standard basis, each a pass over the observations
       (`DeclaredModel.hessian_by_hvps`).  Timed at TWO sample sizes; the per-column time is fitted as a + b N, only
       b N is scaled to n_total, and the N-independent part is added unscaled.  Raw timings are in the record."""
    import numpy as np
    from oracle import packing as opk, models as om
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64, 128) if t <= affinity} | {min(affinity, 256)})
    threads, scan = _blas_threads_scan(np, cands)
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
        limiter = threadpool_limits(limits=threads) if threads else None
        blas = sorted({(p_.get('internal_api'), p_.get('version')) for p_ in threadpool_info()
                       if p_.get('user_api') == 'blas'})
    except Exception:
        limiter, blas = None, []
    if threads is None:
        threads = affinity
    loss_id = {'gaussian': om.GAUSSIAN, 'logistic': om.LOGISTIC, 'poisson': om.POISSON}[loss]
    layout = opk.Layout([opk.box_block(D - n_pos), opk.box_block(n_pos, lb=0.0)])
    n_total = int(n_total)
    raw = {}

    # ---- strong path, full N, chunked ------------------------------------------------------------------
    chunk = 65536
    eta = layout.constrain(theta)
    S = np.zeros((D, D)); g = np.zeros(D)
    t_compute = 0.0
    rows_done = 0
    scaled = None
    budget_strong = 0.45 * budget_s
    # the row scaling diag(c) X is a memory-bound ufunc that numpy runs on one thread: slabs of the chunk go to a small
    # thread pool (ufuncs release the GIL), so that the baseline is not held back by it
    from concurrent.futures import ThreadPoolExecutor
    n_scale = max(1, min(16, threads or 1))
    pool = ThreadPoolExecutor(max_workers=n_scale)

    def scale_rows(xm, cvec, out):
        step = (xm.shape[0] + n_scale - 1) // n_scale
        list(pool.map(lambda i: np.multiply(xm[i:i + step], cvec[i:i + step, None], out=out[i:i + step]),
                      range(0, xm.shape[0], step)))
    for a in range(0, n_total, chunk):
        b = min(a + chunk, n_total)
        xc, yc = fetch_rows(a, b)
        if scaled is None or scaled.shape != xc.shape:
            scaled = np.empty_like(xc)
        t0 = time.perf_counter()
        z = xc @ eta
        _, l1, l2 = om.loss_terms(loss_id, yc, z, lik_info)
        g += xc.T @ l1
        scale_rows(xc, l2, scaled)
        S += xc.T @ scaled
        t_compute += time.perf_counter() - t0
        rows_done = b
        if t_compute > budget_strong:                 # slow host: stop and say so (scaled by rows only)
            break
    pool.shutdown()
    model0 = om.DeclaredModel(layout, loss=0, quad_A=np.full(D, prior_info))
    t0 = time.perf_counter()
    gq = g + model0.grad_vec(eta)
    H_oracle = opk.convert_vector_to_free_hessian(layout, theta, gq, S + model0.hessian_vec(eta))
    t_assembly = time.perf_counter() - t0
    # the checker leg of the parity record: the oracle's free-coordinate Hessian and gradient over ALL rows (kept only when the
    # loop above did see all of them)
    oracle_full = None
    if rows_done == n_total:
        oracle_full = {'hessian': np.asarray(H_oracle.todense() if hasattr(H_oracle, 'todense') else H_oracle),
                       'grad': layout.jac(theta).T @ gq}
    t_strong = t_compute * (n_total / rows_done) + t_assembly
    raw['strong'] = {'rows_timed': rows_done, 'compute_s': t_compute, 'assembly_s': t_assembly,
                     'gflops': 2.0 * rows_done * D * (D + 1) / max(t_compute, 1e-12) / 1e9}

    # ---- port path at two sample sizes -----------------------------------------------------------------
    n_small, n_large = min(10000, n_total), min(100000, n_total)
    per_col = {}
    t_grad = {}
    budget_port = 0.5 * budget_s
    for ns, share in ((n_small, 0.25), (n_large, 0.75)):
        xs, ys = fetch_rows(0, ns)
        model = om.DeclaredModel(layout, loss=loss_id, x=xs, y=ys, lik_info=lik_info, quad_A=np.full(D, prior_info))
        t0 = time.perf_counter()
        model.hessian_by_hvps(theta, n_columns=2)
        est = (time.perf_counter() - t0) / 2
        ncol = int(min(D, max(4, share * budget_port / max(est, 1e-9))))
        t0 = time.perf_counter()
        model.grad(theta)
        t_grad[ns] = time.perf_counter() - t0
        t0 = time.perf_counter()
        model.hessian_by_hvps(theta, n_columns=ncol)
        dt = time.perf_counter() - t0
        per_col[ns] = dt / ncol
        raw['port_n{}'.format(ns)] = {'rows': ns, 'columns_timed': ncol, 'seconds': dt, 'gradient_s': t_grad[ns]}
        if n_small == n_large:
            break
    if n_large > n_small:
        slope = max((per_col[n_large] - per_col[n_small]) / (n_large - n_small), 0.0)
        fixed = max(per_col[n_small] - slope * n_small, 0.0)
        g_slope = max((t_grad[n_large] - t_grad[n_small]) / (n_large - n_small), 0.0)
    else:
        slope, fixed, g_slope = per_col[n_small] / n_small, 0.0, t_grad[n_small] / n_small
    t_port = D * (fixed + slope * n_total) + g_slope * n_total
    raw['port_fit'] = {'per_column_fixed_s': fixed, 'per_column_per_row_s': slope}
    if limiter is not None:
        limiter.restore_original_limits()
    return {
        'value': 1.0 / t_port, 'unit': 'hessian_builds/s', 'cores': int(threads), 'kind': 'port',
        'cpu_affinity': affinity, 'blas': ['{} {}'.format(*b_) for b_ in blas], 'blas_thread_scan_gflops': scan,
        'sample': 'numpy oracle, gradient + D Hessian-vector products (the passes autograd.hessian makes): {} and {} of '
                  '{} rows, a prefix of the {} columns each; per-column time fitted as a + b N, only b N scaled to full '
                  'N'.format(n_small, n_large, n_total, D),
        'seconds_per_build': t_port,
        'strong_numpy_value': 1.0 / t_strong,
        'strong_numpy_seconds_per_build': t_strong,
        'strong_numpy_note': 'closed-form X^T diag(c) X through BLAS (row scaling on a 16-thread pool), measured over {} of {} rows in 65,536-row chunks '
                             '(compute only), N-independent assembly added once'.format(rows_done, n_total),
        'raw_timings': raw,
        '_oracle_full': oracle_full,
    }


# ---- the other BASELINE.json configurations -------------------------------------------------------
# `python bench.py --config cN [--gpus G]`: one JSON line for configuration N, observations sharded over the G ranks (strong
# scaling on the configuration's N; every sum over observations is reduced ON THE DEVICE inside the library's statistics
# call, once per call).  The headline run (`--config h`, what the driver runs) also embeds a short run of c2..c5 at N = 1 in
# its line (`configs`), so that all five rooflines are in the driver's record.
def _timed_steps(step, warmup, steps, ctx, fence):
    """EXACTLY `steps` steps between two fences, timed WITHOUT the library's profile marks (under them a one-call step runs its
    launch chain as plain launches instead of the captured graph); the marks -- HIP-event time of the statistics kernels and of
    the sum-over-ranks hook -- come from a second, shorter run, scaled to `steps` so that the callers' per-step divisions hold."""
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    n_prof = max(1, min(steps, 10))
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(n_prof):
        step()
    fence()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    scale = steps / float(n_prof)
    prof = {k: (v * scale if isinstance(v, float) else int(round(v * scale)) if isinstance(v, int) else v) for k, v in prof.items()}
    return elapsed, prof


def _gather_floats(torch, dist, vals, dev, backend, use_dist):
    """[[vals of rank 0], [vals of rank 1], ...] (one row per rank)."""
    if not use_dist:
        return [list(map(float, vals))]
    t = torch.tensor(list(map(float, vals)), dtype=torch.float64, device=dev if backend != 'gloo' else 'cpu')
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.cpu().tolist()] for o in out]


def _fingerprint(np, H):
    H = np.asarray(H)
    n = min(64, H.shape[0])
    return {'trace': float(np.trace(H)), 'sum_abs_64x64': float(np.abs(H[:n, :n]).sum()), 'last_row_sum': float(H[-1].sum())}


def run_config(args, cfg=None, steps=None, warmup=None, env=None):
    """Configurations 2-5: `value` = builds per second of the configuration's per-step product with the observations,
    the weights and the evaluation point resident in HBM; `roofline` = ALL statistics-kernel launches of the step (HIP-event
    time from the library's profile marks) against the algorithmic bytes / flops of SURVEY.md section 8(d), plus the same
    algorithmic work against the whole step (`step_frac`).  env = (world, rank, dev, backend, use_dist) when called from a
    process that has already set up its rank; None = set it up here."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import lrvb_amd as vb
    from lrvb_amd.distributed import shard_rows, torch_reduce_hook, native_comm_init
    sys.path.insert(0, os.path.join(ROOT, 'tools'))          # synthetic problem generators (no oracle, no test module)
    cfg = cfg or args.config
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    own_group = env is None
    if own_group:
        world, rank, local_rank, dev, backend, use_dist, rehearse = dist_setup(args, torch, dist)
    else:
        world, rank, dev, backend, use_dist = env
    native = use_dist and backend == 'nccl' and os.environ.get('LRVB_BENCH_NATIVE_RCCL', '0') == '1'
    device_index = dev.index or 0
    rng = np.random.default_rng(20240 + int(cfg[1]))          # the same full problem on every rank; each keeps its rows
    n_override = int(args.n_obs) if args.n_obs != 1e6 else None
    extra = {}

    def shard(ctx):
        """Observations are sharded: the context's statistics calls return sums over ALL ranks (one in-place device
        all-reduce per call, RCCL with the nccl backend)."""
        if not use_dist:
            return
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        if native:
            native_comm_init(ctx)
        else:
            ctx.set_reduce_hook(torch_reduce_hook(dev))

    if cfg == 'c2':
        N, k = n_override or 100_000, 21
        x = rng.normal(size=(N, k)); y = x @ rng.normal(size=k) + rng.normal(size=N) / np.sqrt(2.0)
        wts = rng.uniform(0.5, 1.5, N)
        r0, r1 = shard_rows(N, rank, world)
        par = vb.ModelParamsDict('p'); par.push_param(vb.MVNParam('beta', dim=k)); par.push_param(vb.GammaParam('tau'))
        fun = vb.MVNRegressionObjective(par, x[r0:r1], y[r0:r1], prior_mean=np.zeros(k), prior_info=np.eye(k), prior_shape=2.0,
                                        prior_rate=2.0, weights=wts[r0:r1], device=device_index)
        shard(fun.ctx)
        obj = vb.Objective(par, fun)
        theta = par.get_free()
        D = theta.size

        def step():                                           # ONE library call; statistics recomputed from the resident rows and weights
            return fun.device_hessian(theta, want_host=False)    # the result stays in HBM (what chol_factor_last factors)

        def final():
            return obj.fun_free_hessian(theta)
        metric = 'ELBO-Hessian builds/sec, MVNParam regression N={:g} obs x D={} free params'.format(float(N), D)
        workload = ('config 2: MVNParam regression (k=21 -> D={}), N={}; one step = weighted Gram of [x|y] + sum of the weights on the GPU '
                    '+ closed forms evaluated on the device where the statistics lie + Kronecker block + free-Hessian conversion: one library call, '
                    'no copy back inside it, Hessian left in HBM').format(D, N)
        bound, alg, unit, peak = 'hbm', 8.0 * (r1 - r0) * (k + 2), 'GB/s', PEAK_HBM_GBS
        ctx = fun.ctx
    elif cfg == 'c3':
        from synthetic import clustered_problem
        N, V, K = n_override or 1_000_000, 31, 32
        x, w, fg, fz, lam = clustered_problem(N, V, K, seed=11)
        r0, r1 = shard_rows(N, rank, world)
        theta = np.concatenate([fg, fz[r0:r1].ravel()])        # the globals, then THIS rank's simplex rows
        par = vb.ModelParamsDict('params')
        par.push_param(vb.DirichletParamArray('pi', shape=(K,)))
        par.push_param(vb.DirichletParamArray('phi', shape=(V, K)))
        par.push_param(vb.SimplexParam('z', shape=(r1 - r0, K)))
        fun = vb.MixtureObjective(par, x[r0:r1], pi_prior=1.2, phi_prior=0.9, weights=w[r0:r1], device=device_index)
        shard(fun.ctx)
        fun.keep_logits_resident = True                       # the local part of the point (N x 31 logits) stays in HBM
        D = fun.n_global

        def step():
            return fun.global_hessian(theta, want_host=False)      # the result stays in HBM (what chol_factor_last factors)

        def final():
            return fun.global_hessian(theta)
        metric = 'Schur-complement ELBO-Hessian builds/sec, Dirichlet-multinomial mixture K=32, N={:g} obs x D={} global free params'.format(float(N), D)
        workload = ('config 3: Dirichlet-multinomial mixture K=32, V=31, N={}; one step = per-row simplex blocks eliminated on the '
                    'GPU (rows kernel + Kronecker GEMM + statistics) + device Schur assembly of the {} x {} global block, left in HBM').format(N, D, D)
        # the repo's algorithm is a 528 x N x 528 fp64 GEMM (packed lower triangles of both Kronecker factors): matrix-core bound;
        # SURVEY 8(d)'s byte count ("if the blocks stay on chip") is carried beside it
        bound, alg, unit, peak = 'mfma', 2.0 * 528 * 528 * (r1 - r0), 'TFLOP/s', PEAK_FP64_MFMA_TFLOPS
        extra['survey_8d_bytes'] = 8.0 * (r1 - r0) * (K - 1 + V) + 8.0 * D * D
        ctx = fun.ctx
    elif cfg == 'c4':
        from synthetic import lmm_par as _lmm_par
        N, p, G = n_override or 1_250_000, 43, 10_000
        x = rng.normal(size=(N, p)); gid = rng.integers(0, G, size=N).astype(np.int32); gid[:G] = np.arange(G)
        y = x @ rng.normal(size=p) + rng.normal(size=G)[gid] * 0.7 + rng.normal(size=N) * 0.5
        wts = rng.uniform(0.5, 1.5, N)
        r0, r1 = shard_rows(N, rank, world)                    # whole groups per rank are NOT required: group sums are summed too
        par = _lmm_par(vb, p, G)
        fun = vb.LMMObjective(par, x[r0:r1], y[r0:r1], gid[r0:r1], G, weights=wts[r0:r1], device=device_index)
        shard(fun.ctx)
        theta = par.get_free()
        D = fun.n_global

        def step():
            fun.invalidate_stats()                                # the pass over the observations is part of every step
            return fun.global_hessian(theta, want_host=False)      # the result stays in HBM (what chol_factor_last factors)

        def final():
            return fun.global_hessian(theta)
        metric = 'arrow-Hessian Schur-complement builds/sec, hierarchical LMM G=1e4 groups, N={:g} obs x D={} global free params'.format(float(N), D)
        workload = ('config 4: hierarchical LMM p=43, G=1e4, N={} (one of the eight 1.25e6-row shards of the N=1e7 problem unless --n-obs says otherwise); '
                    'one step = ONE library call: sufficient statistics in one pass (Gram q=44 + per-group sums, weights resident in group order) + elimination '
                    'of the 2G local parameters + closed forms evaluated on the device where the statistics lie + arrow-Hessian assembly, left in HBM').format(N)
        bound, alg, unit, peak = 'hbm', float(r1 - r0) * (8.0 * (p + 2) + 4.0), 'GB/s', PEAK_HBM_GBS
        ctx = fun.ctx
    else:
        N, d = n_override or 1_000_000, 63
        yy = rng.normal(size=(N, d))
        r0, r1 = shard_rows(N, rank, world)
        par = vb.ModelParamsDict('p'); par.push_param(vb.MVNParam('mu', dim=d)); par.push_param(vb.WishartParam('lambda', size=d))
        fun = vb.WishartMVNObjective(par, yy[r0:r1], device=device_index)
        shard(fun.ctx)
        par['lambda']['df'].set(d + 5.0)
        theta = par.get_free()
        D = theta.size

        def step():
            return fun.gram(theta, want_host=False)              # operand generated on the device, result left in HBM

        def final():
            return fun.gram(theta)
        metric = 'G^T G (per-observation ELBO-gradient Gram matrix) builds/sec, Wishart+MVN N={:g} obs x D={} free params'.format(float(N), D)
        workload = ('config 5: Wishart + MVN full-covariance model d=63 -> D=4096, N={}; one step = G^T G with the Kronecker rows of G '
                    'generated on chip (fp64-MFMA SYRK over the packed lower triangle of z z^T, 2080 virtual columns; the per-coordinate matrices '
                    'of the model are 0.2 % dense and enter as gathers, never formed; structured packing-Jacobian products), result left in HBM').format(N)
        # algorithmic flops: the SYRK over Pv = q (q + 1) / 2 = 2080 virtual columns (z z^T is symmetric: rounds 1-3 formed all
        # 64 q = 4096 columns of z (x) z, N 4096 4097 = 1.68e13 flops, and priced the step against that)
        pv = (d + 1) * (d + 2) // 2
        bound, alg, unit, peak = 'mfma', float(r1 - r0) * pv * (pv + 1), 'TFLOP/s', PEAK_FP64_MFMA_TFLOPS
        extra['survey_8d_flops'] = float(r1 - r0) * D * (D + 1)
        extra['note'] = ('algorithmic_per_step is the arithmetic of the algorithm that runs: the SYRK over the packed lower triangle of z z^T, '
                         'N Pv (Pv + 1) with Pv = {} virtual columns.  SURVEY 8(d) prices the naive product of the N x D gradient matrix with itself, '
                         'N D (D + 1) = survey_8d_flops ({:.2f} x as many), which this library never forms (G is quadratic in the data: '
                         'G^T G = M~^T K4 M~); priced against that figure the step would read more than the matrix-core peak, so it is carried '
                         'beside, not used').format(pv, D * (D + 1) / float(pv * (pv + 1)))
        ctx = fun.ctx

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
    elapsed, prof = _timed_steps(step, warmup, steps, ctx, fence)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if backend == 'gloo':
            th = t.cpu(); dist.all_reduce(th, op=dist.ReduceOp.MAX); t = th
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = prof['wsyrk_ms'] / steps                      # all statistics-kernel launches of one step
    ms_per_step = elapsed / steps * 1e3
    # the exchange apart from the compute, per rank (HIP events around the sum-over-ranks hook inside the library)
    per_rank = _gather_floats(torch, dist, [kernel_ms, prof['reduce_ms'] / steps, prof['reduce_calls'] / steps], dev, backend, use_dist)
    scale = 1e9 if unit == 'GB/s' else 1e12
    achieved = alg / (kernel_ms * 1e-3) / scale if kernel_ms > 0 else 0.0
    out = {
        'metric': metric, 'value': steps / elapsed, 'unit': 'builds/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': workload, 'n_obs_total': N, 'n_obs_per_gpu': r1 - r0, 'n_free': D,
                   'ranks_seen': dist.get_world_size() if use_dist else 1,
                   'backend': (backend + (' (in-library communicator)' if native else '')) if use_dist else 'none (single process)',
                   'parallelism': 'observation shards x{} + 1 device all-reduce per statistics call'.format(world),
                   'result_fingerprint': _fingerprint(np, final())},
        'roofline': {'bound': bound, 'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak, 'traffic': None,
                     'kernel': 'ALL statistics kernels of one step on this rank (library profile marks: {} launch groups per step)'.format(
                         prof['wsyrk_calls'] // max(steps, 1)),
                     'kernel_ms': kernel_ms, 'algorithmic_per_step': alg,
                     'step_frac': alg / (ms_per_step * 1e-3) / scale / peak},
    }
    out['roofline'].update(extra)
    out['exchange'] = {'per_rank_kernel_ms': [r[0] for r in per_rank], 'allreduce_ms': [r[1] for r in per_rank],
                       'allreduce_calls_per_step': per_rank[0][2],
                       'note': 'HIP events on each rank\'s stream: all statistics kernels of a step / around the sum-over-ranks hook '
                               '(collective + wait for the slowest rank); the rest of ms_per_step is N-independent host and device assembly'}
    if cfg == 'c5' and world == 1 and own_group:
        # the LRVB solve of the configuration: exact Hessian (sufficient statistics), Cholesky, CG on the resident matrix
        obj = vb.Objective(par, fun)
        t0 = time.perf_counter(); H = obj.fun_free_hessian(theta); t1 = time.perf_counter()
        Hs = H + (0.1 - min(0.0, float(np.linalg.eigvalsh(H).min()))) * np.eye(D)      # not at an optimum: shifted for the timing
        fun.ctx.chol_factor(Hs); t2 = time.perf_counter(); fun.ctx.chol_factor(Hs); t3 = time.perf_counter()
        b = rng.normal(size=D)
        fun.ctx.cg_solve_matrix(Hs, b, tol=1e-8); t4 = time.perf_counter()
        _, info, iters = fun.ctx.cg_solve_matrix(None, b, tol=1e-8); t5 = time.perf_counter()
        out['lrvb_solve_ms'] = {'exact_hessian_build': (t1 - t0) * 1e3, 'cho_factor_host_matrix_in': (t3 - t2) * 1e3,
                                'cg_resident_matrix_tol1e-8': (t5 - t4) * 1e3, 'cg_iterations': int(iters), 'cg_info': int(info)}
    if use_dist and own_group:
        if native:
            ctx.comm_destroy()
        dist.destroy_process_group()
    return out if rank == 0 else None


def embedded_configs(args, env, budget_s=90.0):
    """Short runs of c2..c5 inside the headline's process (N = 1): {cfg: {ms_per_step, kernel_ms, roofline...}} for the
    `configs` object of the one JSON line.  Bounded: a configuration is skipped once the budget is spent."""
    out, t0 = {}, time.perf_counter()
    order = (('c2', 100, 20), ('c4', 100, 20), ('c3', 8, 2), ('c5', 5, 2))      # (sub-millisecond steps: enough of them for a steady reading)
    if os.environ.get('LRVB_EMBED_ORDER'):
        order = tuple(o for name in os.environ['LRVB_EMBED_ORDER'].split(',') for o in order if o[0] == name)
    for cfg, steps, warmup in order:
        if time.perf_counter() - t0 > budget_s:
            out[cfg] = {'skipped': 'time budget of the embedded runs spent'}
            continue
        try:
            r = run_config(args, cfg=cfg, steps=steps, warmup=warmup, env=env)
            out[cfg] = {'metric': r['metric'], 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'steps': steps,
                        'kernel_ms': r['roofline']['kernel_ms'], 'roofline': r['roofline'],
                        'result_fingerprint': r['config']['result_fingerprint']}
        except Exception as e:                                # the headline line must not be lost to a secondary configuration
            out[cfg] = {'error': '{}: {}'.format(type(e).__name__, e)}
    out['seconds'] = time.perf_counter() - t0
    return out


def dist_setup(args, torch, dist):
    """Rank environment, device and process group of this process: (world, rank, local_rank, device, backend, use_dist,
    rehearse).  One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torchrun); RCCL ("nccl") unless rehearsing."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench: --gpus {} but WORLD_SIZE is {} (start the ranks with `python bench.py --gpus N` or '
                         'with torchrun --nproc-per-node N bench.py --gpus N)'.format(args.gpus, world))
    # LRVB_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (process group, stats all-reduce,
    # finish) even with one rank -- used to rehearse the N > 1 path on a one-GPU box
    force_sharded = os.environ.get('LRVB_BENCH_FORCE_SHARDED', '0') == '1'
    use_dist = world > 1 or force_sharded
    # LRVB_BENCH_REHEARSE_ONE_GPU=1: every rank uses GPU 0 and the exchange goes over gloo -- a correctness
    # rehearsal of the multi-rank path on a one-GPU box (RCCL refuses two ranks on one device); the numbers it
    # prints are not a measurement
    rehearse = os.environ.get('LRVB_BENCH_REHEARSE_ONE_GPU', '0') == '1'
    backend = None
    if rehearse:
        local_rank = 0
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        raise SystemExit('bench: rank {} needs GPU {} but this node shows {} GPU(s); a one-GPU box can only REHEARSE '
                         'the multi-rank path (LRVB_BENCH_REHEARSE_ONE_GPU=1: all ranks on GPU 0, gloo exchange)'
                         .format(rank, local_rank, n_dev))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = 'gloo' if rehearse else 'nccl'
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    return world, rank, local_rank, dev, backend, use_dist, rehearse


def main(args):
    import numpy as np
    if getattr(args, 'config', 'h') != 'h':
        return run_config(args)
    import torch
    import torch.distributed as dist
    import lrvb_amd as vb
    from lrvb_amd.distributed import ShardedHessian, DeviceEngine, shard_rows

    world, rank, local_rank, dev, backend, use_dist, rehearse = dist_setup(args, torch, dist)

    # The other four BASELINE.json configurations, short runs in this process (SURVEY.md section 8(d) lists all five), FIRST:
    # their steps are a few dozen small kernels and host round trips each, and measured after the headline phase (8 GB of
    # tensors freed, a 30 s numpy leg on 64 BLAS threads) the same steps came out 2-7 x slower than in a process of their own.
    configs_out = None
    if world == 1 and rank == 0 and not args.no_configs:
        import gc
        configs_out = embedded_configs(args, (world, rank, dev, backend, use_dist))
        gc.collect()
        torch.cuda.empty_cache()

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
    # LRVB_BENCH_NATIVE_RCCL=1: the exchange runs inside the library (lrvb_comm_init: its own RCCL communicator, the id
    # carried by the process group) and a step is the single call lrvb_hessian_dev; default: torch.distributed's
    # all-reduce of the statistics buffer between lrvb_hessian_partial_dev and lrvb_hessian_finish_dev
    native = use_dist and backend == 'nccl' and os.environ.get('LRVB_BENCH_NATIVE_RCCL', '0') == '1'
    if native:
        from lrvb_amd.distributed import native_comm_init
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        native_comm_init(ctx)
    engine = DeviceEngine(ctx, dev) if (use_dist and not native) else None
    sharded = ShardedHessian(engine) if (use_dist and not native) else None

    def step():
        if not use_dist or native:
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
    if sharded is not None:
        sharded.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Hout = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    # the three phases of a sharded build on this rank's stream: statistics kernels | exchange | N-independent assembly
    phases = None
    if sharded is not None:
        k_ms, ar_ms, fin_ms, n_marks = sharded.phase_ms()
        sharded.timing(False)
        phases = [k_ms / max(n_marks, 1), ar_ms / max(n_marks, 1), fin_ms / max(n_marks, 1)]
    elif native:
        nb = max(prof['build_calls'], 1)
        k_ms = prof['wsyrk_ms'] / nb + prof['pass_ms'] / nb
        phases = [k_ms, prof['reduce_ms'] / nb, prof['build_ms'] / nb - k_ms - prof['reduce_ms'] / nb]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        if backend == 'gloo':
            th = t.cpu(); dist.all_reduce(th, op=dist.ReduceOp.MAX); t = th
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed

    # dominant kernel: weighted SYRK of this rank's shard (algorithmic flops / HIP-event time)
    ws_ms = prof['wsyrk_ms'] / max(prof['wsyrk_calls'], 1)
    ws_flops = float(n_local) * D * (D + 1)
    achieved = ws_flops / (ws_ms * 1e-3) / 1e12 if ws_ms > 0 else 0.0
    # HBM traffic of that kernel comes from a separate rocprofv3 --pmc run (counters cannot be read inside a timed
    # run): the committed summary names the commit and kernel it was measured on, and is quoted only for the exact
    # shape it was measured at
    traffic, traffic_src = None, None
    tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tfile) and world == 1 and N_total == 1000000 and D == 1024 and not args.n_splits:
        try:
            rec = json.load(open(tfile))
            traffic = rec.get('wsyrk_hbm_bytes_per_launch')
            traffic_src = {k: rec.get(k) for k in ('measured_at_commit', 'kernel', 'source', 'method') if k in rec}
        except Exception:
            traffic = None
    pass_ms = prof['pass_ms'] / max(prof['pass_calls'], 1)
    out = {
        'metric': 'ELBO-Hessian builds/sec, N={:g} obs x D={} free params'.format(float(N_total), D),
        'value': value, 'unit': 'hessian_builds/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'headline dense-design {} GLM-type ELBO term: N={} observations x D={} free '
                               'parameters (last {} box-constrained lb=0), Gaussian prior; one step = one dense '
                               'Hessian build with theta, the weights w (re-read by the kernels in every build, uploaded once: '
                               'the 8 MB host-to-device copy of w is outside the timed region) and X, y resident in HBM, H left '
                               'in HBM'.format(args.loss, N_total, D, n_pos),
                   'n_obs_total': N_total, 'n_obs_per_gpu': n_local, 'n_free': D,
                   'ranks_seen': dist.get_world_size() if use_dist else 1,
                   'backend': (backend + (' (in-library communicator)' if native else '')) if use_dist else 'none (single process)',
                   'parallelism': 'observation shards x{} + 1 sum all-reduce per build'.format(world)},
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / PEAK_FP64_MFMA_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_src,
                     'kernel': 'weighted SYRK (v_mfma_f64_16x16x4_f64)', 'kernel_ms': ws_ms,
                     'flops_per_launch': ws_flops,
                     # Gaussian loss: the build runs no separate pass over X (the gradient comes from the SYRK's own sums)
                     'pass_kernel_ms': pass_ms if prof['pass_calls'] else None,
                     'pass_kernel_GBs': (8.0 * n_local * (D + 3)) / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else None},
    }

    if phases is not None:
        rows = _gather_floats(torch, dist, phases, dev, backend, use_dist)
        out['exchange'] = {'per_rank_kernel_ms': [r[0] for r in rows], 'allreduce_ms': [r[1] for r in rows],
                           'per_rank_finish_ms': [r[2] for r in rows], 'sum_ms_rank0': sum(rows[0]),
                           'note': 'device-event time of the three phases of a sharded build on each rank\'s stream: statistics '
                                   'kernels (partial) | sum all-reduce of the {:.1f} MB statistics buffer (collective + wait for the '
                                   'slowest rank) | replicated N-independent assembly; they are contiguous on one stream, so their sum '
                                   'is the device time of a step'.format(ctx.stats_size() * 8 / 1e6)}
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
        ctx.set_tuning(args.n_splits, 8)                     # matrix-free figures first: never use the resident Hessian of the build
        ctx.cg_solve(theta_h, rhs[0])
        ctx.set_quad_scale(ctx.quad_scale)                   # forget the point state (the first of the 16 solves rebuilds it)
        t3 = time.perf_counter()
        iters = []
        for q in range(16):
            _, info, it = ctx.cg_solve(theta_h, rhs[q], tol=1e-8)
            iters.append(it if info == 0 else -1)
        t4 = time.perf_counter()
        out['lrvb_solve_ms']['cg_16_rhs_one_by_one'] = (t4 - t3) * 1e3
        # the same 16 systems in lockstep: one pair of passes over X per iteration for all of them
        ctx.cg_solve_multi(theta_h, rhs[:2], tol=1e-8)
        ctx.set_quad_scale(ctx.quad_scale)                   # forget the point state: the first solve below prepares it itself
        t5 = time.perf_counter()
        _, infos, its = ctx.cg_solve_multi(theta_h, rhs, tol=1e-8)
        t6 = time.perf_counter()
        ctx.cg_solve_multi(theta_h, rhs, tol=1e-8)            # same point again: eta, J, gradient and curvature are still in place
        t6b = time.perf_counter()
        out['lrvb_solve_ms']['cg_16_rhs_tol1e-8'] = (t6 - t5) * 1e3
        out['lrvb_solve_ms']['cg_16_rhs_tol1e-8_same_point_again'] = (t6b - t6) * 1e3
        out['lrvb_solve_ms']['cg_iterations'] = [int(i) if f == 0 else -1 for i, f in zip(its, infos)]
        out['lrvb_solve_ms']['cg_iterations_one_by_one'] = iters
        # the same 16 systems AFTER A BUILD at the same point (ConjugateGradientSolver at an optimum whose Hessian was just
        # built): the products run against the Hessian the build left in the context -- no pass over X
        ctx.set_tuning(args.n_splits, 0)
        ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), D)
        ctx.cg_solve_multi(theta_h, rhs[:2], tol=1e-8)
        ctx.sync()
        t7 = time.perf_counter()
        Xr, infos_r, its_r = ctx.cg_solve_multi(theta_h, rhs, tol=1e-8)
        t8 = time.perf_counter()
        out['lrvb_solve_ms']['cg_16_rhs_tol1e-8_after_build_resident_hessian'] = (t8 - t7) * 1e3
        out['lrvb_solve_ms']['cg_iterations_resident_hessian'] = [int(i) if f == 0 else -1 for i, f in zip(its_r, infos_r)]
        # ... and one by one at a point NOBODY built (the reference's solver loop over masks): past max(8, D / 64) matrix-free
        # products at the point the library builds its Hessian itself and serves the remaining solves from it
        theta_new = theta_h + 1e-3                           # a point whose Hessian is not resident
        ctx.sync()
        t9 = time.perf_counter()
        for q in range(16):
            ctx.cg_solve(theta_new, rhs[q], tol=1e-8)
        t10 = time.perf_counter()
        out['lrvb_solve_ms']['cg_16_rhs_one_by_one_hessian_built_on_the_way'] = (t10 - t9) * 1e3
        if args.loss == 'gaussian':
            # the headline (Gaussian) build needs no separate pass over X; a loss whose curvature depends on the linear
            # predictor does (pass -> SYRK).  Same X, same layout, logistic loss on thresholded responses, for the record:
            yb = (y > 0).double()
            ctx2 = vb.DeviceContext(blocks, loss='logistic', n_obs=n_local, n_cols=D, quad_kind=vb._hip.QUAD_DIAG,
                                    device=local_rank)
            ctx2.set_data_dev(vb._hip.SLOT_X, X.data_ptr(), n_local, D)
            ctx2.set_data_dev(vb._hip.SLOT_Y, yb.data_ptr(), n_local, 1)
            ctx2.set_weights_dev(w.data_ptr(), n_local)
            ctx2.set_data(vb._hip.SLOT_QUAD_A, np.full(D, prior_info))
            H2 = torch.empty((D, D), dtype=torch.float64, device=dev)
            for _ in range(2):
                ctx2.hessian_dev(theta.data_ptr(), H2.data_ptr(), D)
            ctx2.sync()
            t7 = time.perf_counter()
            for _ in range(5):
                ctx2.hessian_dev(theta.data_ptr(), H2.data_ptr(), D)
            ctx2.sync()
            out['config']['logistic_build_ms'] = (time.perf_counter() - t7) / 5 * 1e3
            del ctx2, H2, yb
        # SURVEY 8(d) defines a build as "(theta, w) on the host -> H": the same step with theta and the 8 MB weight vector
        # uploaded in every build (pinned host buffers, stream-ordered copies into the buffers the context has adopted), H
        # left in HBM.  `value` above keeps the inputs resident; this is the PCIe-inclusive figure beside it.
        th_host = theta.cpu().pin_memory()
        w_host = w.cpu().pin_memory()

        def step_host_inputs():
            theta.copy_(th_host, non_blocking=True)
            w.copy_(w_host, non_blocking=True)
            ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), D)
        for _ in range(2):
            step_host_inputs()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_host_inputs()
        fence()
        out['ms_per_step_host_inputs'] = (time.perf_counter() - t0) / args.steps * 1e3
        out['config']['host_inputs_note'] = ('ms_per_step_host_inputs: theta ({} B) and w ({:.1f} MB) copied from pinned host memory '
                                             'in every build, H left in HBM'.format(8 * D, 8e-6 * n_local))
        if not args.no_cpu_baseline and rank == 0:
            def fetch_rows(a, b):
                return X[a:b].cpu().numpy(), y[a:b].cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(fetch_rows, N_total, D, n_pos, args.loss, lik_info, prior_info,
                                               theta.cpu().numpy(), budget_s=args.cpu_budget_s)
            # PARITY at the benchmark's own size: the matrix the timed region built against the oracle's Hessian over the same
            # 1e6 rows (the strong-numpy leg forms it anyway), plus the gradient.  The checker leg: outside every timed region.
            full = out['cpu_baseline'].pop('_oracle_full', None)
            if full is not None:
                ctx.hessian_dev(theta.data_ptr(), H.data_ptr(), D)
                ctx.sync()
                H_gpu = H.cpu().numpy()
                g_gpu = ctx.grad(theta.cpu().numpy())
                Ho, go = full['hessian'], full['grad']
                out['parity'] = {
                    'max_rel_err_hessian': float(np.max(np.abs(H_gpu - Ho)) / np.max(np.abs(Ho))),
                    'max_rel_err_grad': float(np.max(np.abs(g_gpu - go)) / np.max(np.abs(go))),
                    'asymmetry': float(np.max(np.abs(H_gpu - H_gpu.T)) / np.max(np.abs(Ho))),
                    'rows': int(N_total), 'n_free': int(D),
                    'against': 'oracle (numpy fp64: X^T diag(c) X over all rows in 65,536-row chunks + convert_vector_to_free_hessian); '
                               'errors relative to the largest entry; tolerance of the GPU suite for sums over observations: 1e-11'}
            else:
                out['parity'] = {'skipped': 'the strong-numpy leg did not reach all rows within its time budget'}
    if configs_out is not None:
        out['configs'] = configs_out
    if use_dist:
        dist.destroy_process_group()
    return out if rank == 0 else None


def run_with_clean_stdout(args=None):
    """The contract is ONE JSON line on stdout.  Libraries underneath write there too (RCCL prints a five-line
    version banner to stdout when the first communicator comes up, on every rank), so file descriptor 1 points at
    stderr while the bench runs and is restored only for the result line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    out = main(args) if args is not None else main()
    sys.stdout.flush()
    if out is not None:                      # rank 0; the other ranks keep writing to stderr until they exit
        os.dup2(saved, 1)
        os.close(saved)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    _args = parse()
    if needs_launch(_args, os.environ):
        sys.exit(launch_ranks(_args, sys.argv[1:]))
    run_with_clean_stdout(_args)
