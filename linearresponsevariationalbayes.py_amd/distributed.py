"""Observation-sharded Hessian build over the GPUs of one node: one process and one device
context per GPU, one sum all-reduce of the packed sufficient statistics per build.

The reference has no distributed code; this is the data-parallel
form of its `Objective.fun_free_hessian` (LRVB/SparseObjectives.py:156-158).  The objective is a
sum over observations, so

    stats_r = [ value_r | d f_data,r / d beta | tile-packed X_r^T diag(w loss'') X_r ]    (rank r's rows)
    stats   = all-reduce-sum_r stats_r                      (RCCL over xGMI: torch.distributed "nccl")
    H_free  = finish(theta, stats)                          (N-independent, replicated on every rank)

At D = 1024 the exchanged buffer is 4.7 MB (36 tiles of 128 x 128 fp64 + 1025 doubles), i.e.
one latency-bound collective per ~2-15 ms of compute, so a single un-bucketed all-reduce is
used; the solve after it is replicated.  `engine` abstracts who produces the statistics so
that the same sharding / collective logic is exercised on CPU with the `gloo` backend in the
tests.
"""
import numpy as np


def shard_rows(n_total, rank, world_size):
    """Contiguous row block [start, stop) of rank `rank`; blocks differ by at most one row."""
    base, rem = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class DeviceEngine(object):
    """Statistics and assembly on this rank's GPU through the C ABI (device pointers of torch
    tensors; torch is used for memory and the collective only)."""

    def __init__(self, ctx, torch_device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.device = torch_device
        self.stats = torch.empty(ctx.stats_size(), dtype=torch.float64, device=torch_device)
        self.H = torch.empty((ctx.D, ctx.D), dtype=torch.float64, device=torch_device)
        # The context runs on torch's current stream: the kernels, the RCCL all-reduce (which
        # orders itself after the current stream) and the assembly are stream-ordered, with no
        # host synchronisation between them.
        ctx.set_stream(torch.cuda.current_stream(torch_device).cuda_stream)

    def partial(self, theta):
        self.ctx.hessian_partial_dev(theta.data_ptr(), self.stats.data_ptr())
        return self.stats

    def finish(self, theta, stats):
        self.ctx.hessian_finish_dev(theta.data_ptr(), stats.data_ptr(), self.H.data_ptr(), self.ctx.D)
        return self.H


class ShardedHessian(object):
    """build(theta) -> free-coordinate Hessian, identical on every rank."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self._marks = None              # timing on: list of (start, after partial, after all-reduce, after finish) event tuples

    def timing(self, on=True):
        """Record device events around the three phases of every build (partial statistics | exchange | assembly) on the
        stream the build runs on: the first multi-GPU run can then say which phase a shortfall sits in.  No host
        synchronisation is added; `phase_ms()` collects after the caller has synchronised."""
        self._marks = [] if on else None

    def phase_ms(self):
        """(kernel_ms, allreduce_ms, finish_ms) summed over the builds since `timing(True)`, and their count."""
        marks, self._marks = self._marks or [], ([] if self._marks is not None else None)
        tot = [0.0, 0.0, 0.0]
        for ev in marks:
            for k in range(3):
                tot[k] += ev[k].elapsed_time(ev[k + 1])
        return tot[0], tot[1], tot[2], len(marks)

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_world_size(self.group)
        return None, 1

    def build(self, theta):
        dist, world = self._world()
        ev = None
        if self._marks is not None and getattr(self.engine, 'torch', None) is not None:
            torch = self.engine.torch
            stream = torch.cuda.current_stream(self.engine.device)      # the stream the engine's context runs on
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(stream)
        stats = self.engine.partial(theta)
        if ev is not None:
            ev[1].record(stream)
        if dist is not None:            # also with one rank: keeps the collective path exercised
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        if ev is not None:
            ev[2].record(stream)
        H = self.engine.finish(theta, stats)
        if ev is not None:
            ev[3].record(stream)
            self._marks.append(ev)
        return H


class _DevicePointer(object):
    """A raw device address as a `__cuda_array_interface__` object, so that torch can wrap library-owned memory."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}


def torch_reduce_hook(torch_device, group=None):
    """The callable for `DeviceContext.set_reduce_hook`: one `torch.distributed` sum all-reduce (RCCL over xGMI with
    the `nccl` backend) of the library's own device buffer, in place -- no host hop.  The context must run on
    torch's current stream (`ctx.set_stream(torch.cuda.current_stream().cuda_stream)`) so that the collective is
    ordered between the kernels that produce and consume the buffer.  (With the `gloo` backend -- one-GPU rehearsals
    of the multi-rank path -- torch stages the same device tensor through the host itself.)"""
    import torch
    import torch.distributed as dist
    views = {}

    def hook(ptr, n, stream):
        t = views.get((ptr, n))
        if t is None:
            t = views[(ptr, n)] = torch.as_tensor(_DevicePointer(ptr, n), device=torch_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return hook


def native_comm_init(ctx, group=None):
    """Create this rank's part of an in-library RCCL communicator for `ctx` (lrvb_comm_init): rank 0 draws the id, the
    process group (any backend) carries its 128 bytes to the other ranks.  Afterwards every observation sum of the
    declared-objective entry points is all-reduced inside the library, with no torch tensor in the data path."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [type(ctx).comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.comm_init(world, rank, box[0])
    return world, rank


class ShardedObjective(object):
    """Value, gradient, Hessian-vector products, CG solves and trust-ncg fits of an objective whose observations
    are sharded over the ranks of a process group ("one D-vector all-reduce per CG
    iteration when the HVP is data-sharded").  Every rank holds a context over ITS rows; every method returns the
    GLOBAL quantity, bit-identical on every rank, so host-side iterations (scipy's cg / trust-ncg, as the reference
    drives them) stay in lockstep without further coordination.

    * `DeviceContext` on a GPU (`torch_device` given): the library's sum-over-ranks hook is installed
      (`lrvb_set_reduce_hook`).  Each sum over observations is all-reduced ON THE DEVICE, in the library's own buffer,
      before the N-independent terms are added; nothing about the context is rescaled, so the same context goes on to
      build the sharded Hessian (`ShardedHessian`, or `ctx.hessian` directly) correctly, and the device-side loops
      (`cg_solve`, `cg_solve_multi`, `minimize_trust_ncg(on_device=True)`) run sharded with no host hop per product.
    * any other object with `value / grad / hvp / set_quad_scale` (the CPU stand-ins of the gloo tests): the
      N-independent quadratic term is scaled by 1 / world_size AROUND each call and restored afterwards, which makes the
      global objective the plain sum of the local ones; the local result is summed with one all-reduce."""

    def __init__(self, ctx, torch_device=None, group=None):
        import torch.distributed as dist
        self.ctx = ctx
        self.device = torch_device
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.on_device = torch_device is not None and hasattr(ctx, 'set_reduce_hook')
        if self.on_device:
            import torch
            ctx.set_stream(torch.cuda.current_stream(torch_device).cuda_stream)
            if self.world > 1 or (dist.is_available() and dist.is_initialized()):
                ctx.set_reduce_hook(torch_reduce_hook(torch_device, group))

    def close(self):
        """Remove the hook (the context is a plain single-process context again)."""
        if self.on_device:
            self.ctx.set_reduce_hook(None)

    def _summed(self, call, *args):
        if self.on_device:
            return np.atleast_1d(np.asarray(call(*args), dtype=np.float64))        # reduced inside the library
        base = float(getattr(self.ctx, 'quad_scale', 1.0))
        self.ctx.set_quad_scale(base / self.world)
        try:
            local = np.atleast_1d(np.asarray(call(*args), dtype=np.float64))
        finally:
            self.ctx.set_quad_scale(base)
        return allreduce_stats(local, self.device, self.group)

    def value(self, theta):
        return float(self._summed(self.ctx.value, theta)[0])

    def grad(self, theta):
        return self._summed(self.ctx.grad, theta)

    def hvp(self, theta, v):
        return self._summed(self.ctx.hvp, theta, v)

    def cg_solve(self, theta, b, x0=None, tol=1e-8, maxiter=None, M=None):
        """H(theta)^-1 b (LRVB/ConjugateGradient.py:63-85): returns (x, info).  On the device the whole loop runs in
        the library (`lrvb_cg_solve`), one in-place D-vector all-reduce per iteration; otherwise scipy's cg drives
        the sharded product."""
        theta = np.asarray(theta, dtype=np.float64)
        if self.on_device:
            x, info, _ = self.ctx.cg_solve(theta, np.asarray(b, dtype=np.float64), x0=x0, Minv=M, tol=tol,
                                           maxiter=maxiter or 0)
            return x, info
        import scipy.sparse.linalg as sla
        D = theta.size
        op = sla.LinearOperator((D, D), matvec=lambda v: self.hvp(theta, np.asarray(v, dtype=np.float64).ravel()))
        return sla.cg(op, np.asarray(b, dtype=np.float64), x0=x0, rtol=tol, atol=0.0, maxiter=maxiter, M=M)

    def minimize_trust_ncg(self, x0, gtol=1e-6, maxiter=50, disp=False, on_device=False):
        """The fit of `minimize_objective_trust_ncg` (LRVB/OptimizationUtils.py:44-75) over sharded observations;
        `on_device=True` runs the optimiser loop in the library (`lrvb_minimize_trust_ncg`) on every rank."""
        if on_device:
            if not self.on_device:
                raise ValueError('on_device needs a DeviceContext on a GPU')
            import scipy.optimize
            _, x, info = self.ctx.minimize_trust_ncg(np.asarray(x0, dtype=np.float64), gtol=gtol, maxiter=maxiter)
            return scipy.optimize.OptimizeResult(x=x, fun=info['fun'], status=info['status'], success=info['status'] == 0,
                                                 nit=info['nit'], nfev=info['nfev'], njev=info['njev'], nhev=info['nhev'])
        import scipy.optimize
        return scipy.optimize.minimize(self.value, np.asarray(x0, dtype=np.float64), jac=self.grad, hessp=self.hvp,
                                       method='trust-ncg', options={'maxiter': maxiter, 'gtol': gtol, 'disp': disp})


def allreduce_stats(flat, torch_device=None, group=None):
    """Sum a flat float64 statistics vector over all ranks (sufficient statistics of the
    quadratic-in-data and hierarchical objectives: `LMMObjective.local_stats()`), returning a numpy
    array.  With the `nccl` backend pass the rank's CUDA device; with `gloo` leave it None."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float64).copy())
    if torch_device is not None:
        t = t.to(torch_device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def stats_layout(n_cols):
    """(offset of value, offset of gradient, offset of tiles, total) in doubles; mirrors
    lrvb_stats_size in csrc/lrvb_api.hip."""
    nb = (n_cols + 127) // 128
    tiles = nb * (nb + 1) // 2 * 128 * 128
    return 0, 1, 1 + n_cols, 1 + n_cols + tiles


def pack_tiles(S):
    """Tile-packed lower triangle (128 x 128 tiles, row-major tile order) of a symmetric P x P
    matrix -- the layout `lrvb_hessian_partial_dev` writes.  Used by host-side engines/tests."""
    P = S.shape[0]
    nb = (P + 127) // 128
    out = np.zeros((nb * (nb + 1) // 2, 128, 128))
    t = 0
    for bi in range(nb):
        for bj in range(bi + 1):
            blk = S[bi * 128:(bi + 1) * 128, bj * 128:(bj + 1) * 128]
            out[t, :blk.shape[0], :blk.shape[1]] = blk
            t += 1
    return out.ravel()


def unpack_tiles(flat, P):
    nb = (P + 127) // 128
    tiles = np.asarray(flat).reshape(nb * (nb + 1) // 2, 128, 128)
    S = np.zeros((nb * 128, nb * 128))
    t = 0
    for bi in range(nb):
        for bj in range(bi + 1):
            S[bi * 128:(bi + 1) * 128, bj * 128:(bj + 1) * 128] = tiles[t]
            t += 1
    S = np.tril(S) + np.tril(S, -1).T
    return S[:P, :P]
