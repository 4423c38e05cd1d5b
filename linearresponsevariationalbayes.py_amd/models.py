"""Declared objectives evaluated on the MI355X through liblrvb_hip.so.

The reference's `Objective(par, fun)` takes an opaque zero-argument Python closure and lets
autograd trace it (LRVB/SparseObjectives.py:95-116).  A HIP kernel cannot trace Python, so the
objective is DECLARED instead: a `DeviceObjective` is a functor that is still callable with no
arguments (it returns the value at the current state of `par`, like the reference's `fun`) but
also tells the device what to differentiate:

    f(eta) = sum_n w_n loss(y_n, x_n . eta[off:off+P]) + s (1/2 (eta-m)^T A (eta-m) + b^T eta)

`Objective`, `TwoParameterObjective`, `ParametricSensitivityLinearApproximation` and
`ConjugateGradientSolver` accept such a functor where the reference accepts a closure and route
every derivative to the C ABI.  There is no CPU path: without the HIP library, construction
raises.
"""
import ctypes
import numpy as np

from . import _hip
from .packing import VectorParam, HyperVectorParam, ResidentVector, tril_indices

_LOSSES = {None: _hip.LOSS_NONE, 'none': _hip.LOSS_NONE, 'gaussian': _hip.LOSS_GAUSSIAN,
           'logistic': _hip.LOSS_LOGISTIC, 'poisson': _hip.LOSS_POISSON, 'data_only': _hip.LOSS_DATA_ONLY}


def sym_to_vech(A):
    """Row-major lower triangle of (the symmetric part of) A: the vector form of a symmetric matrix parameter
    (LRVB/MatrixParameters.py:16-41, index j + i (i + 1) / 2)."""
    A = _hip.as_f64(A)
    A = 0.5 * (A + A.T)
    return np.ascontiguousarray(A[tril_indices(A.shape[0])])


def vech_to_sym(v):
    v = _hip.as_f64(v).ravel()
    k = int(round((np.sqrt(8.0 * v.size + 1.0) - 1.0) / 2.0))
    L = np.zeros((k, k))
    L[tril_indices(k)] = v
    return L + L.T - np.diag(np.diag(L))


def _block_array(blocks):
    arr = (_hip.BlockDesc * len(blocks))()
    fo = vo = 0
    for i, b in enumerate(blocks):
        arr[i].kind = b['kind']
        arr[i].free_off, arr[i].vec_off = fo, vo
        arr[i].free_size, arr[i].vec_size = b['free_size'], b['vec_size']
        arr[i].dim0, arr[i].dim1 = b['dim0'], b['dim1']
        arr[i].lb, arr[i].ub = b['lb'], b['ub']
        fo += b['free_size']
        vo += b['vec_size']
    return arr, fo, vo


def scratch_context(device=0):
    """A minimal context (one unconstrained parameter, unit quadratic term) for the element-wise device utilities that
    need a device and a stream but no model: `DeviceContext.gh_logistic`, the `ctx=` argument of the Modeling functions."""
    ctx = DeviceContext([dict(kind=_hip.BLOCK_BOX, free_size=1, vec_size=1, dim0=1, dim1=0, lb=-np.inf, ub=np.inf)],
                        loss=None, quad_kind=_hip.QUAD_DIAG, device=device)
    ctx.set_data(_hip.SLOT_QUAD_A, np.ones(1))
    return ctx


class DeviceContext(object):
    """Owns one lrvb_ctx (one HIP device + stream).  Thin, typed wrappers over the C ABI."""

    def __init__(self, blocks, loss=None, n_obs=0, n_cols=0, glm_off=0, lik_info=1.0,
                 quad_kind=_hip.QUAD_NONE, device=0):
        self._lib = _hip.load()
        self._blocks, self.D, self.V = _block_array(blocks)
        desc = _hip.ModelDesc()
        desc.n_blocks = len(blocks)
        desc.loss = _LOSSES[loss] if not isinstance(loss, int) else loss
        desc.blocks = self._blocks
        desc.n_obs, desc.n_cols, desc.glm_off = int(n_obs), int(n_cols), int(glm_off)
        desc.lik_info = float(lik_info)
        desc.quad_kind = int(quad_kind)
        self._h = ctypes.c_void_p()
        self.chol_token = 0            # bumped by every factorisation: lets holders of a factor notice a replacement
        self._hv_ops = None                  # recorded blocks of an hvec assembly (None: not recording)
        self._check(self._lib.lrvb_ctx_create(ctypes.byref(self._h), int(device), ctypes.byref(desc)))
        self.n_obs, self.n_cols = int(n_obs), int(n_cols)
        self.device = int(device)
        self.quad_scale = 1.0
        self.has_reduce_hook = False   # a sum-over-ranks hook is installed: the statistics calls return GLOBAL sums
        self.hook_epoch = 0            # bumped whenever a hook / communicator is installed or removed: host caches of statistics key on it
        self._hook_cb = None

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.lrvb_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data -------------------------------------------------------------------------
    def set_data(self, slot, arr):
        a = _hip.as_f64(arr)
        rows, cols = (a.shape[0], a.shape[1]) if a.ndim == 2 else (a.size, 1)
        self._check(self._lib.lrvb_set_data(self._h, slot, _hip.ptr(a), rows, cols))

    def set_data_dev(self, slot, dev_ptr, rows, cols):
        self._check(self._lib.lrvb_set_data_dev(self._h, slot, ctypes.c_void_p(dev_ptr), rows, cols))

    def set_weights(self, w):
        a = _hip.as_f64(w).ravel()
        self._check(self._lib.lrvb_set_weights(self._h, _hip.ptr(a), a.size))

    def set_weights_dev(self, dev_ptr, n):
        self._check(self._lib.lrvb_set_weights_dev(self._h, ctypes.c_void_p(dev_ptr), n))

    def set_quad_scale(self, s):
        self._check(self._lib.lrvb_set_quad_scale(self._h, float(s)))
        self.quad_scale = float(s)

    def set_lik_info(self, tau):
        self._check(self._lib.lrvb_set_lik_info(self._h, float(tau)))

    def set_reduce_hook(self, fn):
        """Install the sum-over-ranks hook of include/lrvb_hip.h (`lrvb_set_reduce_hook`): `fn(dev_ptr, n, hip_stream)`
        must replace the n doubles at device address dev_ptr by their sum over all ranks, in stream order on
        hip_stream (the context's stream).  None removes it.  An exception raised by `fn` fails the library call
        that triggered it and is re-raised from there."""
        self._hook_error = None
        self.hook_epoch += 1
        if fn is None:
            self._check(self._lib.lrvb_set_reduce_hook(self._h, None, None))
            self._hook_cb = None
            self.has_reduce_hook = False
            return

        def trampoline(_user, buf, n, stream):
            try:
                fn(int(buf or 0), int(n), int(stream or 0))
                return 0
            except BaseException as e:            # never let an exception cross the C frame
                self._hook_error = e
                return -1
        cb = _hip.REDUCE_FN(trampoline)
        self._check(self._lib.lrvb_set_reduce_hook(self._h, ctypes.cast(cb, ctypes.c_void_p), None))
        self._hook_cb = cb      # keep the callback object alive as long as it is installed (the old one until replaced)
        self.has_reduce_hook = True

    # -- in-library RCCL communicator (one process per GPU) ------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes naming a new communicator (rank 0 creates them, every rank passes the same bytes to comm_init)."""
        buf = ctypes.create_string_buffer(128)
        _hip.check(_hip.load().lrvb_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return buf.raw

    def comm_init(self, world_size, rank, comm_id):
        """Join the communicator (collective) and make it this context's sum-over-ranks hook."""
        if len(comm_id) != 128:
            raise ValueError('a communicator id has 128 bytes')
        buf = ctypes.create_string_buffer(bytes(comm_id), 128)
        self._check(self._lib.lrvb_comm_init(self._h, int(world_size), int(rank), ctypes.cast(buf, ctypes.c_void_p)))
        self._hook_cb = None          # only now: a failed call leaves the C side pointing at the old trampoline
        self.has_reduce_hook = True
        self.hook_epoch += 1

    def comm_destroy(self):
        self._check(self._lib.lrvb_comm_destroy(self._h))
        self.hook_epoch += 1
        if self._hook_cb is None:
            self.has_reduce_hook = False

    def allreduce_hessian(self, stats_ptr, n):
        self._check(self._lib.lrvb_allreduce_hessian(self._h, ctypes.c_void_p(stats_ptr), int(n)))

    def _check(self, status):
        """_hip.check, re-raising an exception that the reduce hook raised inside the call."""
        err = getattr(self, '_hook_error', None)
        if err is not None:
            self._hook_error = None
            raise err
        _hip.check(status)

    def set_tuning(self, n_splits=0, flags=0):
        self._check(self._lib.lrvb_set_tuning(self._h, int(n_splits), int(flags)))

    def sync(self):
        self._check(self._lib.lrvb_ctx_sync(self._h))

    def set_stream(self, hip_stream_handle):
        """Run on a caller-owned HIP stream (an integer handle such as
        torch.cuda.current_stream().cuda_stream); None restores a private stream."""
        if hip_stream_handle is None:
            self._check(self._lib.lrvb_ctx_set_stream(self._h, None, 0))
        else:       # 0 is a valid handle: the legacy default stream
            self._check(self._lib.lrvb_ctx_set_stream(self._h, ctypes.c_void_p(int(hip_stream_handle)), 1))

    def wait_stream(self, hip_stream_handle):
        """The context's stream waits for everything queued so far on the caller's stream (an integer handle; 0 / None =
        the legacy default stream): call before passing device pointers produced there (include/lrvb_hip.h, "stream
        ordering")."""
        self._check(self._lib.lrvb_ctx_wait_stream(self._h, ctypes.c_void_p(int(hip_stream_handle or 0))))

    def stream_wait(self, hip_stream_handle):
        """The caller's stream waits for everything this context has queued: call before consuming `_dev` results there."""
        self._check(self._lib.lrvb_stream_wait_ctx(self._h, ctypes.c_void_p(int(hip_stream_handle or 0))))

    # -- packing ------------------------------------------------------------------------
    def constrain(self, free):
        f = _hip.as_f64(free).ravel()
        out = np.empty(self.V)
        self._check(self._lib.lrvb_constrain(self._h, _hip.ptr(f), f.size, _hip.ptr(out), out.size))
        return out

    def unconstrain(self, vec):
        v = _hip.as_f64(vec).ravel()
        out = np.empty(self.D)
        self._check(self._lib.lrvb_unconstrain(self._h, _hip.ptr(v), v.size, _hip.ptr(out), out.size))
        return out

    def free_to_vector_jac(self, free):
        f = _hip.as_f64(free).ravel()
        if f.size != self.D:
            raise ValueError('Wrong size for free vector.  Expected {}, got {}'.format(self.D, f.size))
        out = np.empty((self.V, self.D))
        self._check(self._lib.lrvb_free_to_vector_jac(self._h, _hip.ptr(f), f.size, _hip.ptr(out)))
        return out

    def free_hessian_from_vector(self, free, g_vec, H_vec):
        f, g, H = _hip.as_f64(free).ravel(), _hip.as_f64(g_vec).ravel(), _hip.as_f64(H_vec)
        if f.size != self.D or g.size != self.V or H.shape != (self.V, self.V):
            raise ValueError('Wrong sizes for free_hessian_from_vector')
        out = np.empty((self.D, self.D))
        self._check(self._lib.lrvb_free_hessian_from_vector(self._h, _hip.ptr(f), _hip.ptr(g), _hip.ptr(H), _hip.ptr(out)))
        return out

    # -- objective ----------------------------------------------------------------------
    def _n(self, is_free):
        return self.D if is_free else self.V

    def value(self, x, is_free=True):
        x = _hip.as_f64(x).ravel()
        out = np.empty(1)
        fn = self._lib.lrvb_value if is_free else self._lib.lrvb_value_vec
        self._check(fn(self._h, _hip.ptr(x), x.size, _hip.ptr(out)))
        return float(out[0])

    def grad(self, x, is_free=True):
        x = _hip.as_f64(x).ravel()
        g = np.empty(self._n(is_free))
        fn = self._lib.lrvb_grad if is_free else self._lib.lrvb_grad_vec
        self._check(fn(self._h, _hip.ptr(x), x.size, None, _hip.ptr(g)))
        return g

    def hessian(self, x, is_free=True):
        x = _hip.as_f64(x).ravel()
        n = self._n(is_free)
        H = np.empty((n, n))
        fn = self._lib.lrvb_hessian if is_free else self._lib.lrvb_hessian_vec
        self._check(fn(self._h, _hip.ptr(x), x.size, _hip.ptr(H), n))
        return H

    def hvp(self, x, v, is_free=True):
        x, v = _hip.as_f64(x).ravel(), _hip.as_f64(v).ravel()
        if v.size != x.size:
            raise ValueError('Wrong size for the vector of a Hessian-vector product')
        out = np.empty(self._n(is_free))
        fn = self._lib.lrvb_hvp if is_free else self._lib.lrvb_hvp_vec
        self._check(fn(self._h, _hip.ptr(x), _hip.ptr(v), x.size, _hip.ptr(out)))
        return out

    def obs_grad(self, x, n0=0, n1=None, is_free=True):
        x = _hip.as_f64(x).ravel()
        n1 = self.n_obs if n1 is None else n1
        G = np.empty((max(n1 - n0, 0), self._n(is_free)))
        fn = self._lib.lrvb_obs_grad if is_free else self._lib.lrvb_obs_grad_vec
        self._check(fn(self._h, _hip.ptr(x), x.size, n0, n1, _hip.ptr(G)))
        return G

    def obs_loss(self, x, n0=0, n1=None, is_free=True):
        """l(y_n, z_n) for rows n0..n1 at the point x: d f / d w_n."""
        x = _hip.as_f64(x).ravel()
        n1 = self.n_obs if n1 is None else n1
        out = np.empty(max(n1 - n0, 0))
        self._check(self._lib.lrvb_obs_loss(self._h, _hip.ptr(x), x.size, 1 if is_free else 0, n0, n1, _hip.ptr(out)))
        return out

    def gh_logistic(self, z_mean, z_sd, gh_x, gh_w, order=0):
        """Gauss-Hermite value of E log(1 + e^z), z ~ N(z_mean, z_sd^2), per element (LRVB/Modeling.py:36-52), and
        with order 1 / 2 the first (n x 2: mean, sd) and second (n x 3: mean-mean, mean-sd, sd-sd) derivatives of that sum."""
        zm, zs = _hip.as_f64(z_mean), _hip.as_f64(z_sd)
        if zm.shape != zs.shape:
            raise ValueError('z_mean and z_sd must have one shape')
        gx, gw = _hip.as_f64(gh_x).ravel(), _hip.as_f64(gh_w).ravel()
        if gx.size != gw.size:
            raise ValueError('as many quadrature weights as nodes')
        n = zm.size
        val = np.empty(n)
        d1 = np.empty((n, 2)) if order >= 1 else None
        d2 = np.empty((n, 3)) if order >= 2 else None
        self._check(self._lib.lrvb_gh_logistic(self._h, n, _hip.ptr(zm.ravel()), _hip.ptr(zs.ravel()), _hip.ptr(gx), _hip.ptr(gw),
                                              gx.size, int(order), _hip.ptr(val), _hip.ptr(d1), _hip.ptr(d2)))
        out = [val.reshape(zm.shape)]
        if order >= 1:
            out.append(d1.reshape(zm.shape + (2,)))
        if order >= 2:
            out.append(d2.reshape(zm.shape + (3,)))
        return out[0] if order == 0 else tuple(out)

    def logitnormal_terms(self, mean, var, gh_x, gh_w, want_grad=True, want_hess=True):
        """Data term of logistic regression under q(beta_j) = N(mean_j, var_j) in the coordinates (mean, var):
        value, gradient (2 P) and the Hessian blocks (H_mm, H_mv, H_vv), each P x P (lrvb_logitnormal_terms)."""
        m, v = _hip.as_f64(mean).ravel(), _hip.as_f64(var).ravel()
        gx, gw = _hip.as_f64(gh_x).ravel(), _hip.as_f64(gh_w).ravel()
        P = self.n_cols
        if m.size != P or v.size != P or gx.size != gw.size:
            raise ValueError('expected mean and var of length {} and as many weights as nodes'.format(P))
        val = np.empty(1)
        g = np.empty(2 * P) if want_grad else None
        Hb = np.empty((3, P, P)) if want_hess else None
        self._check(self._lib.lrvb_logitnormal_terms(self._h, _hip.ptr(m), _hip.ptr(v), P, _hip.ptr(gx), _hip.ptr(gw), gx.size,
                                                    _hip.ptr(val), _hip.ptr(g), _hip.ptr(Hb)))
        return float(val[0]), g, (None if Hb is None else (Hb[0], Hb[1], Hb[2]))

    def obs_influence(self, x, moment_jac, n0=0, n1=None, is_free=True):
        """d moments / d weights for observations n0..n1 ((n1 - n0) x Q), from the resident factor."""
        x = _hip.as_f64(x).ravel()
        M = _hip.as_f64(moment_jac)
        if M.ndim != 2 or M.shape[1] != self._n(is_free):
            raise ValueError('moment Jacobian must have {} columns'.format(self._n(is_free)))
        n1 = self.n_obs if n1 is None else n1
        out = np.empty((max(n1 - n0, 0), M.shape[0]))
        fn = self._lib.lrvb_obs_influence if is_free else self._lib.lrvb_obs_influence_vec
        self._check(fn(self._h, _hip.ptr(x), x.size, _hip.ptr(M), M.shape[0], n0, n1, _hip.ptr(out)))
        return out

    # -- vector-coordinate Hessian assembled on the device from small host blocks ---------------------
    # The assembly is RECORDED and sent as one call (lrvb_hvec_program): a begin / add / ... / finish sequence used to cost a
    # binding call, an upload and a launch per block -- more host time than the kernels of the small configurations need.
    # `hvec_begin(immediate=True)` keeps the call-per-block entry points of the C ABI (same result; tests/test_gpu_parity.py).
    def hvec_begin(self, immediate=False):
        self._hv_ops = None
        if immediate:
            self._check(self._lib.lrvb_hvec_begin(self._h))
            return
        self._hv_ops, self._hv_data, self._hv_len, self._hv_seen = [], [], 0, {}

    def _hv_put(self, arr):
        """Position of an operand in the program's data.  The operand is COPIED at record time (the caller may reuse its
        array for the next block, as the call-per-block entry points allow); an operand with the same contents as one
        already recorded travels once."""
        flat = np.array(arr, dtype=np.float64).ravel()          # private copy
        key = (flat.size, flat.tobytes()) if flat.size <= 8192 else None
        if key is not None and key in self._hv_seen:
            return self._hv_seen[key]
        off = self._hv_len
        self._hv_data.append(flat)
        self._hv_len += flat.size
        if key is not None:
            self._hv_seen[key] = off
        return off

    def hvec_add_block(self, block, row_off, col_off, mirror=False):
        B = _hip.as_f64(block)
        B = B.reshape(B.shape[0], -1) if B.ndim > 1 else B.reshape(-1, 1)
        if self._hv_ops is None:
            self._check(self._lib.lrvb_hvec_add_block(self._h, _hip.ptr(B), B.shape[0], B.shape[1], int(row_off), int(col_off), int(bool(mirror))))
            return
        if row_off < 0 or col_off < 0 or row_off + B.shape[0] > self.V or col_off + B.shape[1] > self.V or (mirror and row_off == col_off):
            raise ValueError('block [{}+{}, {}+{}) outside the {} x {} matrix (or a mirrored block on the diagonal)'.format(
                row_off, B.shape[0], col_off, B.shape[1], self.V, self.V))
        self._hv_ops.append((0, self._hv_put(B), B.shape[0], B.shape[1], int(row_off), int(col_off), int(bool(mirror)), 0))

    def hvec_add_indexed(self, block, rows, cols):
        """H[rows[a], cols[b]] += block[a, b] (lrvb_hvec_add_indexed)."""
        B = _hip.as_f64(block)
        r = np.ascontiguousarray(rows, dtype=np.int64).ravel()
        cidx = np.ascontiguousarray(cols, dtype=np.int64).ravel()
        if B.shape != (r.size, cidx.size):
            raise ValueError('block must be len(rows) x len(cols)')
        if np.unique(r).size != r.size or np.unique(cidx).size != cidx.size:
            raise ValueError('an index list names an element twice (the device adds the entries in parallel)')
        if self._hv_ops is None:
            self._check(self._lib.lrvb_hvec_add_indexed(self._h, _hip.ptr(B), r.size, cidx.size,
                                                       r.ctypes.data_as(ctypes.c_void_p), cidx.ctypes.data_as(ctypes.c_void_p)))
            return
        if r.size == 0 or cidx.size == 0 or r.min() < 0 or r.max() >= self.V or cidx.min() < 0 or cidx.max() >= self.V:
            raise ValueError('index outside [0, {})'.format(self.V))
        pack = np.concatenate([B.ravel(), r.astype(np.float64), cidx.astype(np.float64)])     # block, then the two index lists
        self._hv_ops.append((1, self._hv_put(pack), r.size, cidx.size, 0, 0, 0, 0))

    def hvec_add_symkron(self, A, B, coef, row_off, col_off, mirror=False):
        A, B = _hip.as_f64(A), _hip.as_f64(B)
        if A.ndim != 2 or A.shape[0] != A.shape[1] or A.shape != B.shape:
            raise ValueError('expected two square matrices of the same order')
        if self._hv_ops is None:
            self._check(self._lib.lrvb_hvec_add_symkron(self._h, _hip.ptr(A), _hip.ptr(B), A.shape[0], float(coef),
                                                       int(row_off), int(col_off), int(bool(mirror))))
            return
        m = A.shape[0] * (A.shape[0] + 1) // 2
        if row_off < 0 or col_off < 0 or row_off + m > self.V or col_off + m > self.V or (mirror and row_off == col_off):
            raise ValueError('Kronecker block of order {} at ({}, {}) outside the {} x {} matrix'.format(m, row_off, col_off, self.V, self.V))
        cpos = self._hv_put(np.array([float(coef)]))
        self._hv_ops.append((2, self._hv_put(A), A.shape[0], cpos, int(row_off), int(col_off), int(bool(mirror)), self._hv_put(B)))

    def hvec_finish(self, x, g_vec, is_free=True, want_host=True):
        x = _hip.as_f64(x).ravel()
        g = _hip.as_f64(g_vec).ravel()
        n = self._n(is_free)
        out = np.empty((n, n)) if want_host else None
        if self._hv_ops is None:
            self._check(self._lib.lrvb_hvec_finish(self._h, _hip.ptr(x), x.size, int(bool(is_free)), _hip.ptr(g), _hip.ptr(out)))
            return out
        ops = np.array(self._hv_ops, dtype=np.int64).reshape(-1, 8)
        data = np.concatenate(self._hv_data) if self._hv_data else np.zeros(0)
        self._hv_ops = None
        self._hv_seen = {}
        self._check(self._lib.lrvb_hvec_program(self._h, ops.ctypes.data_as(ctypes.c_void_p), ops.shape[0], _hip.ptr(data), data.size,
                                               _hip.ptr(x), x.size, int(bool(is_free)), _hip.ptr(g), _hip.ptr(out)))
        return out

    def cross_hessian_tilt(self, free):
        f = _hip.as_f64(free).ravel()
        C = np.empty((self.D, self.V))
        self._check(self._lib.lrvb_cross_hessian_tilt(self._h, _hip.ptr(f), f.size, _hip.ptr(C)))
        return C

    def hyper_size(self, kind):
        n = ctypes.c_int64(0)
        self._check(self._lib.lrvb_hyper_size(self._h, int(kind), ctypes.byref(n)))
        return n.value

    def cross_hessian_hyper(self, kind, x, is_free=True):
        """d2 f / d x d eps^T for the hyper-parameter `kind` (_hip.HYPER_*) in its vector coordinates: (n, Ph)."""
        x = _hip.as_f64(x).ravel()
        Ph = self.hyper_size(kind)
        C = np.empty((self._n(is_free), Ph))
        self._check(self._lib.lrvb_cross_hessian_hyper(self._h, int(kind), _hip.ptr(x), x.size, 1 if is_free else 0, _hip.ptr(C), Ph))
        return C

    def hyper_grad(self, kind, x, is_free=True):
        """d f / d eps for the hyper-parameter `kind` in its vector coordinates: (Ph,)."""
        x = _hip.as_f64(x).ravel()
        Ph = self.hyper_size(kind)
        g = np.empty(Ph)
        self._check(self._lib.lrvb_hyper_grad(self._h, int(kind), _hip.ptr(x), x.size, 1 if is_free else 0, _hip.ptr(g), Ph))
        return g

    def jac_t_matmul(self, free, B):
        """J(free)^T B on the device for a host matrix B (V x Q): the chain rule of a vector-coordinate cross Hessian."""
        f, B = _hip.as_f64(free).ravel(), _hip.as_f64(B)
        B2 = B.reshape(B.shape[0], -1)
        if f.size != self.D or B2.shape[0] != self.V:
            raise ValueError('expected a free vector of length {} and a matrix with {} rows'.format(self.D, self.V))
        out = np.empty((self.D, B2.shape[1]))
        self._check(self._lib.lrvb_jac_t_matmul(self._h, _hip.ptr(f), f.size, _hip.ptr(B2), B2.shape[1], _hip.ptr(out)))
        return out

    def gram(self, free):
        f = _hip.as_f64(free).ravel()
        G = np.empty((self.D, self.D))
        self._check(self._lib.lrvb_gram(self._h, _hip.ptr(f), f.size, _hip.ptr(G), self.D))
        return G

    # -- objectives quadratic in the data -------------------------------------------------
    def weighted_gram(self, with_sum=False):
        """S = Z^T diag(w) Z; with_sum also returns W = sum w, formed on the device and (under a reduce hook) summed over
        the ranks in the same reduction as S."""
        S = np.empty((self.n_cols, self.n_cols))
        if not with_sum:
            self._check(self._lib.lrvb_weighted_gram(self._h, _hip.ptr(S), self.n_cols))
            return S
        W = np.empty(1)
        self._check(self._lib.lrvb_weighted_gram_sum(self._h, _hip.ptr(S), self.n_cols, _hip.ptr(W)))
        return S, float(W[0])

    def obs_quadform(self, M, c=None, n0=0, n1=None):
        M = _hip.as_f64(M)
        K = M.shape[0]
        if M.shape[1:] != (self.n_cols, self.n_cols):
            raise ValueError('M must be K x {0} x {0}'.format(self.n_cols))
        c = None if c is None else _hip.as_f64(c).ravel()
        n1 = self.n_obs if n1 is None else n1
        out = np.empty((max(n1 - n0, 0), K))
        self._check(self._lib.lrvb_obs_quadform(self._h, _hip.ptr(M), _hip.ptr(c), K, n0, n1, _hip.ptr(out)))
        return out

    def mixture_rows(self, K, theta_z, lam, want_grad=True, want_schur=True):
        """theta_z None: the simplex logits of the previous call, still resident on the device."""
        tz, lam = (None if theta_z is None else _hip.as_f64(theta_z).ravel()), _hip.as_f64(lam)
        V = self.n_cols
        if (tz is not None and tz.size != self.n_obs * (K - 1)) or lam.shape != (V + 1, K):
            raise ValueError('expected theta_z with {} entries and Lam of shape {}'.format(self.n_obs * (K - 1), (V + 1, K)))
        val2 = np.empty(2)
        gfree = np.empty((self.n_obs, K - 1)) if want_grad else None
        S64 = np.empty((64, 64))
        R = np.empty(((V + 1) ** 2, K * K)) if want_schur else None
        self._check(self._lib.lrvb_mixture_rows(self._h, int(K), _hip.ptr(tz), _hip.ptr(lam), _hip.ptr(val2),
                                               _hip.ptr(gfree), _hip.ptr(S64), _hip.ptr(R)))
        return val2, gfree, S64, R

    def mixture_schur(self, K, q, R, jlam, hgg, scale=None, diag_add=None):
        """H = diag(scale) hgg diag(scale) + diag(diag_add) - sym(jlam^T Rm jlam) on the device (lrvb_mixture_schur);
        R=None uses the operand the last mixture_rows call left on the device."""
        n = int(K) * int(q)
        jlam, hgg = _hip.as_f64(jlam), _hip.as_f64(hgg)
        if jlam.shape != (n, n) or hgg.shape != (n, n):
            raise ValueError('expected {0} x {0} matrices'.format(n))
        if R is not None:
            R = _hip.as_f64(R)
            if R.shape != (q * q, K * K):
                raise ValueError('expected R of shape {}'.format((q * q, K * K)))
        scale = None if scale is None else _hip.as_f64(scale).ravel()
        diag_add = None if diag_add is None else _hip.as_f64(diag_add).ravel()
        for v in (scale, diag_add):
            if v is not None and v.size != n:
                raise ValueError('expected vectors of length {}'.format(n))
        out = np.empty((n, n))
        self._check(self._lib.lrvb_mixture_schur(self._h, int(K), int(q), _hip.ptr(R), _hip.ptr(jlam), _hip.ptr(hgg),
                                                _hip.ptr(scale), _hip.ptr(diag_add), _hip.ptr(out)))
        return out

    def set_groups(self, gid, n_groups):
        g = np.ascontiguousarray(gid, dtype=np.int32).ravel()
        self._check(self._lib.lrvb_set_groups(self._h, g.ctypes.data_as(ctypes.c_void_p), g.size, int(n_groups)))
        self.n_groups = int(n_groups)

    def group_sums(self):
        out = np.empty((self.n_groups, self.n_cols + 1))
        self._check(self._lib.lrvb_group_sums(self._h, _hip.ptr(out)))
        return out

    def grouped_stats(self, want_S=True, want_gs=False):
        """[S | group sums] on the device in one buffer (lrvb_grouped_stats): summed over the ranks once, resident for
        `lmm_group_terms`.  Returns (S or None, group sums or None) -- whichever host copies were asked for."""
        S = np.empty((self.n_cols, self.n_cols)) if want_S else None
        gs = np.empty((self.n_groups, self.n_cols + 1)) if want_gs else None
        self._check(self._lib.lrvb_grouped_stats(self._h, _hip.ptr(S), _hip.ptr(gs)))
        return S, gs

    def lmm_group_terms(self, par, f_local):
        """Elimination of the 2 G local parameters of the hierarchical model from the resident grouped statistics
        (lrvb_lmm_group_terms): returns (sums (128,), M (p + 5, p + 5))."""
        par, fl = _hip.as_f64(par).ravel(), _hip.as_f64(f_local).ravel()
        R = self.n_cols - 1 + 5
        out = np.empty(128 + R * R)
        self._check(self._lib.lrvb_lmm_group_terms(self._h, _hip.ptr(par), par.size, _hip.ptr(fl), fl.size, _hip.ptr(out)))
        return out[:128], out[128:].reshape(R, R)

    def mvnreg_hessian(self, free, hp, idx, want_value=False, want_host=True):
        """A whole Hessian build of the MVNParam regression as one call (lrvb_mvnreg_hessian): statistics, closed forms,
        Kronecker block and free conversion on the device, no copy back inside.  Returns (H or None, value or None)."""
        f, hp = _hip.as_f64(free).ravel(), _hip.as_f64(hp).ravel()
        ix = np.ascontiguousarray(idx, dtype=np.int32).ravel()
        out = np.empty((self.D, self.D)) if want_host else None
        val = np.empty(1) if want_value else None
        self._check(self._lib.lrvb_mvnreg_hessian(self._h, _hip.ptr(f), f.size, _hip.ptr(hp), hp.size, ix.ctypes.data_as(ctypes.c_void_p),
                                                 _hip.ptr(val), _hip.ptr(out)))
        return out, (None if val is None else float(val[0]))

    def lmm_global_hessian(self, gctx, free, hp, idx, info_lb, want_sums=False, want_host=True):
        """The Schur complement of the hierarchical model's arrow Hessian onto its global block as one call
        (lrvb_lmm_global_hessian): `self` holds the rows and groups, `gctx` the packing of the global parameters and, afterwards,
        the result.  Returns (H or None, sums (128,) or None)."""
        f, hp = _hip.as_f64(free).ravel(), _hip.as_f64(hp).ravel()
        ix = np.ascontiguousarray(idx, dtype=np.int32).ravel()
        out = np.empty((gctx.D, gctx.D)) if want_host else None
        sums = np.empty(128) if want_sums else None
        self._check(self._lib.lrvb_lmm_global_hessian(self._h, gctx._h, _hip.ptr(f), f.size, _hip.ptr(hp), hp.size,
                                                     ix.ctypes.data_as(ctypes.c_void_p), float(info_lb), _hip.ptr(sums), _hip.ptr(out)))
        return out, sums

    def mixture_stats(self, K, theta_z, lam, want_schur=True):
        """`mixture_rows` without the per-row gradient and with the Schur operand left on the device (summed over the
        ranks there): returns (val2, S64)."""
        tz, lam = (None if theta_z is None else _hip.as_f64(theta_z).ravel()), _hip.as_f64(lam)
        V = self.n_cols
        if (tz is not None and tz.size != self.n_obs * (K - 1)) or lam.shape != (V + 1, K):
            raise ValueError('expected theta_z with {} entries and Lam of shape {}'.format(self.n_obs * (K - 1), (V + 1, K)))
        val2, S64 = np.empty(2), np.empty((64, 64))
        self._check(self._lib.lrvb_mixture_stats(self._h, int(K), _hip.ptr(tz), _hip.ptr(lam), 1 if want_schur else 0,
                                                _hip.ptr(val2), _hip.ptr(S64)))
        return val2, S64

    def mixture_schur_dirichlet(self, K, q, dl_diag, dl_const, h_diag, h_const, scale, diag_add, want_host=True):
        """The Schur complement of `mixture_schur` with d Lam / d alpha and the global Hessian block generated on the
        device from their diagonals and per-Dirichlet constants (lrvb_mixture_schur_dirichlet); the result stays on the
        device for `chol_factor_last`, and comes back only when want_host."""
        n = int(K) * int(q)
        vecs = np.concatenate([_hip.as_f64(v).ravel() for v in (dl_diag, h_diag, scale, diag_add)])
        consts = np.concatenate([_hip.as_f64(v).ravel() for v in (dl_const, h_const)])
        if vecs.size != 4 * n or consts.size != 2 * (K + 1):
            raise ValueError('expected four vectors of length {} and two of length {}'.format(n, K + 1))
        out = np.empty((n, n)) if want_host else None
        self._check(self._lib.lrvb_mixture_schur_dirichlet(self._h, int(K), int(q), _hip.ptr(vecs), _hip.ptr(consts), _hip.ptr(out)))
        return out

    def quadform_gram(self, M, c, free):
        M, c, f = _hip.as_f64(M), _hip.as_f64(c).ravel(), _hip.as_f64(free).ravel()
        if M.shape != (self.V, self.n_cols, self.n_cols) or c.size != self.V:
            raise ValueError('expected M of shape ({0}, {1}, {1}) and c of length {0}'.format(self.V, self.n_cols))
        out = np.empty((self.D, self.D))
        self._check(self._lib.lrvb_quadform_gram(self._h, _hip.ptr(M), _hip.ptr(c), self.V, _hip.ptr(f),
                                                _hip.ptr(out), self.D))
        return out

    def wishart_gram(self, d, offsets, nu, m, v, c, free, want_host=True):
        """G^T G of the Wishart + MVN model with the per-coordinate matrices generated on the device (lrvb_wishart_gram);
        want_host=False leaves the result in HBM (for `chol_factor_last`) and returns None."""
        m, v, c, f = _hip.as_f64(m).ravel(), _hip.as_f64(v), _hip.as_f64(c).ravel(), _hip.as_f64(free).ravel()
        if m.size != d or v.shape != (d, d) or c.size != self.V or f.size != self.D:
            raise ValueError('expected m of length {0}, v of shape ({0}, {0}), c of length {1} and a free vector of length {2}'.format(d, self.V, self.D))
        offs = np.ascontiguousarray(offsets, dtype=np.int64).ravel()
        out = np.empty((self.D, self.D)) if want_host else None
        self._check(self._lib.lrvb_wishart_gram(self._h, int(d), offs.ctypes.data_as(ctypes.c_void_p), float(nu), _hip.ptr(m), _hip.ptr(v),
                                               _hip.ptr(c), _hip.ptr(f), _hip.ptr(out), self.D))
        return out

    def cg_solve_matrix(self, H, b, x0=None, Minv=None, tol=1e-8, maxiter=0):
        b = _hip.as_f64(b).ravel()
        D = b.size
        H = None if H is None else _hip.as_f64(H)
        x0 = None if x0 is None else _hip.as_f64(x0).ravel()
        Minv = None if Minv is None else _hip.as_f64(Minv)
        x = np.empty(D)
        info = ctypes.c_int(0)
        iters = ctypes.c_int64(0)
        self._check(self._lib.lrvb_cg_solve_matrix(self._h, _hip.ptr(H), _hip.ptr(b), _hip.ptr(x0), _hip.ptr(Minv),
                                                  float(tol), int(maxiter), D, _hip.ptr(x),
                                                  ctypes.byref(info), ctypes.byref(iters)))
        return x, info.value, iters.value

    # -- solves ---------------------------------------------------------------------------
    def chol_factor(self, H):
        H = _hip.as_f64(H)
        if H.ndim != 2 or H.shape[0] != H.shape[1]:
            raise ValueError('expected a square matrix')
        self._check(self._lib.lrvb_chol_factor(self._h, _hip.ptr(H), H.shape[0]))
        self.chol_token += 1

    def chol_factor_last(self):
        self._check(self._lib.lrvb_chol_factor_last(self._h))
        self.chol_token += 1

    def chol_solve(self, B):
        B = _hip.as_f64(B)
        B2 = B.reshape(B.shape[0], -1)
        X = np.empty_like(B2)
        self._check(self._lib.lrvb_chol_solve(self._h, _hip.ptr(B2), B2.shape[0], B2.shape[1], _hip.ptr(X)))
        return X.reshape(B.shape)

    def lrvb_cov(self, M):
        M = _hip.as_f64(M)
        Q, D = M.shape
        out = np.empty((Q, Q))
        self._check(self._lib.lrvb_lrvb_cov(self._h, _hip.ptr(M), Q, D, _hip.ptr(out)))
        return out

    def cg_solve(self, free, b, x0=None, Minv=None, tol=1e-8, maxiter=0):
        f, b = _hip.as_f64(free).ravel(), _hip.as_f64(b).ravel()
        if b.size != self.D:
            raise ValueError('Wrong size for the right-hand side.  Expected {}, got {}'.format(self.D, b.size))
        x0 = None if x0 is None else _hip.as_f64(x0).ravel()
        Minv = None if Minv is None else _hip.as_f64(Minv)
        x = np.empty(self.D)
        info = ctypes.c_int(0)
        iters = ctypes.c_int64(0)
        self._check(self._lib.lrvb_cg_solve(self._h, _hip.ptr(f), _hip.ptr(b), _hip.ptr(x0), _hip.ptr(Minv),
                                           float(tol), int(maxiter), f.size, _hip.ptr(x),
                                           ctypes.byref(info), ctypes.byref(iters)))
        return x, info.value, iters.value

    def dk_grad_vec(self, eta, U=None, w_override=None, include_quad=True):
        """D^j g [u_1 .. u_j] in vector coordinates (lrvb_dk_grad_vec): rows of U are the directions, U None or empty
        gives the gradient itself; w_override evaluates with other observation weights (a direction in weight space)."""
        eta = _hip.as_f64(eta).ravel()
        if eta.size != self.V:
            raise ValueError('Wrong size for the vector.  Expected {}, got {}'.format(self.V, eta.size))
        order = 0
        if U is not None:
            U = _hip.as_f64(U).reshape(-1, self.V) if np.size(U) else None
            order = 0 if U is None else U.shape[0]
        if w_override is not None:
            w_override = _hip.as_f64(w_override).ravel()
            if w_override.size != self.n_obs:
                raise ValueError('expected {} weights'.format(self.n_obs))
        out = np.empty(self.V)
        self._check(self._lib.lrvb_dk_grad_vec(self._h, _hip.ptr(eta), eta.size, int(order), _hip.ptr(U), _hip.ptr(w_override),
                                              1 if include_quad else 0, _hip.ptr(out)))
        return out

    def minimize_trust_ncg(self, y0, precond=None, gtol=1e-6, maxiter=0, initial_trust_radius=1.0,
                           max_trust_radius=1000.0, eta=0.15):
        """Trust-region Newton-CG on the device (lrvb_minimize_trust_ncg).  The iterate y lives in the
        optimiser's coordinates, x = precond @ y (precond None = identity).  Returns (y, x, info dict)."""
        y0 = _hip.as_f64(y0).ravel()
        if y0.size != self.D:
            raise ValueError('Wrong size for the starting point.  Expected {}, got {}'.format(self.D, y0.size))
        A = None if precond is None else _hip.as_f64(precond)
        if A is not None and A.shape != (self.D, self.D):
            raise ValueError('preconditioner must be {0} x {0}'.format(self.D))
        y, x = np.empty(self.D), np.empty(self.D)
        res = _hip.OptResult()
        self._check(self._lib.lrvb_minimize_trust_ncg(self._h, _hip.ptr(y0), y0.size, _hip.ptr(A), float(gtol), int(maxiter),
                                                     float(initial_trust_radius), float(max_trust_radius), float(eta),
                                                     _hip.ptr(y), _hip.ptr(x), ctypes.byref(res)))
        info = dict(fun=res.fun, jac_mag=res.jac_mag, trust_radius=res.trust_radius, status=res.status, nit=res.nit,
                    nfev=res.nfev, njev=res.njev, nhev=res.nhev, nbuild=res.nbuild)
        return y, x, info

    def cg_solve_multi(self, free, B, X0=None, Minv=None, tol=1e-8, maxiter=0):
        """Rows of B are right-hand sides; returns (X (Q x D), info (Q,), iterations (Q,))."""
        f, B = _hip.as_f64(free).ravel(), _hip.as_f64(B)
        if B.ndim != 2 or B.shape[1] != self.D:
            raise ValueError('Wrong size for the right-hand sides.  Expected (Q, {}), got {}'.format(self.D, B.shape))
        Q = B.shape[0]
        X0 = None if X0 is None else _hip.as_f64(X0).reshape(Q, self.D)
        Minv = None if Minv is None else _hip.as_f64(Minv)
        X = np.empty((Q, self.D))
        info = np.zeros(Q, dtype=np.int32)
        iters = np.zeros(Q, dtype=np.int64)
        self._check(self._lib.lrvb_cg_solve_multi(self._h, _hip.ptr(f), _hip.ptr(B), _hip.ptr(X0), _hip.ptr(Minv),
                                                 float(tol), int(maxiter), f.size, Q, _hip.ptr(X),
                                                 info.ctypes.data_as(ctypes.c_void_p), iters.ctypes.data_as(ctypes.c_void_p)))
        return X, info, iters

    # -- device-resident / multi-GPU -------------------------------------------------------
    def stats_size(self):
        n = ctypes.c_int64(0)
        self._check(self._lib.lrvb_stats_size(self._h, ctypes.byref(n)))
        return n.value

    def hessian_partial_dev(self, free_ptr, stats_ptr):
        self._check(self._lib.lrvb_hessian_partial_dev(self._h, ctypes.c_void_p(free_ptr), ctypes.c_void_p(stats_ptr)))

    def hessian_finish_dev(self, free_ptr, stats_ptr, H_ptr, ld):
        self._check(self._lib.lrvb_hessian_finish_dev(self._h, ctypes.c_void_p(free_ptr), ctypes.c_void_p(stats_ptr),
                                                     ctypes.c_void_p(H_ptr), ld))

    def hessian_dev(self, free_ptr, H_ptr, ld):
        self._check(self._lib.lrvb_hessian_dev(self._h, ctypes.c_void_p(free_ptr), ctypes.c_void_p(H_ptr), ld))

    def hvp_dev(self, free_ptr, v_ptr, out_ptr):
        self._check(self._lib.lrvb_hvp_dev(self._h, ctypes.c_void_p(free_ptr), ctypes.c_void_p(v_ptr), ctypes.c_void_p(out_ptr)))

    def gram_dev(self, free_ptr, G_ptr, ld):
        self._check(self._lib.lrvb_gram_dev(self._h, ctypes.c_void_p(free_ptr), ctypes.c_void_p(G_ptr), ld))

    def chol_factor_dev(self, H_ptr, D, ld):
        self.chol_token += 1
        self._check(self._lib.lrvb_chol_factor_dev(self._h, ctypes.c_void_p(H_ptr), D, ld))

    def chol_solve_dev(self, B_ptr, D, nrhs):
        self._check(self._lib.lrvb_chol_solve_dev(self._h, ctypes.c_void_p(B_ptr), D, nrhs))

    def lrvb_cov_dev(self, M_ptr, Q, D, cov_ptr):
        self._check(self._lib.lrvb_lrvb_cov_dev(self._h, ctypes.c_void_p(M_ptr), Q, D, ctypes.c_void_p(cov_ptr)))

    # -- profiling ---------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self._lib.lrvb_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.lrvb_profile_reset(self._h))

    def profile_get(self):
        p = _hip.Prof()
        self._check(self._lib.lrvb_profile_get(self._h, ctypes.byref(p)))
        return {name: getattr(p, name) for name, _ in _hip.Prof._fields_}


def refuse_double_reduction(ctx, flat):
    """`set_reduced_stats(...)` installs statistics that the CALLER summed over the ranks.  With a sum-over-ranks hook on the
    context the statistics calls already return global sums: summing those again multiplies them by the world size."""
    if flat is not None and getattr(ctx, 'has_reduce_hook', False):
        raise RuntimeError('this context reduces inside the library (reduce hook / in-library communicator): its statistics '
                           'are already sums over all ranks -- do not all-reduce and install them again')


class DeclaredHypers(object):
    """Registry of the hyper-parameters a declared objective exposes to `TwoParameterObjective` /
    `ParametricSensitivityLinearApproximation` / `ParametricSensitivityTaylorExpansion` (the reference accepts ANY
    parameter object as `hyper_par`, LRVB/ModelSensitivity.py:555-612; a declared objective lists the ones it can
    differentiate): attribute `<name>_par` per hyper-parameter, `hyper_pars` (name -> parameter), `hyper_kind(par)`."""

    def _declare_hyper(self, name, par):
        if not hasattr(self, '_hyper_names'):
            self._hyper_names = []
        if name not in self._hyper_names:
            self._hyper_names.append(name)
        setattr(self, name + '_par', par)
        return par

    @property
    def hyper_pars(self):
        """name -> parameter object of every declared hyper-parameter (an attribute `<name>_par` that the caller replaced
        by another parameter object of the same size is honoured)."""
        return {name: getattr(self, name + '_par') for name in getattr(self, '_hyper_names', [])
                if getattr(self, name + '_par', None) is not None}

    def hyper_kind(self, hyper_par):
        for name, par in self.hyper_pars.items():
            if hyper_par is par:
                return name
        raise NotImplementedError(
            'the second parameter must be one of this objective\'s declared hyper-parameters ({}); any other '
            'hyper-parameter would need tracing of a Python closure, which the device path cannot do'.format(
                ', '.join(n + '_par' for n in getattr(self, '_hyper_names', []))))

    def _hyper_vec(self, name):
        return np.asarray(getattr(self, name + '_par').get_vector(), dtype=np.float64).ravel()

    def _hyper_derived(self, name, make):
        """make(vector value of the hyper-parameter), computed once per value (version stamp) of the parameter object."""
        par = getattr(self, name + '_par')
        version = getattr(par, 'version', None)
        memo = self.__dict__.setdefault('_hyper_memo', {})
        hit = memo.get(name)
        if version is not None and hit is not None and hit[0] == (id(par), version):
            return hit[1]
        val = make(self._hyper_vec(name))
        if version is not None:
            memo[name] = ((id(par), version), val)
        return val

    def _hyper_state_key(self, skip=('weights',)):
        """Cheap identity of the current values of the hyper-parameters (for result memos): version stamps where the
        parameter has them, contents otherwise."""
        key = []
        for name, par in self.hyper_pars.items():
            if name in skip:
                continue
            version = getattr(par, 'version', None)
            key.append((name, version if version is not None else np.asarray(par.get_vector(), dtype=np.float64).tobytes()))
        return tuple(key)


class DeviceObjective(DeclaredHypers):
    """A declared objective bound to a parameter object `par` (any object with the packing
    protocol and `layout_blocks()`).

    Callable like the reference's `fun`: `objective_fun()` returns the value at the current
    state of `par`.  Extra positional / keyword arguments given to the Objective methods are
    forwarded to `scale_fun(*argv, **argk)`, whose result multiplies the quadratic term -- the
    declared counterpart of the keyword pass-through at LRVB/test_objectives.py:161-217.

    Hyper-parameters (for TwoParameterObjective / ParametricSensitivityLinearApproximation):
      `weights_par`  HyperVectorParam('weights', N) holding the per-observation weights (a VectorParam whose value is a
                     private read-only copy with a version stamp: the resident copy is checked in O(1))
                     (Example.ipynb:254, 425-441);
      `tilt_par`     VectorParam('tilt', V) holding the linear tilt b
                     (the `hyper_param @ theta` term of LRVB/test_model_sensitivity.py:56-66).
    Their current values are pushed to the device before every evaluation.
    """
    _lrvb_device_functor = True

    def __init__(self, par, x=None, y=None, loss=None, glm_param=None, lik_info=1.0,
                 quad_A=None, quad_m=None, quad_b=None, scale_fun=None, weights=None, device=0):
        self.par = par
        blocks = par.layout_blocks()
        glm_off = 0
        if loss is not None and loss != 'none':
            x = _hip.as_f64(x)
            if x.ndim != 2:
                raise ValueError('x must be a 2-d array (observations x columns)')
            n_obs, n_cols = x.shape
            if glm_param is not None:
                glm_off = par.vector_indices_dict[glm_param].start
                if len(par.vector_indices_dict[glm_param]) != n_cols:
                    raise ValueError('Wrong size for the coefficient parameter {}.  Expected {}, got {}'.format(
                        glm_param, n_cols, len(par.vector_indices_dict[glm_param])))
        else:
            n_obs = n_cols = 0
            loss = None
        V = par.vector_size()
        if quad_A is None:
            quad_kind = _hip.QUAD_NONE if (quad_b is None and quad_m is None) else _hip.QUAD_DIAG
            if quad_kind == _hip.QUAD_DIAG:
                quad_A = np.zeros(V)
        else:
            quad_A = _hip.as_f64(quad_A)
            quad_kind = _hip.QUAD_DIAG if quad_A.ndim == 1 else _hip.QUAD_DENSE
        self.ctx = DeviceContext(blocks, loss=loss, n_obs=n_obs, n_cols=n_cols, glm_off=glm_off,
                                 lik_info=lik_info, quad_kind=quad_kind, device=device)
        if self.ctx.D != par.free_size() or self.ctx.V != par.vector_size():
            raise ValueError('layout_blocks() of the parameter disagrees with its free/vector sizes')
        self.n_obs = n_obs
        self.scale_fun = scale_fun
        self._quad_kind = quad_kind
        self._hyper_names = []
        self._res = {}
        self.weights_par = self.tilt_par = None
        self.prior_mean_par = self.prior_info_par = self.quad_scale_par = self.lik_info_par = None
        if loss is not None:
            self.ctx.set_data(_hip.SLOT_X, x)
            self.ctx.set_data(_hip.SLOT_Y, _hip.as_f64(y).ravel())
            w0 = np.ones(n_obs) if weights is None else _hip.as_f64(weights).ravel().copy()
            self._declare_hyper('weights', HyperVectorParam('weights', n_obs, val=w0))
            if _LOSSES[loss] == _hip.LOSS_GAUSSIAN:
                self._declare_hyper('lik_info', HyperVectorParam('lik_info', 1, lb=0.0, val=np.array([float(lik_info)])))
        if quad_kind != _hip.QUAD_NONE:
            b0 = np.zeros(V) if quad_b is None else _hip.as_f64(quad_b).ravel().copy()
            m0 = np.zeros(V) if quad_m is None else _hip.as_f64(quad_m).ravel().copy()
            self._declare_hyper('tilt', HyperVectorParam('tilt', V, val=b0))
            self._declare_hyper('prior_mean', HyperVectorParam('prior_mean', V, val=m0))
            if quad_kind == _hip.QUAD_DIAG:
                self._declare_hyper('prior_info', HyperVectorParam('prior_info', V, val=quad_A.ravel()))
            else:       # the symmetric matrix in the reference's vector form: its row-major lower triangle
                self._declare_hyper('prior_info', HyperVectorParam('prior_info', V * (V + 1) // 2, val=sym_to_vech(quad_A)))
            self._declare_hyper('quad_scale', HyperVectorParam('quad_scale', 1, val=np.ones(1)))
        self._push_state()

    # ---- declared hyper-parameters ------------------------------------------------------------
    def _declare_hyper(self, name, par):
        self._res[name] = ResidentVector()
        return DeclaredHypers._declare_hyper(self, name, par)

    # ---- state pushed before every evaluation ------------------------------------------
    def _push_state(self):
        for name in self._hyper_names:
            par = getattr(self, name + '_par')
            if par is None:
                continue
            v = self._res[name].changed(par)                   # O(1) for the objective's own HyperVectorParams
            if v is None:
                continue
            if name == 'weights':
                self.ctx.set_weights(v)
            elif name == 'tilt':
                self.ctx.set_data(_hip.SLOT_QUAD_B, v)
            elif name == 'prior_mean':
                self.ctx.set_data(_hip.SLOT_QUAD_M, v)
            elif name == 'prior_info':
                self.ctx.set_data(_hip.SLOT_QUAD_A, v if self._quad_kind == _hip.QUAD_DIAG else vech_to_sym(v))
            elif name == 'lik_info':
                self.ctx.set_lik_info(float(np.ravel(v)[0]))
            # quad_scale: read by _scale() at every evaluation

    def _scale(self, argv=(), argk=None):
        """The multiplier of the quadratic term: the `quad_scale` hyper-parameter times scale_fun(*argv, **argk)."""
        s = 1.0 if self.quad_scale_par is None else float(np.ravel(self.quad_scale_par.get_vector())[0])
        if self.scale_fun is not None:
            s *= float(self.scale_fun(*argv, **(argk or {})))
        elif argv or argk:
            raise TypeError('this objective takes no extra arguments (no scale_fun declared)')
        return s

    def _push(self, argv=(), argk=None):
        self._push_state()
        s = self._scale(argv, argk)
        if s != self.ctx.quad_scale:
            self.ctx.set_quad_scale(s)

    # ---- the reference's `fun` protocol --------------------------------------------------
    def __call__(self, *argv, **argk):
        self._push(argv, argk)
        return self.ctx.value(np.asarray(self.par.get_free(), dtype=np.float64), True)

    # ---- derivative protocol used by Objective / TwoParameterObjective ---------------------
    def value(self, x, is_free, *argv, **argk):
        self._push(argv, argk)
        return self.ctx.value(x, is_free)

    def grad(self, x, is_free, *argv, **argk):
        self._push(argv, argk)
        return self.ctx.grad(x, is_free)

    def hessian(self, x, is_free, *argv, **argk):
        self._push(argv, argk)
        return self.ctx.hessian(x, is_free)

    def hvp(self, x, v, is_free, *argv, **argk):
        self._push(argv, argk)
        return self.ctx.hvp(x, v, is_free)

    def jacobian(self, x, is_free, *argv, **argk):
        # scalar objective: the Jacobian is the gradient (shape (D,)), as autograd.jacobian gives
        return self.grad(x, is_free, *argv, **argk)

    _HYPER_KINDS = {'tilt': _hip.HYPER_TILT, 'prior_mean': _hip.HYPER_QUAD_M, 'prior_info': _hip.HYPER_QUAD_A,
                    'quad_scale': _hip.HYPER_QUAD_SCALE, 'lik_info': _hip.HYPER_LIK_INFO}

    def cross_hessian(self, hyper_par, val1, val1_is_free, *argv, **argk):
        """d2 f / d par1 d hyper^T with hyper in VECTOR coordinates; shape (n1, hyper size).  Closed forms on the device
        (csrc/k_hyper.hip); the weights' cross Hessian is the per-observation gradient matrix."""
        kind = self.hyper_kind(hyper_par)
        self._push(argv, argk)
        if kind == 'weights':
            return np.ascontiguousarray(self.ctx.obs_grad(val1, 0, self.n_obs, val1_is_free).T)
        return self.ctx.cross_hessian_hyper(self._HYPER_KINDS[kind], val1, val1_is_free)

    def hyper_grad(self, hyper_par, val1, val1_is_free, *argv, **argk):
        """d f / d hyper with hyper in VECTOR coordinates (TwoParameterObjective.fun_grad2,
        LRVB/SparseObjectives.py:381-387): the per-observation loss l(y_n, z_n) for the weights (one skinny pass over X),
        closed forms in eta - m for the others (csrc/k_hyper.hip)."""
        kind = self.hyper_kind(hyper_par)
        self._push(argv, argk)
        if kind == 'weights':
            return self.ctx.obs_loss(val1, 0, self.n_obs, val1_is_free)
        return self.ctx.hyper_grad(self._HYPER_KINDS[kind], val1, val1_is_free)

    def hyper_direction_vec(self, hyper_par, eta, U, eps_dir):
        """D_eta^r [ d g_eta / d eps [eps_dir] ] [u_1 .. u_r] in vector coordinates (rows of U are the u's; None / empty:
        r = 0): the mixed directional derivatives `ParametricSensitivityTaylorExpansion` needs (the reference nests
        autograd JVPs for them, LRVB/ModelSensitivity.py:38-62).  The gradient is linear in every declared
        hyper-parameter; the O(N) kinds (weights, lik_info) are `lrvb_dk_grad_vec` passes, the others V-sized closed forms."""
        kind = self.hyper_kind(hyper_par)
        self._push()
        eta = _hip.as_f64(eta).ravel()
        U = None if U is None or np.size(U) == 0 else _hip.as_f64(U).reshape(-1, self.ctx.V)
        r = 0 if U is None else U.shape[0]
        eps_dir = _hip.as_f64(eps_dir).ravel()
        if kind == 'weights':
            return self.ctx.dk_grad_vec(eta, U, eps_dir, False)
        if kind == 'lik_info':                               # the data gradient is tau times a tau-free sum
            tau = float(np.ravel(self.lik_info_par.get_vector())[0])
            return (eps_dir[0] / tau) * self.ctx.dk_grad_vec(eta, U, None, False)
        s = self.ctx.quad_scale
        zero = np.zeros(self.ctx.V)
        m = np.asarray(self.prior_mean_par.get_vector(), dtype=np.float64)
        a = np.asarray(self.prior_info_par.get_vector(), dtype=np.float64)
        diag = self._quad_kind == _hip.QUAD_DIAG
        A_apply = (lambda v: a * v) if diag else (lambda v: vech_to_sym(a) @ v)
        if kind == 'tilt':
            return s * eps_dir if r == 0 else zero
        if kind == 'prior_mean':
            return -s * A_apply(eps_dir) if r == 0 else zero
        if kind == 'prior_info':
            dA = (lambda v: eps_dir * v) if diag else (lambda v: vech_to_sym(eps_dir) @ v)
            return s * dA(eta - m) if r == 0 else (s * dA(U[0]) if r == 1 else zero)
        if kind == 'quad_scale':
            b = np.asarray(self.tilt_par.get_vector(), dtype=np.float64)
            return eps_dir[0] * (A_apply(eta - m) + b) if r == 0 else (eps_dir[0] * A_apply(U[0]) if r == 1 else zero)
        raise NotImplementedError(kind)

    def gram(self, free_val):
        self._push_state()
        return self.ctx.gram(free_val)


def GLMObjective(par, x, y, loss='gaussian', glm_param=None, lik_info=1.0, prior_info=None,
                 prior_mean=None, weights=None, device=0):
    """Dense-design GLM-type objective (the headline workload of BASELINE.json):
    sum_n w_n loss(y_n, x_n . beta) + 1/2 (eta - prior_mean)^T diag(prior_info) (eta - prior_mean)."""
    V = par.vector_size()
    A = None
    if prior_info is not None:
        A = np.full(V, float(prior_info)) if np.isscalar(prior_info) else _hip.as_f64(prior_info)
    return DeviceObjective(par, x=x, y=y, loss=loss, glm_param=glm_param, lik_info=lik_info,
                           quad_A=A, quad_m=prior_mean, weights=weights, device=device)


def QuadraticObjective(par, A, m=None, b=None, scale_fun=None, device=0):
    """s (1/2 (eta-m)^T A (eta-m) + b^T eta): the closed-form models of the reference's tests
    (LRVB/test_objectives.py:14-57, 161-217; LRVB/test_model_sensitivity.py:36-88;
    LRVB/test_optimization_utils.py:10-24)."""
    return DeviceObjective(par, quad_A=A, quad_m=m, quad_b=b, scale_fun=scale_fun, device=device)


class LinearMoments(object):
    """Vector-valued functor m(eta) = B eta (B: Q x V), or a sub-parameter's vector when built
    with `select=name` -- the `summary` of Example.ipynb:380-396.  Its free-coordinate Jacobian
    B J(theta) is what `Objective.fun_free_jacobian` returns for a vector-valued `fun`
    (LRVB/SparseObjectives.py:160-162)."""
    _lrvb_device_functor = True

    def __init__(self, par, B=None, select=None, device=0):
        self.par = par
        V = par.vector_size()
        if select is not None:
            idx = np.asarray(list(par.vector_indices_dict[select]))
            B = np.zeros((idx.size, V))
            B[np.arange(idx.size), idx] = 1.0
        self.B = _hip.as_f64(B)
        if self.B.shape[1] != V:
            raise ValueError('B must have {} columns'.format(V))
        self._layout = DeviceContext(par.layout_blocks(), quad_kind=_hip.QUAD_DIAG, device=device)

    def __call__(self):
        return self.B @ np.asarray(self.par.get_vector(), dtype=np.float64)

    def value(self, x, is_free):
        eta = self._layout.constrain(x) if is_free else _hip.as_f64(x).ravel()
        return self.B @ eta

    def jacobian(self, x, is_free):
        if not is_free:
            return self.B.copy()
        return self.B @ self._layout.free_to_vector_jac(x)

    def grad(self, *a, **k):
        raise TypeError('gradient of a vector-valued functor; use the Jacobian')

    hessian = hvp = grad
