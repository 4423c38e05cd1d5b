"""ctypes binding of liblrvb_hip.so (the C ABI declared in include/lrvb_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception
is raised.  Status codes are translated into the exception types the reference raises for the
same mistake (ValueError for wrong sizes / bounds: LRVB/ParameterDictionary.py:56-60,
LRVB/Parameters.py:15-28; numpy.linalg.LinAlgError for a failed Cholesky, as scipy's cho_factor
at LRVB/ModelSensitivity.py:594).
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblrvb_hip.so')

OK, ERR_INVALID, ERR_SIZE, ERR_HIP, ERR_STATE, ERR_NOT_POSDEF, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
BLOCK_BOX, BLOCK_PSD, BLOCK_SIMPLEX = 0, 1, 2
LOSS_NONE, LOSS_GAUSSIAN, LOSS_LOGISTIC, LOSS_POISSON, LOSS_DATA_ONLY = 0, 1, 2, 3, 4
QUAD_NONE, QUAD_DIAG, QUAD_DENSE = 0, 1, 2
SLOT_X, SLOT_Y, SLOT_QUAD_A, SLOT_QUAD_M, SLOT_QUAD_B = 0, 1, 2, 3, 4
HYPER_TILT, HYPER_QUAD_M, HYPER_QUAD_A, HYPER_QUAD_SCALE, HYPER_LIK_INFO = 0, 1, 2, 3, 4

c_double_p = ctypes.POINTER(ctypes.c_double)
c_i64 = ctypes.c_int64


class BlockDesc(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('free_off', c_i64), ('vec_off', c_i64), ('free_size', c_i64), ('vec_size', c_i64),
                ('dim0', c_i64), ('dim1', c_i64), ('lb', ctypes.c_double), ('ub', ctypes.c_double)]


class ModelDesc(ctypes.Structure):
    _fields_ = [('n_blocks', ctypes.c_int32), ('loss', ctypes.c_int32),
                ('blocks', ctypes.POINTER(BlockDesc)),
                ('n_obs', c_i64), ('n_cols', c_i64), ('glm_off', c_i64),
                ('lik_info', ctypes.c_double), ('quad_kind', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class Prof(ctypes.Structure):
    _fields_ = [('wsyrk_ms', ctypes.c_double), ('wsyrk_calls', c_i64),
                ('wsyrk_flops', ctypes.c_double), ('wsyrk_bytes', ctypes.c_double),
                ('pass_ms', ctypes.c_double), ('pass_calls', c_i64), ('pass_bytes', ctypes.c_double),
                ('build_ms', ctypes.c_double), ('build_calls', c_i64),
                ('reduce_ms', ctypes.c_double), ('reduce_calls', c_i64)]


# name -> argtypes; every function returns int except the two noted.  This table is also what
# tests/test_abi_symbols.py checks against include/lrvb_hip.h.
_VP = ctypes.c_void_p
_SIGNATURES = {
    'lrvb_version': [],
    'lrvb_device_count': [ctypes.POINTER(ctypes.c_int)],
    'lrvb_ctx_create': [ctypes.POINTER(_VP), ctypes.c_int, ctypes.POINTER(ModelDesc)],
    'lrvb_ctx_destroy': [_VP],
    'lrvb_ctx_sync': [_VP],
    'lrvb_ctx_set_stream': [_VP, _VP, ctypes.c_int],
    'lrvb_ctx_wait_stream': [_VP, _VP],
    'lrvb_stream_wait_ctx': [_VP, _VP],
    'lrvb_ctx_sizes': [_VP, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)],
    'lrvb_set_data': [_VP, ctypes.c_int, _VP, c_i64, c_i64],
    'lrvb_set_data_dev': [_VP, ctypes.c_int, _VP, c_i64, c_i64],
    'lrvb_set_weights': [_VP, _VP, c_i64],
    'lrvb_set_weights_dev': [_VP, _VP, c_i64],
    'lrvb_set_quad_scale': [_VP, ctypes.c_double],
    'lrvb_set_lik_info': [_VP, ctypes.c_double],
    'lrvb_constrain': [_VP, _VP, c_i64, _VP, c_i64],
    'lrvb_unconstrain': [_VP, _VP, c_i64, _VP, c_i64],
    'lrvb_free_to_vector_jac': [_VP, _VP, c_i64, _VP],
    'lrvb_free_hessian_from_vector': [_VP, _VP, _VP, _VP, _VP],
    'lrvb_value': [_VP, _VP, c_i64, _VP],
    'lrvb_grad': [_VP, _VP, c_i64, _VP, _VP],
    'lrvb_hessian': [_VP, _VP, c_i64, _VP, c_i64],
    'lrvb_hvp': [_VP, _VP, _VP, c_i64, _VP],
    'lrvb_value_vec': [_VP, _VP, c_i64, _VP],
    'lrvb_grad_vec': [_VP, _VP, c_i64, _VP, _VP],
    'lrvb_hessian_vec': [_VP, _VP, c_i64, _VP, c_i64],
    'lrvb_hvp_vec': [_VP, _VP, _VP, c_i64, _VP],
    'lrvb_obs_grad': [_VP, _VP, c_i64, c_i64, c_i64, _VP],
    'lrvb_obs_grad_vec': [_VP, _VP, c_i64, c_i64, c_i64, _VP],
    'lrvb_obs_loss': [_VP, _VP, c_i64, ctypes.c_int, c_i64, c_i64, _VP],
    'lrvb_gh_logistic': [_VP, c_i64, _VP, _VP, _VP, _VP, ctypes.c_int32, ctypes.c_int32, _VP, _VP, _VP],
    'lrvb_logitnormal_terms': [_VP, _VP, _VP, c_i64, _VP, _VP, ctypes.c_int32, _VP, _VP, _VP],
    'lrvb_hvec_begin': [_VP],
    'lrvb_hvec_add_block': [_VP, _VP, c_i64, c_i64, c_i64, c_i64, ctypes.c_int],
    'lrvb_hvec_add_indexed': [_VP, _VP, c_i64, c_i64, _VP, _VP],
    'lrvb_hvec_add_symkron': [_VP, _VP, _VP, c_i64, ctypes.c_double, c_i64, c_i64, ctypes.c_int],
    'lrvb_hvec_finish': [_VP, _VP, c_i64, ctypes.c_int, _VP, _VP],
    'lrvb_hvec_program': [_VP, _VP, c_i64, _VP, c_i64, _VP, c_i64, ctypes.c_int, _VP, _VP],
    'lrvb_obs_influence': [_VP, _VP, c_i64, _VP, c_i64, c_i64, c_i64, _VP],
    'lrvb_obs_influence_vec': [_VP, _VP, c_i64, _VP, c_i64, c_i64, c_i64, _VP],
    'lrvb_cross_hessian_tilt': [_VP, _VP, c_i64, _VP],
    'lrvb_hyper_size': [_VP, ctypes.c_int, ctypes.POINTER(c_i64)],
    'lrvb_cross_hessian_hyper': [_VP, ctypes.c_int, _VP, c_i64, ctypes.c_int, _VP, c_i64],
    'lrvb_hyper_grad': [_VP, ctypes.c_int, _VP, c_i64, ctypes.c_int, _VP, c_i64],
    'lrvb_jac_t_matmul': [_VP, _VP, c_i64, _VP, c_i64, _VP],
    'lrvb_gram': [_VP, _VP, c_i64, _VP, c_i64],
    'lrvb_weighted_gram': [_VP, _VP, c_i64],
    'lrvb_weighted_gram_sum': [_VP, _VP, c_i64, _VP],
    'lrvb_obs_quadform': [_VP, _VP, _VP, c_i64, c_i64, c_i64, _VP],
    'lrvb_mixture_rows': [_VP, ctypes.c_int32, _VP, _VP, _VP, _VP, _VP, _VP],
    'lrvb_dk_grad_vec': [_VP, _VP, ctypes.c_int64, ctypes.c_int32, _VP, _VP, ctypes.c_int32, _VP],
    'lrvb_minimize_trust_ncg': [_VP, _VP, ctypes.c_int64, _VP, ctypes.c_double, ctypes.c_int64, ctypes.c_double,
                                ctypes.c_double, ctypes.c_double, _VP, _VP, _VP],
    'lrvb_mixture_schur': [_VP, ctypes.c_int32, ctypes.c_int32, _VP, _VP, _VP, _VP, _VP, _VP],
    'lrvb_mixture_stats': [_VP, ctypes.c_int32, _VP, _VP, ctypes.c_int32, _VP, _VP],
    'lrvb_mixture_schur_dirichlet': [_VP, ctypes.c_int32, ctypes.c_int32, _VP, _VP, _VP],
    'lrvb_set_groups': [_VP, _VP, c_i64, c_i64],
    'lrvb_group_sums': [_VP, _VP],
    'lrvb_grouped_stats': [_VP, _VP, _VP],
    'lrvb_lmm_group_terms': [_VP, _VP, c_i64, _VP, c_i64, _VP],
    'lrvb_mvnreg_hessian': [_VP, _VP, c_i64, _VP, c_i64, _VP, _VP, _VP],
    'lrvb_lmm_global_hessian': [_VP, _VP, _VP, c_i64, _VP, c_i64, _VP, ctypes.c_double, _VP, _VP],
    'lrvb_quadform_gram': [_VP, _VP, _VP, c_i64, _VP, _VP, c_i64],
    'lrvb_wishart_gram': [_VP, c_i64, _VP, ctypes.c_double, _VP, _VP, _VP, _VP, _VP, c_i64],
    'lrvb_cg_solve_matrix': [_VP, _VP, _VP, _VP, _VP, ctypes.c_double, c_i64, c_i64, _VP,
                             ctypes.POINTER(ctypes.c_int), ctypes.POINTER(c_i64)],
    'lrvb_chol_factor': [_VP, _VP, c_i64],
    'lrvb_chol_factor_last': [_VP],
    'lrvb_chol_solve': [_VP, _VP, c_i64, c_i64, _VP],
    'lrvb_lrvb_cov': [_VP, _VP, c_i64, c_i64, _VP],
    'lrvb_chol_factor_dev': [_VP, _VP, c_i64, c_i64],
    'lrvb_chol_solve_dev': [_VP, _VP, c_i64, c_i64],
    'lrvb_lrvb_cov_dev': [_VP, _VP, c_i64, c_i64, _VP],
    'lrvb_cg_solve': [_VP, _VP, _VP, _VP, _VP, ctypes.c_double, c_i64, c_i64, _VP,
                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(c_i64)],
    'lrvb_cg_solve_multi': [_VP, _VP, _VP, _VP, _VP, ctypes.c_double, c_i64, c_i64, c_i64, _VP, _VP, _VP],
    'lrvb_stats_size': [_VP, ctypes.POINTER(c_i64)],
    'lrvb_hessian_partial_dev': [_VP, _VP, _VP],
    'lrvb_hessian_finish_dev': [_VP, _VP, _VP, _VP, c_i64],
    'lrvb_hessian_dev': [_VP, _VP, _VP, c_i64],
    'lrvb_hvp_dev': [_VP, _VP, _VP, _VP],
    'lrvb_gram_dev': [_VP, _VP, _VP, c_i64],
    'lrvb_profile_enable': [_VP, ctypes.c_int],
    'lrvb_profile_get': [_VP, ctypes.POINTER(Prof)],
    'lrvb_profile_reset': [_VP],
    'lrvb_set_tuning': [_VP, ctypes.c_int, ctypes.c_int],
    'lrvb_set_reduce_hook': [_VP, _VP, _VP],
    'lrvb_comm_unique_id': [_VP],
    'lrvb_comm_init': [_VP, ctypes.c_int, ctypes.c_int, _VP],
    'lrvb_comm_destroy': [_VP],
    'lrvb_allreduce_hessian': [_VP, _VP, c_i64],
}

# lrvb_reduce_fn of include/lrvb_hip.h: int fn(void* user, double* buf_dev, int64_t n, void* hip_stream)
REDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

class OptResult(ctypes.Structure):
    """lrvb_opt_result of include/lrvb_hip.h."""
    _fields_ = [('fun', ctypes.c_double), ('jac_mag', ctypes.c_double), ('trust_radius', ctypes.c_double),
                ('status', ctypes.c_int32), ('nit', ctypes.c_int32),
                ('nfev', ctypes.c_int32), ('njev', ctypes.c_int32), ('nhev', ctypes.c_int32), ('nbuild', ctypes.c_int32)]


_lib = None


def _map_shared_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 (soname libamdhip64.so.7, next to its
    own HSA runtime) and this library is linked against the same soname, which the dynamic linker resolves ONCE per
    process: to whichever copy is mapped first.  If torch is already imported its copy is mapped and there is nothing to
    do.  Otherwise the copy torch WOULD load is mapped here, by path, before liblrvb_hip.so -- so that a later
    `import torch` finds its own runtime already in place instead of the system one (mixing torch's HSA runtime with
    the system HIP runtime made torch report "No HIP GPUs are available").  torch itself is not imported; without an
    installed torch the system ROCm copy is used.  LRVB_NO_TORCH_PRELOAD=1 skips this."""
    import importlib.util
    import sys
    if os.environ.get('LRVB_NO_TORCH_PRELOAD', '0') == '1' or 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)


def load():
    """Loads the shared library (once).  Raises OSError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _map_shared_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise OSError(
            'liblrvb_hip.so not found at {}: build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (or csrc/build.sh).  There is no CPU fallback.'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.lrvb_last_error.argtypes = []
    lib.lrvb_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def device_count():
    n = ctypes.c_int(0)
    check(load().lrvb_device_count(ctypes.byref(n)))
    return n.value


def last_error():
    return load().lrvb_last_error().decode('utf-8', 'replace')


def check(status):
    if status == OK:
        return
    msg = last_error()
    if status in (ERR_INVALID, ERR_SIZE):
        raise ValueError(msg)
    if status == ERR_NOT_POSDEF:
        raise np.linalg.LinAlgError(msg)
    if status == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError('liblrvb_hip: {} (status {})'.format(msg, status))


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    """Host pointer of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(ctypes.c_void_p)


# ---- host closed forms -------------------------------------------------------------------------------------------
# The N-independent algebra of the model classes (a few hundred small products per evaluation) runs through numpy's BLAS.
# With the library's default pool -- one thread per hardware thread of a 128+-core host -- every small product pays for
# waking the pool: the LMM step of bench.py --config c4 took 40-68 ms with it and 10.6 ms with 16 threads.
try:
    from threadpoolctl import ThreadpoolController as _ThreadpoolController
except Exception:                                    # not installed: run with the pool as it is
    _ThreadpoolController = None
HOST_BLAS_THREADS = int(os.environ.get('LRVB_HOST_BLAS_THREADS', '16'))
_blas_controller = None
_blas_depth = 0


def host_blas(fn):
    """Decorator: run the method with the BLAS pool limited to HOST_BLAS_THREADS (0 = leave the pool alone).  The
    controller (one scan of the loaded libraries) is built once; nested decorated calls do not re-enter it."""
    if _ThreadpoolController is None:
        return fn
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        global _blas_controller, _blas_depth
        if HOST_BLAS_THREADS <= 0 or _blas_depth > 0:
            return fn(*a, **k)
        if _blas_controller is None:
            _blas_controller = _ThreadpoolController()
        _blas_depth += 1
        try:
            with _blas_controller.limit(limits=HOST_BLAS_THREADS, user_api='blas'):
                return fn(*a, **k)
        finally:
            _blas_depth -= 1
    return wrapped
