"""Objectives that are quadratic in the data: the conjugate-normal family of models whose
per-observation term is  l_n(eta) = 1/2 z_n^T Q(eta) z_n + c(eta),  so that

    f(eta) = sum_n w_n l_n(eta) + R(eta) = 1/2 tr(Q(eta) S(w)) + W c(eta) + R(eta),
    S(w) = sum_n w_n z_n z_n^T   (device: weighted-SYRK kernel),   W = sum_n w_n.

The O(N) arithmetic -- the weighted sufficient statistics S(w) for every new weight vector, and the
N x V per-observation gradient matrix for weight sensitivities -- runs on the GPU
(`lrvb_weighted_gram`, `lrvb_obs_quadform`); the packing Jacobians / free-coordinate conversion run
on the GPU too (`lrvb_free_to_vector_jac`, `lrvb_free_hessian_from_vector`).  What stays on the
host is the N-independent closed form of Q, c, R and their first two derivatives in the
constrained vector eta (a few Kronecker products of small matrices).

`NormalRegressionObjective` is the model of the reference's Example.ipynb:247-274
(`-loglik = 1/2 sum_n w_n r_n^T Lambda r_n - 1/2 (sum w) log|Lambda|`, r_n = y_n - beta^T x_n,
beta an ArrayParam with lb = 0, Lambda a PosDefMatrixParam) -- BASELINE.json config 1.
"""
import numpy as np

from . import _hip
from .models import DeviceContext, DeclaredHypers, sym_to_vech, vech_to_sym, refuse_double_reduction
from .packing import VectorParam, HyperVectorParam, ResidentVector


def duplication_matrix(k):
    """Dup (k^2 x k(k+1)/2): vec_rowmajor(A) = Dup @ tril_vector(A) for symmetric A, with the
    lower-triangle vector in the reference's row-major order (LRVB/MatrixParameters.py:16-23)."""
    m = k * (k + 1) // 2
    D = np.zeros((k * k, m))
    for i in range(k):
        for j in range(k):
            a, b = (i, j) if i >= j else (j, i)
            D[i * k + j, b + a * (a + 1) // 2] = 1.0
    return D


def vech_dupT(A):
    """Dup^T vec(A) for a symmetric A (or a stack A[..., k, k]): its row-major lower triangle with the off-diagonal entries
    doubled -- the gradient with respect to the vector form of a symmetric matrix parameter."""
    k = A.shape[-1]
    r, c = np.tril_indices(k)
    return A[..., r, c] * np.where(r == c, 1.0, 2.0)


# ---- priors as hyper-parameters: closed forms shared by the normal-family models -------------------------------------
# mvn_prior (LRVB/ExponentialFamilies.py:186-189) enters the -ELBO as  1/2 (m - mu0)^T L0 (m - mu0) + 1/2 tr(L0 P),  P the
# variational covariance; gamma_prior (:194-195) as  -(a0 - 1) E log tau + b0 E tau.  Derivatives with respect to
# (mu0, vech(L0)) and (a0, b0) -- what `jacobian(grad_1, argnum=hyper)` (LRVB/SparseObjectives.py:333-339) returns.
def mvn_prior_hyper_grad(kind, dm, P, lam0):
    if kind == 'mean':
        return -lam0 @ dm
    r, c = np.tril_indices(dm.size)
    return (dm[r] * dm[c] + P[r, c]) * np.where(r == c, 0.5, 1.0)


def mvn_prior_hyper_cross(kind, V, ms, ls, dm, P, lam0):
    """(V x Ph) cross block in vector coordinates; ms / ls the index ranges of the mean and of vech(information)."""
    k = dm.size
    if kind == 'mean':
        C = np.zeros((V, k))
        C[ms.start:ms.stop, :] = -lam0
        return C
    r, c = np.tril_indices(k)
    n, cols, off = r.size, np.arange(r.size), r != c
    C = np.zeros((V, n))
    C[ms.start + r, cols] += dm[c]                               # d/dL0_(ij) of L0 (m - mu0)
    C[ms.start + c[off], cols[off]] += dm[r[off]]
    # d/dL0_(ij) of vech-gradient(-1/2 P L0 P): -1/2 P E_(ij) P with E_(ij) = e_i e_j^T + e_j e_i^T (once for i = j)
    Pi, Pj = P[:, r], P[:, c]
    M = -0.5 * np.where(off, 1.0, 0.5)[None, None, :] * (Pi[:, None, :] * Pj[None, :, :] + Pj[:, None, :] * Pi[None, :, :])
    C[ls.start:ls.stop, :] = vech_dupT(np.moveaxis(M, 2, 0)).T
    return C


def gamma_prior_hyper_grad(a, b, special):
    """[d/da0, d/db0] of -(a0 - 1) (psi(a) - log b) + b0 a / b."""
    return np.array([-(special.digamma(a) - np.log(b)), a / b])


def gamma_prior_hyper_cross(V, ia, ib, a, b, special):
    """(V x 2): columns a0, b0."""
    C = np.zeros((V, 2))
    C[ia, 0], C[ib, 0] = -special.polygamma(1, a), 1.0 / b
    C[ia, 1], C[ib, 1] = 1.0 / b, -a / b ** 2
    return C


class QuadraticDataObjective(DeclaredHypers):
    """Base functor.  Subclasses provide the N-independent closed forms:

      _terms(eta, S, W)   -> (value, grad (V,), hess (V, V)) in vector coordinates
      _obs_terms(eta)     -> (M (V, q, q) symmetric, c (V,)):  d l_n / d eta_k = 1/2 z_n^T M_k z_n + c_k
      _prior_hyper(kind, eta, want) -> d f / d eps (Ph,) or d2 f / d eta d eps^T (V, Ph) for the priors they declare
    """
    _lrvb_device_functor = True

    def __init__(self, par, z, weights=None, device=0):
        self.par = par
        z = _hip.as_f64(z)
        self.n_obs, self.q = z.shape
        self.ctx = DeviceContext(par.layout_blocks(), loss='data_only', n_obs=self.n_obs, n_cols=self.q, device=device)
        if self.ctx.D != par.free_size() or self.ctx.V != par.vector_size():
            raise ValueError('layout_blocks() of the parameter disagrees with its free/vector sizes')
        self.ctx.set_data(_hip.SLOT_X, z)
        w0 = np.ones(self.n_obs) if weights is None else _hip.as_f64(weights).ravel().copy()
        self._declare_hyper('weights', HyperVectorParam('weights', self.n_obs, val=w0))
        self.tilt_par = None
        self._w_res = ResidentVector()
        self._S = None
        self._W = None

    # ---- device state ----------------------------------------------------------------------
    def _push_state(self):
        w = self._w_res.changed(self.weights_par)              # O(1) for the objective's own HyperVectorParam
        if w is not None:
            self.ctx.set_weights(w)
            self._S = None

    def _stats(self):
        ext = getattr(self, '_external_stats', None)
        if ext is not None:
            return ext[:-1].reshape(self.q, self.q), float(ext[-1])
        self._push_state()
        if self._S is None or getattr(self, '_S_epoch', None) != self.ctx.hook_epoch:
            # GPU: sum_n w_n z_n z_n^T and sum_n w_n in one buffer -- with a reduce hook on the context (observations sharded
            # over ranks) both arrive summed over all ranks, one reduction.  (A hook installed or removed since the last
            # evaluation makes the cached statistics stale: they key on the context's hook epoch.)
            self._S, self._W = self.ctx.weighted_gram(with_sum=True)
            self._S_epoch = self.ctx.hook_epoch
        return self._S, self._W

    # ---- observations sharded over GPUs: the statistics are sums over rows ---------------------------
    def local_stats(self):
        """[S (q*q) | W]: the statistics every evaluation is a closed form of.  Without a reduce hook these are THIS
        process's rows -- the buffer of the one sum all-reduce per evaluation that a host-side
        exchange (`distributed.allreduce_stats` + `set_reduced_stats`) performs.  With a hook on the context
        (`ShardedObjective`, `native_comm_init`) they are ALREADY the sums over all ranks, reduced on the device inside
        the statistics call; `gram` likewise returns the global G^T G then."""
        self._external_stats = None
        S, W = self._stats()
        return np.concatenate([np.asarray(S, dtype=np.float64).ravel(), [W]])

    def set_reduced_stats(self, flat):
        """Install statistics summed over all shards (None = use this process's own)."""
        refuse_double_reduction(getattr(self, "ctx", None), flat)
        if flat is None:
            self._external_stats = None
            return
        flat = np.asarray(flat, dtype=np.float64).ravel()
        if flat.size != self.q * self.q + 1:
            raise ValueError('expected {} statistics'.format(self.q * self.q + 1))
        self._external_stats = flat.copy()
        self._h_key = None

    def _eta(self, x, is_free):
        x = _hip.as_f64(x).ravel()
        return self.ctx.constrain(x) if is_free else x

    # ---- functor protocol ------------------------------------------------------------------
    def __call__(self):
        return self.value(np.asarray(self.par.get_free(), dtype=np.float64), True)

    def _terms_vg(self, eta, S, W):
        """(value, gradient) only; subclasses whose Hessian blocks are expensive override this."""
        return self._terms(eta, S, W)[:2]

    @_hip.host_blas
    def value(self, x, is_free):
        S, W = self._stats()
        return float(self._terms_vg(self._eta(x, is_free), S, W)[0])

    @_hip.host_blas
    def grad(self, x, is_free):
        S, W = self._stats()
        g = self._terms_vg(self._eta(x, is_free), S, W)[1]
        if not is_free:
            return g
        return self.ctx.free_to_vector_jac(x).T @ g

    jacobian = grad

    @_hip.host_blas
    def hessian(self, x, is_free):
        S, W = self._stats()
        _, g, H = self._terms(self._eta(x, is_free), S, W)
        if not is_free:
            return H
        return self.ctx.free_hessian_from_vector(x, g, H)      # J^T H J + sum_k g_k d2 eta_k, on device

    def _hessian_cached(self, x, is_free):
        self._push_state()
        key = (bool(is_free), np.asarray(x, dtype=np.float64).tobytes(), self._w_res.key, self._hyper_state_key(), self.ctx.hook_epoch)
        if getattr(self, '_h_key', None) != key:
            self._h_val = self.hessian(x, is_free)
            self._h_key = key
        return self._h_val

    def hvp(self, x, v, is_free):
        return self._hessian_cached(x, is_free) @ _hip.as_f64(v).ravel()

    @_hip.host_blas
    def gram(self, free_val):
        """G^T G of the per-observation gradient matrix in free coordinates (Kronecker rows generated
        on chip, contracted on the fp64 matrix cores)."""
        M, c = self._obs_terms(self._eta(free_val, True))
        return self.ctx.quadform_gram(M, c, free_val)

    def cg_solve(self, free_val, b, x0=None, Minv=None, tol=1e-8, maxiter=0):
        """H(free_val)^-1 b by conjugate gradients on the device; the Hessian is assembled once and
        stays resident for further right-hand sides at the same point."""
        H = self._hessian_cached(free_val, True)
        resident = getattr(self, '_cg_key', None) == self._h_key
        out = self.ctx.cg_solve_matrix(None if resident else H, b, x0=x0, Minv=Minv, tol=tol, maxiter=maxiter)
        self._cg_key = self._h_key
        return out

    def _prior_hyper(self, kind, eta, want):
        raise NotImplementedError('this objective declares no prior hyper-parameter `{}`'.format(kind))

    def hyper_grad(self, hyper_par, val1, val1_is_free):
        """d f / d hyper (vector coordinates of the hyper-parameter) for the declared PRIOR hyper-parameters."""
        kind = self.hyper_kind(hyper_par)
        if kind == 'weights':
            raise NotImplementedError('d f / d weights of a quadratic-in-data objective: use the rows of `cross_hessian`')
        return self._prior_hyper(kind, self._eta(val1, val1_is_free), 'grad')

    @_hip.host_blas
    def cross_hessian(self, hyper_par, val1, val1_is_free):
        """d2 f / d par1 d hyper^T.  Weights: (n1, N) -- rows of G come from the device, the chain rule through the packing
        Jacobian is one small product.  Priors: the N-independent closed form in vector coordinates (V x Ph), multiplied
        by J^T on the device (`lrvb_jac_t_matmul`)."""
        kind = self.hyper_kind(hyper_par)
        if kind != 'weights':
            Cv = self._prior_hyper(kind, self._eta(val1, val1_is_free), 'cross')
            return self.ctx.jac_t_matmul(val1, Cv) if val1_is_free else Cv
        self._push_state()
        M, c = self._obs_terms(self._eta(val1, val1_is_free))
        G = self.ctx.obs_quadform(M, c)                          # N x V
        if val1_is_free:
            G = G @ self.ctx.free_to_vector_jac(val1)            # N x D
        return np.ascontiguousarray(G.T)


class NormalRegressionObjective(QuadraticDataObjective):
    """Example.ipynb:247-274.  `par` must contain an array parameter `beta_name` of shape
    (dx, dy) and a positive-definite matrix parameter `lambda_name` of size dy, in that order or
    any other (offsets come from the dictionary)."""

    def __init__(self, par, x, y, beta_name='beta', lambda_name='lambda', weights=None, device=0):
        x, y = _hip.as_f64(x), _hip.as_f64(y)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError('x must be (N, dx) and y (N, dy)')
        self.dx, self.dy = x.shape[1], y.shape[1]
        self._bs = par.vector_indices_dict[beta_name]
        self._ls = par.vector_indices_dict[lambda_name]
        if len(self._bs) != self.dx * self.dy:
            raise ValueError('Wrong size for {}.  Expected {}, got {}'.format(beta_name, self.dx * self.dy, len(self._bs)))
        if len(self._ls) != self.dy * (self.dy + 1) // 2:
            raise ValueError('Wrong size for {}'.format(lambda_name))
        self._dup = duplication_matrix(self.dy)
        super().__init__(par, np.hstack([x, y]), weights=weights, device=device)

    def _unpack(self, eta):
        beta = eta[self._bs.start:self._bs.stop].reshape(self.dx, self.dy)
        lam = (self._dup @ eta[self._ls.start:self._ls.stop]).reshape(self.dy, self.dy)
        return beta, lam

    def _terms(self, eta, S, W):
        dx, dy = self.dx, self.dy
        beta, lam = self._unpack(eta)
        Sxx, Sxy, Syy = S[:dx, :dx], S[:dx, dx:], S[dx:, dx:]
        E = Sxx @ beta - Sxy                                   # dx x dy
        Mres = Syy - beta.T @ Sxy - Sxy.T @ beta + beta.T @ Sxx @ beta     # sum_n w_n r_n r_n^T
        sign, logdet = np.linalg.slogdet(lam)
        if sign <= 0:
            raise ValueError('Matrix is not positive definite')
        P = np.linalg.inv(lam)
        value = 0.5 * np.sum(lam * Mres) - 0.5 * W * logdet
        V = eta.size
        g = np.zeros(V)
        g[self._bs.start:self._bs.stop] = (E @ lam).ravel()
        g[self._ls.start:self._ls.stop] = self._dup.T @ (0.5 * Mres - 0.5 * W * P).ravel()
        H = np.zeros((V, V))
        bs, ls = slice(self._bs.start, self._bs.stop), slice(self._ls.start, self._ls.stop)
        H[bs, bs] = np.kron(Sxx, lam)
        Hbl = np.kron(E, np.eye(dy)) @ self._dup
        H[bs, ls] = Hbl
        H[ls, bs] = Hbl.T
        H[ls, ls] = self._dup.T @ (0.5 * W * np.kron(P, P)) @ self._dup
        return value, g, H

    def _obs_terms(self, eta):
        dx, dy, q = self.dx, self.dy, self.q
        beta, lam = self._unpack(eta)
        B = np.hstack([-beta.T, np.eye(dy)])                  # r_n = B z_n
        P = np.linalg.inv(lam)
        V = eta.size
        M = np.zeros((V, q, q))
        c = np.zeros(V)
        LB = lam @ B
        for a in range(dx):
            for b in range(dy):
                dB = np.zeros((dy, q))
                dB[b, a] = -1.0
                t = dB.T @ LB
                M[self._bs.start + a * dy + b] = t + t.T
        for i in range(dy):
            for j in range(i + 1):
                Eij = np.zeros((dy, dy))
                Eij[i, j] = 1.0
                Eij[j, i] = 1.0
                k = self._ls.start + j + i * (i + 1) // 2
                M[k] = B.T @ Eij @ B
                c[k] = -0.5 * (P[i, j] if i == j else 2.0 * P[i, j])
        return M, c


class MVNRegressionObjective(QuadraticDataObjective):
    """BASELINE.json config 2: conjugate-normal regression y_n = x_n^T beta + eps with
    q(beta) = MVNParam(k) (mean m, information matrix Lambda), q(tau) = GammaParam (shape a, rate b):

      -ELBO = sum_n w_n [ 1/2 E[tau] ((y_n - x_n^T m)^2 + x_n^T Sigma x_n) - 1/2 E[log tau] ]
              - mvn_prior(mu0, Lambda0; m, Sigma) - gamma_prior(a0, b0; E tau, E log tau)
              - multivariate_normal_entropy(Lambda) - gamma_entropy(a, b),         Sigma = Lambda^-1,

    assembled from the reference's building blocks: NormalParams.py:6-23, GammaParams.py:4-16,
    ExponentialFamilies.py:27-35 (entropies), :111-112 (E log tau), :186-195 (priors).  `par` is a
    ModelParamsDict holding an MVNParam named `beta_name` and a GammaParam named `tau_name`.
    Per observation l_n = 1/2 z_n^T Q z_n + c with z = [x; y],
    Q = E[tau] [[m m^T + Sigma, -m], [-m^T, 1]], c = -1/2 E[log tau]."""

    def __init__(self, par, x, y, prior_mean=None, prior_info=None, prior_shape=1.0, prior_rate=1.0,
                 beta_name='beta', tau_name='tau', weights=None, device=0):
        from scipy import special
        self._special = special
        x = _hip.as_f64(x)
        y = _hip.as_f64(y).reshape(-1, 1)
        self.k = k = x.shape[1]
        b0 = par.vector_indices_dict[beta_name].start
        sub = par[beta_name]
        self._ms = range(b0 + sub.vector_indices_dict['mean'].start, b0 + sub.vector_indices_dict['mean'].stop)
        self._ls = range(b0 + sub.vector_indices_dict['info'].start, b0 + sub.vector_indices_dict['info'].stop)
        t0 = par.vector_indices_dict[tau_name].start
        tsub = par[tau_name]
        self._ia = t0 + tsub.vector_indices_dict['shape'].start
        self._ib = t0 + tsub.vector_indices_dict['rate'].start
        if len(self._ms) != k or len(self._ls) != k * (k + 1) // 2:
            raise ValueError('Wrong size for {}.  Expected dimension {}'.format(beta_name, k))
        self._dup = duplication_matrix(k)
        self._names = (beta_name, tau_name)
        super().__init__(par, np.hstack([x, y]), weights=weights, device=device)
        self._declare_priors(prior_mean, prior_info, prior_shape, prior_rate)

    def _declare_priors(self, prior_mean, prior_info, prior_shape, prior_rate):
        """The priors are hyper-parameters (LRVB/ModelSensitivity.py:555-612: prior sensitivity is the defining use of
        the class): prior_mean_par (k), prior_info_par (the symmetric matrix in vector form), prior_shape_par, prior_rate_par."""
        k = self.k
        self._declare_hyper('prior_mean', HyperVectorParam('prior_mean', k, val=np.zeros(k) if prior_mean is None else _hip.as_f64(prior_mean).ravel()))
        self._declare_hyper('prior_info', HyperVectorParam('prior_info', k * (k + 1) // 2,
                                                           val=sym_to_vech(np.eye(k) if prior_info is None else prior_info)))
        self._declare_hyper('prior_shape', HyperVectorParam('prior_shape', 1, lb=0.0, val=np.array([float(prior_shape)])))
        self._declare_hyper('prior_rate', HyperVectorParam('prior_rate', 1, lb=0.0, val=np.array([float(prior_rate)])))

    mu0 = property(lambda self: self._hyper_vec('prior_mean'))
    lam0 = property(lambda self: self._hyper_derived('prior_info', vech_to_sym))
    a0 = property(lambda self: float(self._hyper_vec('prior_shape')[0]))
    b0 = property(lambda self: float(self._hyper_vec('prior_rate')[0]))

    def _unpack(self, eta):
        k = self.k
        m = eta[self._ms.start:self._ms.stop]
        lam = (self._dup @ eta[self._ls.start:self._ls.stop]).reshape(k, k)
        return m, lam, eta[self._ia], eta[self._ib]

    def _prior_hyper(self, kind, eta, want):
        m, lam, a, b = self._unpack(eta)
        if kind in ('prior_mean', 'prior_info'):
            P = np.linalg.inv(lam)
            sub = kind[len('prior_'):]
            if want == 'grad':
                return mvn_prior_hyper_grad(sub, m - self.mu0, P, self.lam0)
            return mvn_prior_hyper_cross(sub, eta.size, self._ms, self._ls, m - self.mu0, P, self.lam0)
        col = {'prior_shape': 0, 'prior_rate': 1}[kind]
        if want == 'grad':
            return gamma_prior_hyper_grad(a, b, self._special)[col:col + 1]
        return gamma_prior_hyper_cross(eta.size, self._ia, self._ib, a, b, self._special)[:, col:col + 1]

    @_hip.host_blas
    def hessian(self, x, is_free):
        """The (k(k+1)/2)^2 block of q(beta)'s information matrix is three Kronecker products of k x k matrices: the device
        writes it from the factors (lrvb_hvec_add_symkron) and converts the assembled matrix to free coordinates; the host
        sends the rest of the vector-coordinate Hessian (its other blocks are O(V k) numbers)."""
        if is_free and getattr(self, '_external_stats', None) is None and not getattr(self, 'stepwise', False) and hasattr(self.ctx, 'mvnreg_hessian') \
                and self._ia == self._ls.stop and self._ib == self._ia + 1 and self._ms.stop == self._ls.start:
            return self.device_hessian(x)[0]
        S, W = self._stats()
        eta = self._eta(x, is_free)
        _, g, H = self._terms(eta, S, W, kron=False)
        G, P = self._kron_factors
        c, l0 = self.ctx, self._ls.start
        c.hvec_begin()
        c.hvec_add_block(H, 0, 0)
        c.hvec_add_symkron(G, P, 0.5, l0, l0)
        c.hvec_add_symkron(P, G, 0.5, l0, l0)
        c.hvec_add_symkron(P, P, -0.5, l0, l0)
        return c.hvec_finish(x if is_free else eta, g, is_free)

    def _host_pack(self, free_val):
        """The theta-only coefficients of the closed forms (csrc/k_lmm.hip, `mvnreg_closed_forms_kernel`): the point is
        constrained by the host parameter objects -- no device round trip -- and P = Lambda^-1, P Lambda0 P, the polygamma
        values and log|Lambda| are formed here; everything that touches the statistics happens on the device."""
        sp, k = self._special, self.k
        beta, tau = self.par[self._names[0]], self.par[self._names[1]]
        fi = self.par.free_indices_dict
        x = np.asarray(free_val, dtype=np.float64)
        beta.set_free(x[fi[self._names[0]].start:fi[self._names[0]].stop])
        tau.set_free(x[fi[self._names[1]].start:fi[self._names[1]].stop])
        m = np.asarray(beta['mean'].get(), dtype=np.float64).ravel()
        lam = np.asarray(beta['info'].get(), dtype=np.float64)
        a, b = float(np.ravel(tau['shape'].get())[0]), float(np.ravel(tau['rate'].get())[0])
        sign, logdet = np.linalg.slogdet(lam)
        if sign <= 0:
            raise ValueError('Matrix is not positive definite')
        P = np.linalg.inv(lam)
        lam0 = self.lam0
        hp = np.zeros(32 + 2 * k + 3 * k * k)
        hp[:9] = [a, b, self.a0, self.b0, sp.digamma(a), sp.polygamma(1, a), sp.polygamma(2, a), sp.gammaln(a), logdet]
        o = 32
        for arr in (m, self.mu0, P, lam0, P @ lam0 @ P):
            hp[o:o + arr.size] = np.ravel(arr)
            o += arr.size
        return hp

    @_hip.host_blas
    def device_hessian(self, free_val, want_value=False, want_host=True):
        """The free-coordinate Hessian (and, if asked, the value) as ONE library call: the weighted Gram of [x | y] and the sum
        of the weights, the closed forms in (m, Lambda, a, b) where the statistics lie, the Kronecker block of Lambda and
        the free conversion -- nothing is copied back inside the call (round 3: statistics to the host, ~40 numpy calls on
        21 x 21 matrices, blocks back up).  want_host=False leaves the matrix in HBM (`ctx.chol_factor_last()`)."""
        self._push_state()
        return self.ctx.mvnreg_hessian(free_val, self._host_pack(free_val), [self._ms.start, self._ls.start, self._ia, self._ib],
                                       want_value=want_value, want_host=want_host)

    def _terms(self, eta, S, W, kron=True):
        sp = self._special
        k = self.k
        m, lam, a, b = self._unpack(eta)
        Sxx, Sxy, Syy = S[:k, :k], S[:k, k], S[k, k]
        sign, logdet = np.linalg.slogdet(lam)
        if sign <= 0:
            raise ValueError('Matrix is not positive definite')
        P = np.linalg.inv(lam)
        e, L = a / b, sp.digamma(a) - np.log(b)
        psi1, psi2 = sp.polygamma(1, a), sp.polygamma(2, a)
        u = Sxx @ m - Sxy
        rss = Syy - 2.0 * m @ Sxy + m @ Sxx @ m
        C = e * Sxx + self.lam0
        dm = m - self.mu0
        entropy_gamma = a - np.log(b) + sp.gammaln(a) + (1.0 - a) * sp.digamma(a)
        value = (0.5 * e * rss + 0.5 * np.sum(C * P) - 0.5 * W * L + 0.5 * dm @ self.lam0 @ dm
                 - (self.a0 - 1.0) * L + self.b0 * e + 0.5 * logdet - entropy_gamma
                 - 0.5 * (k + k * np.log(2.0 * np.pi)))
        V = eta.size
        g = np.zeros(V)
        H = np.zeros((V, V))
        ms, ls = slice(self._ms.start, self._ms.stop), slice(self._ls.start, self._ls.stop)
        ia, ib = self._ia, self._ib
        G = P @ C @ P
        Gs = -0.5 * (P @ Sxx @ P)                              # d g_vecLambda / d e
        f_e = 0.5 * rss + 0.5 * np.sum(Sxx * P) + self.b0
        f_L = -0.5 * W - (self.a0 - 1.0)
        g[ms] = e * u + self.lam0 @ dm
        g[ls] = self._dup.T @ (-0.5 * G + 0.5 * P).ravel()
        g[ia] = f_e / b + f_L * psi1 - (1.0 + (1.0 - a) * psi1)
        g[ib] = -f_e * a / b ** 2 - f_L / b + 1.0 / b
        H[ms, ms] = C
        if kron:                          # dense duplication-matrix algebra: 2 k^6 flops, the reference's shape of the computation
            H[ls, ls] = self._dup.T @ (0.5 * (np.kron(G, P) + np.kron(P, G)) - 0.5 * np.kron(P, P)) @ self._dup
        else:
            self._kron_factors = (G, P)
        H[ia, ia] = f_L * psi2 + psi1 - (1.0 - a) * psi2
        H[ia, ib] = H[ib, ia] = -f_e / b ** 2
        H[ib, ib] = 2.0 * f_e * a / b ** 3 + f_L / b ** 2 - 1.0 / b ** 2
        H[ms, ia] = H[ia, ms] = u / b
        H[ms, ib] = H[ib, ms] = -u * a / b ** 2
        gl = self._dup.T @ Gs.ravel()
        H[ls, ia] = H[ia, ls] = gl / b
        H[ls, ib] = H[ib, ls] = -gl * a / b ** 2
        return value, g, H

    def _obs_terms(self, eta):
        sp = self._special
        k, q = self.k, self.q
        m, lam, a, b = self._unpack(eta)
        P = np.linalg.inv(lam)
        e = a / b
        Q0 = np.zeros((q, q))
        Q0[:k, :k] = np.outer(m, m) + P
        Q0[:k, k] = Q0[k, :k] = -m
        Q0[k, k] = 1.0
        V = eta.size
        M = np.zeros((V, q, q))
        c = np.zeros(V)
        for i in range(k):
            dQ = np.zeros((q, q))
            dQ[i, :k] += m
            dQ[:k, i] += m
            dQ[i, k] = dQ[k, i] = -1.0
            M[self._ms.start + i] = e * dQ
        for i in range(k):
            for j in range(i + 1):
                Eij = np.zeros((k, k))
                Eij[i, j] = Eij[j, i] = 1.0
                dQ = np.zeros((q, q))
                dQ[:k, :k] = -P @ Eij @ P
                M[self._ls.start + j + i * (i + 1) // 2] = e * dQ
        M[self._ia] = Q0 / b
        M[self._ib] = -Q0 * a / b ** 2
        c[self._ia] = -0.5 * sp.polygamma(1, a)
        c[self._ib] = 0.5 / b
        return M, c


class WishartMVNObjective(QuadraticDataObjective):
    """BASELINE.json config 5: full-covariance normal model y_n ~ N(mu, Lambda^-1) with
    q(mu) = MVNParam(d) (mean m, information Lambda_mu) and q(Lambda) = WishartParam(d) (df nu, scale V):

      -ELBO = sum_n w_n [ 1/2 nu ((y_n - m)^T V (y_n - m) + tr(V Sigma_mu)) - 1/2 E log|Lambda| ]
              - mvn_prior(mu0, Lambda0; m, Sigma_mu) - [1/2 (nu0 - d - 1) E log|Lambda| - 1/2 nu tr(W0 V)]
              - multivariate_normal_entropy(Lambda_mu) - wishart_entropy(nu, V),
      E log|Lambda| = psi_d(nu/2) + d log 2 + log|V|,    Sigma_mu = Lambda_mu^-1,

    from the reference's blocks: NormalParams.py:6-23, WishartParams.py:6-35 (with the `size`
    defect fixed), ExponentialFamilies.py:5-13, 27-31, 72-94, 186-189.  Free size
    d + d(d+1)/2 + 1 + d(d+1)/2 = (d+1)^2: d = 63 gives D = 4096.  Per observation
    l_n = 1/2 z^T Q z + c with z = [y; 1], Q = nu [[V, -V m], [-m^T V, m^T V m]]."""

    def __init__(self, par, y, prior_mean=None, prior_info=None, prior_df=None, prior_inv_scale=None,
                 mu_name='mu', lambda_name='lambda', weights=None, device=0):
        from scipy import special
        self._special = special
        y = _hip.as_f64(y)
        self.d = d = y.shape[1]
        mm = d * (d + 1) // 2
        o = par.vector_indices_dict[mu_name].start
        sub = par[mu_name]
        self._ms = range(o + sub.vector_indices_dict['mean'].start, o + sub.vector_indices_dict['mean'].stop)
        self._ls = range(o + sub.vector_indices_dict['info'].start, o + sub.vector_indices_dict['info'].stop)
        o = par.vector_indices_dict[lambda_name].start
        sub = par[lambda_name]
        self._inu = o + sub.vector_indices_dict['df'].start
        self._vs = range(o + sub.vector_indices_dict['v'].start, o + sub.vector_indices_dict['v'].stop)
        if len(self._ms) != d or len(self._ls) != mm or len(self._vs) != mm:
            raise ValueError('parameter sizes do not match the data dimension {}'.format(d))
        self._dup = duplication_matrix(d)
        super().__init__(par, np.hstack([y, np.ones((y.shape[0], 1))]), weights=weights, device=device)
        self._declare_priors(prior_mean, prior_info, prior_df, prior_inv_scale)

    def _declare_priors(self, prior_mean, prior_info, prior_df, prior_inv_scale):
        """prior_mean_par (d), prior_info_par, prior_inv_scale_par (symmetric matrices in vector form), prior_df_par."""
        d = self.d
        mm = d * (d + 1) // 2
        self._declare_hyper('prior_mean', HyperVectorParam('prior_mean', d, val=np.zeros(d) if prior_mean is None else _hip.as_f64(prior_mean).ravel()))
        self._declare_hyper('prior_info', HyperVectorParam('prior_info', mm, val=sym_to_vech(np.eye(d) if prior_info is None else prior_info)))
        self._declare_hyper('prior_df', HyperVectorParam('prior_df', 1, lb=float(d - 1), val=np.array([float(d + 2) if prior_df is None else float(prior_df)])))
        self._declare_hyper('prior_inv_scale', HyperVectorParam('prior_inv_scale', mm, val=sym_to_vech(np.eye(d) if prior_inv_scale is None else prior_inv_scale)))

    mu0 = property(lambda self: self._hyper_vec('prior_mean'))
    lam0 = property(lambda self: self._hyper_derived('prior_info', vech_to_sym))
    nu0 = property(lambda self: float(self._hyper_vec('prior_df')[0]))
    w0 = property(lambda self: self._hyper_derived('prior_inv_scale', vech_to_sym))

    def _prior_hyper(self, kind, eta, want):
        """Wishart prior (LRVB/ExponentialFamilies.py:72-94 terms): -1/2 (nu0 - d - 1) E log|Lambda| + 1/2 nu tr(W0 V)."""
        d = self.d
        m, lam_mu, nu, v = self._unpack(eta)
        if kind in ('prior_mean', 'prior_info'):
            P = np.linalg.inv(lam_mu)
            sub = kind[len('prior_'):]
            if want == 'grad':
                return mvn_prior_hyper_grad(sub, m - self.mu0, P, self.lam0)
            return mvn_prior_hyper_cross(sub, eta.size, self._ms, self._ls, m - self.mu0, P, self.lam0)
        vs = slice(self._vs.start, self._vs.stop)
        r, c = np.tril_indices(d)
        if kind == 'prior_df':
            kap, kap1, _, _ = self._kappa(nu)
            if want == 'grad':
                return np.array([-0.5 * (kap + d * np.log(2.0) + np.linalg.slogdet(v)[1])])
            C = np.zeros((eta.size, 1))
            C[self._inu, 0] = -0.5 * kap1
            C[vs, 0] = -0.5 * vech_dupT(np.linalg.inv(v))
            return C
        fac = np.where(r == c, 0.5, 1.0)                         # prior_inv_scale: 1/2 nu tr(W0 V) in the vector form of W0
        if want == 'grad':
            return nu * v[r, c] * fac
        C = np.zeros((eta.size, r.size))
        C[self._inu, :] = v[r, c] * fac
        C[self._vs.start + np.arange(r.size), np.arange(r.size)] = nu * fac
        return C

    def _unpack(self, eta):
        d = self.d
        m = eta[self._ms.start:self._ms.stop]
        lam_mu = (self._dup @ eta[self._ls.start:self._ls.stop]).reshape(d, d)
        nu = eta[self._inu]
        v = (self._dup @ eta[self._vs.start:self._vs.stop]).reshape(d, d)
        return m, lam_mu, nu, v

    def _kappa(self, nu):
        sp, d = self._special, self.d
        args = 0.5 * nu - 0.5 * np.arange(d)
        return (np.sum(sp.digamma(args)), 0.5 * np.sum(sp.polygamma(1, args)), 0.25 * np.sum(sp.polygamma(2, args)),
                np.sum(sp.gammaln(args)) + 0.25 * np.log(np.pi) * d * (d - 1.0))

    def _terms(self, eta, S, W, want_hess=True):
        d = self.d
        m, lam_mu, nu, v = self._unpack(eta)
        Syy, sy = S[:d, :d], S[:d, d]
        s1, ld_mu = np.linalg.slogdet(lam_mu)
        s2, ld_v = np.linalg.slogdet(v)
        if s1 <= 0 or s2 <= 0:
            raise ValueError('Matrix is not positive definite')
        P = np.linalg.inv(lam_mu)
        Vi = np.linalg.inv(v)
        kap, kap1, kap2, lgam = self._kappa(nu)
        A = Syy - np.outer(sy, m) - np.outer(m, sy) + W * np.outer(m, m)
        B = A + self.w0
        C = W * nu * v + self.lam0
        alpha = 0.5 * (W + self.nu0 - d - 1.0)
        gam = alpha + 0.5 * (d + 1.0)
        dm = m - self.mu0
        u = W * m - sy
        e_log_det = kap + d * np.log(2.0) + ld_v
        value = (0.5 * nu * np.sum(v * B) + 0.5 * np.sum(C * P) - alpha * e_log_det + 0.5 * dm @ self.lam0 @ dm
                 + 0.5 * ld_mu - 0.5 * d * (1.0 + np.log(2.0 * np.pi))
                 - 0.5 * (d + 1.0) * ld_v - 0.5 * d * (d + 1.0) * np.log(2.0) - lgam
                 + 0.5 * (nu - d - 1.0) * kap - 0.5 * nu * d)
        Vn = eta.size
        g = np.zeros(Vn)
        ms, ls, vs = (slice(r.start, r.stop) for r in (self._ms, self._ls, self._vs))
        inu = self._inu
        Dup = self._dup
        G = P @ C @ P
        g[ms] = nu * (v @ u) + self.lam0 @ dm
        g[ls] = Dup.T @ (-0.5 * G + 0.5 * P).ravel()
        g[inu] = 0.5 * np.sum(v * B) + 0.5 * W * np.sum(v * P) - alpha * kap1 + 0.5 * (nu - d - 1.0) * kap1 - 0.5 * d
        g[vs] = Dup.T @ (0.5 * nu * B + 0.5 * W * nu * P - gam * Vi).ravel()
        if not want_hess:
            return value, g, None
        # The Hessian as a list of blocks: ('dense', block, row, col, mirror) and ('symkron', A, B, coef, row,
        # col, mirror) = coef * Dup^T (A (x) B) Dup.  The same list feeds the host assembler (small d, tests)
        # and the device assembler (`lrvb_hvec_*`), which never forms a V x V matrix on the host.
        r, cidx = np.tril_indices(d)
        Hmv = np.zeros((d, r.size))                         # nu * kron(I, u^T) Dup, written directly
        cols = np.arange(r.size)
        np.add.at(Hmv, (r, cols), nu * u[cidx])
        off = r != cidx
        np.add.at(Hmv, (cidx[off], cols[off]), nu * u[r[off]])
        blocks = [
            ('dense', C, ms.start, ms.start, False),
            ('dense', (v @ u)[:, None], ms.start, inu, True),
            ('dense', Hmv, ms.start, vs.start, True),
            ('symkron', G, P, 0.5, ls.start, ls.start, False),
            ('symkron', P, G, 0.5, ls.start, ls.start, False),
            ('symkron', P, P, -0.5, ls.start, ls.start, False),
            ('dense', (Dup.T @ (-0.5 * W * (P @ v @ P)).ravel())[:, None], ls.start, inu, True),
            ('symkron', P, P, -0.5 * W * nu, ls.start, vs.start, True),
            ('dense', np.array([[0.5 * kap1 + (0.5 * (nu - d - 1.0) - alpha) * kap2]]), inu, inu, False),
            ('dense', (Dup.T @ (0.5 * B + 0.5 * W * P).ravel())[None, :], inu, vs.start, True),
            ('symkron', Vi, Vi, gam, vs.start, vs.start, False),
        ]
        if want_hess == 'blocks':
            return value, g, blocks
        return value, g, self._assemble_host(blocks, Vn)

    def _terms_vg(self, eta, S, W):
        return self._terms(eta, S, W, want_hess=False)[:2]

    def _assemble_host(self, blocks, Vn):
        H = np.zeros((Vn, Vn))
        Dup = self._dup
        for blk in blocks:
            if blk[0] == 'dense':
                _, Bk, ro, co, mirror = blk
                H[ro:ro + Bk.shape[0], co:co + Bk.shape[1]] += Bk
                if mirror:
                    H[co:co + Bk.shape[1], ro:ro + Bk.shape[0]] += Bk.T
            else:
                _, A, Bm, coef, ro, co, mirror = blk
                Bk = coef * (Dup.T @ np.kron(A, Bm) @ Dup)
                H[ro:ro + Bk.shape[0], co:co + Bk.shape[1]] += Bk
                if mirror:
                    H[co:co + Bk.shape[1], ro:ro + Bk.shape[0]] += Bk.T
        return H

    def hessian(self, x, is_free):
        """Dense Hessian.  From d = 16 upwards the Kronecker blocks are written by the device
        (`lrvb_hvec_add_symkron`) and the free-coordinate conversion runs on the resident matrix."""
        if self.d < 16:
            return super().hessian(x, is_free)
        S, W = self._stats()
        x = _hip.as_f64(x).ravel()
        _, g, blocks = self._terms(self._eta(x, is_free), S, W, want_hess='blocks')
        self.ctx.hvec_begin()
        for blk in blocks:
            if blk[0] == 'dense':
                self.ctx.hvec_add_block(blk[1], blk[2], blk[3], blk[4])
            else:
                self.ctx.hvec_add_symkron(blk[1], blk[2], blk[3], blk[4], blk[5], blk[6])
        return self.ctx.hvec_finish(x, g, is_free)

    def _obs_constants(self, eta):
        """c (V,): the constant part of d l_n / d eta_k = 1/2 z_n^T M_k z_n + c_k (everything that needs the d x d inverses and
        the polygamma sum; the matrices M_k themselves are written by the device, `lrvb_wishart_gram`)."""
        d = self.d
        m, lam_mu, nu, v = self._unpack(eta)
        P = np.linalg.inv(lam_mu)
        Vi = np.linalg.inv(v)
        _, kap1, _, _ = self._kappa(nu)
        c = np.zeros(eta.size)
        r, cidx = np.tril_indices(d)
        fac = np.where(r == cidx, 1.0, 2.0)
        c[self._ls.start:self._ls.stop] = -0.5 * nu * (P @ v @ P)[r, cidx] * fac
        c[self._inu] = 0.5 * np.sum(v * P) - 0.5 * kap1
        c[self._vs.start:self._vs.stop] = (0.5 * nu * P - 0.5 * Vi)[r, cidx] * fac
        return c, m, nu, v

    @_hip.host_blas
    def gram(self, free_val, want_host=True):
        """G^T G in free coordinates.  The V = (d + 1)^2 matrices of the per-observation gradient are generated on the device
        from (nu, m, V) -- nothing of size V q^2 is built on the host or sent over PCIe -- and with want_host=False the D x D
        result stays in HBM too (returns None; `ctx.chol_factor_last()` factors it)."""
        c, m, nu, v = self._obs_constants(self._eta(free_val, True))
        return self.ctx.wishart_gram(self.d, [self._ms.start, self._ls.start, self._inu, self._vs.start], nu, m, v, c, free_val,
                                     want_host=want_host)

    def _obs_terms(self, eta):
        d, q = self.d, self.q
        m, lam_mu, nu, v = self._unpack(eta)
        P = np.linalg.inv(lam_mu)
        Vi = np.linalg.inv(v)
        _, kap1, _, _ = self._kappa(nu)
        Vn = eta.size
        M = np.zeros((Vn, q, q))
        c = np.zeros(Vn)
        vm = v @ m
        Mm = M[self._ms.start:self._ms.start + d]
        Mm[:, :d, d] = -nu * v.T
        Mm[:, d, :d] = -nu * v.T
        Mm[:, d, d] = 2.0 * nu * vm
        PVP = P @ v @ P
        r, cidx = np.tril_indices(d)
        fac = np.where(r == cidx, 1.0, 2.0)
        c[self._ls.start:self._ls.stop] = -0.5 * nu * PVP[r, cidx] * fac
        Q0 = np.zeros((q, q))
        Q0[:d, :d] = v
        Q0[:d, d] = Q0[d, :d] = -vm
        Q0[d, d] = m @ vm
        M[self._inu] = Q0
        c[self._inu] = 0.5 * np.sum(v * P) - 0.5 * kap1
        cmat = 0.5 * nu * P - 0.5 * Vi
        c[self._vs.start:self._vs.stop] = cmat[r, cidx] * fac
        # the d (d + 1) / 2 matrices of the scale parameters, all at once (a Python loop over 2016 of them was ~10 ms at d = 63):
        # M_k = nu [[E_ij + E_ji, -Em], [-Em^T, m . Em]],  Em = e_i m_j + e_j m_i  (once for i = j)
        kk = np.arange(r.size)
        Mv = M[self._vs.start:self._vs.stop]
        off = r != cidx
        Mv[kk, r, cidx] += nu
        Mv[kk[off], cidx[off], r[off]] += nu
        Em = np.zeros((r.size, d))
        Em[kk, r] += m[cidx]
        Em[kk[off], cidx[off]] += m[r[off]]
        Mv[:, :d, d] = -nu * Em
        Mv[:, d, :d] = -nu * Em
        Mv[:, d, d] = nu * (Em @ m)
        return M, c
