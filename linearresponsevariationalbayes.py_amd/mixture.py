"""Dirichlet-multinomial mixture (BASELINE.json config 3): SimplexParam responsibilities per
observation with DirichletParamArray globals (LRVB/SimplexParams.py:69-175, DirichletParams.py:8-33).

    x_n | c_n = k ~ Multinomial(phi_k),   c_n ~ Categorical(pi),   pi ~ Dir(a0),   phi_k ~ Dir(b0)
    q(pi) = Dir(alpha) (K,),   q(phi_k) = Dir(beta[:, k]) (V, K),   q(c_n) = z_n  (SimplexParam (N, K))

The -ELBO is

    f = -sum_n w_n z_n^T Lam^T x~_n + sum_n w_n z_n . log z_n - prior(pi) - prior(phi) - H(pi) - sum_k H(phi_k)

with x~_n = (1, x_n) and Lam = [E log pi; E log phi] ((V + 1) x K).  Its Hessian is an ARROW: a dense
global block (K + V K parameters) bordered by N independent (K - 1) x (K - 1) simplex blocks.  The
reference would assemble those blocks through the triple Python loop over COO triplets of
`SimplexParam.free_to_vector_hess` (SimplexParams.py:106-155); here one wavefront per observation
forms the block from the closed forms (SimplexParams.py:33-63), factors it in registers and emits
the row of the Schur-complement operand (`lrvb_mixture_rows`, csrc/k_mixture.hip), and an MFMA GEMM
over the observation axis reduces the operand.  Everything that remains on the host is independent
of N: digamma chains of the Dirichlet blocks and the (K + V K)^2 Schur assembly.

Parameter order in `par`: the Dirichlet blocks first (any order), the SimplexParam LAST.
"""
import numpy as np
from scipy import special, sparse

from . import _hip
from .models import DeviceContext, DeclaredHypers, refuse_double_reduction
from .packing import VectorParam, HyperVectorParam, ResidentVector
from .specfun import polygamma12


def _dirichlet_terms(alpha, d):
    """t(alpha) = -sum_m d_m E log p_m - H(alpha) for ONE Dirichlet (vector alpha) with data/prior
    coefficients d: value, gradient and Hessian in alpha.  With e = alpha - 1 - d the entropy and the
    expectation terms collapse to  grad = e psi1(alpha) - sum(e) psi1(alpha_0)."""
    a0 = np.sum(alpha)
    M = alpha.size
    elog = special.digamma(alpha) - special.digamma(a0)
    ent = (np.sum(special.gammaln(alpha)) - special.gammaln(a0) + (a0 - M) * special.digamma(a0)
           - np.sum((alpha - 1.0) * special.digamma(alpha)))
    e = alpha - 1.0 - d
    p1, p2 = special.polygamma(1, alpha), special.polygamma(2, alpha)
    p10, p20 = special.polygamma(1, a0), special.polygamma(2, a0)
    val = -np.sum(d * elog) - ent
    grad = e * p1 - np.sum(e) * p10
    hess = np.diag(p1 + e * p2) - p10 - np.sum(e) * p20
    return val, grad, hess


class MixtureObjective(DeclaredHypers):
    _lrvb_device_functor = True

    def __init__(self, par, x, pi_prior=1.0, phi_prior=1.0, names=('pi', 'phi', 'z'), weights=None, device=0):
        self.par = par
        x = _hip.as_f64(x)
        self.n_obs, self.V = x.shape
        N, V = self.n_obs, self.V
        npi, nphi, nz = names
        vi, fi = par.vector_indices_dict, par.free_indices_dict
        zshape = par[nz].get().shape
        if zshape[0] != N:
            raise ValueError('the SimplexParam must have one row per observation')
        self.K = K = int(zshape[1])
        if par[nphi]['alpha'].get().shape != (V, K) or par[npi]['alpha'].get().shape != (K,):
            raise ValueError('expected pi of shape ({0},) and phi of shape ({1}, {0})'.format(K, V))
        self.n_global = ng = K + V * K
        if vi[nz].start != ng or vi[nz].stop != par.vector_size() or fi[nz].start != ng:
            raise ValueError('the SimplexParam must be pushed last, after the two Dirichlet blocks')
        self._ipi = np.arange(vi[npi].start, vi[npi].stop)                        # (K,)
        self._iphi = np.arange(vi[nphi].start, vi[nphi].stop).reshape(V, K)      # (V, K)
        self._lb = np.zeros(ng)
        self._lb[self._ipi] = par[npi]['alpha']._lb
        self._lb[self._iphi.ravel()] = par[nphi]['alpha']._lb
        if np.isfinite(par[npi]['alpha']._ub) or np.isfinite(par[nphi]['alpha']._ub):
            raise ValueError('the Dirichlet parameters must be bounded below only')
        self._declare_priors(pi_prior, phi_prior)
        # the device context carries the packing of the GLOBAL blocks only: the simplex rows are
        # constrained inside the row kernel and never materialised on the host
        blocks, size = [], 0
        for b in par.layout_blocks():
            if size >= ng:
                break
            blocks.append(b)
            size += b['vec_size']
        assert size == ng
        self.ctx = DeviceContext(blocks, loss='data_only', n_obs=N, n_cols=V, device=device)
        self.ctx.set_data(_hip.SLOT_X, x)
        w0 = np.ones(N) if weights is None else _hip.as_f64(weights).ravel().copy()
        self._declare_hyper('weights', HyperVectorParam('weights', N, val=w0))
        self.tilt_par = None
        self._w_res = ResidentVector()
        self._external_stats = None
        # Opt-in for repeated evaluations at ONE local point (benchmarks, several moment sets at an optimum): the
        # N (K - 1) simplex logits are uploaded by the first call and reused from HBM afterwards.  The caller promises
        # not to change them while this is set; `drop_resident_logits()` forces the next upload.
        self.keep_logits_resident = False
        self._fz_on_device = False

    # ---- the Dirichlet priors are hyper-parameters (LRVB/ModelSensitivity.py:555-612: prior sensitivity) ----------------
    def _declare_priors(self, pi_prior, phi_prior):
        """pi_prior_par (K) and phi_prior_par (V K, the (V, K) array flattened row-major): the concentration parameters of
        `dirichlet_prior` (LRVB/ExponentialFamilies.py:201-204) on the mixing weights and on the K topics."""
        K, V = self.K, self.V
        self._declare_hyper('pi_prior', HyperVectorParam('pi_prior', K, lb=0.0,
                                                         val=np.broadcast_to(_hip.as_f64(pi_prior), (K,)).astype(np.float64)))
        self._declare_hyper('phi_prior', HyperVectorParam('phi_prior', V * K, lb=0.0,
                                                          val=np.broadcast_to(_hip.as_f64(phi_prior), (V, K)).astype(np.float64).ravel()))

    a0 = property(lambda self: self._hyper_vec('pi_prior'))
    b0 = property(lambda self: self._hyper_vec('phi_prior').reshape(self.V, self.K))

    def _prior_hyper(self, kind, alpha, beta, want):
        """d f / d eps (Ph,) or the GLOBAL rows of d2 f / d eta d eps^T (n_global x Ph), vector coordinates: the prior enters
        through d = C + prior - 1 in `_dirichlet_terms`, so d f / d prior_m = -E log p_m and the gradient e psi1 - sum(e) psi1_0
        (e = alpha - 1 - d) moves by -psi1 on the diagonal, +psi1_0 inside each Dirichlet."""
        V, K, ng = self.V, self.K, self.n_global
        if kind == 'pi_prior':
            if want == 'grad':
                return -(special.digamma(alpha) - special.digamma(np.sum(alpha)))
            C = np.zeros((ng, K))
            C[np.ix_(self._ipi, np.arange(K))] = -np.diag(special.polygamma(1, alpha)) + special.polygamma(1, np.sum(alpha))
            return C
        if kind != 'phi_prior':
            raise NotImplementedError(kind)
        b0s = np.sum(beta, axis=0)
        if want == 'grad':
            return -(special.digamma(beta) - special.digamma(b0s)[None, :]).ravel()
        C = np.zeros((ng, V * K))
        p1, p10 = special.polygamma(1, beta), special.polygamma(1, b0s)
        cols = np.arange(V * K).reshape(V, K)
        for k in range(K):
            C[np.ix_(self._iphi[:, k], cols[:, k])] = -np.diag(p1[:, k]) + p10[k]
        return C

    def hyper_grad(self, hyper_par, val1, val1_is_free=True):
        kind = self.hyper_kind(hyper_par)
        if kind == 'weights':
            raise NotImplementedError('d f / d weights of the mixture is not declared')
        fg, _ = self._split(val1)
        alpha, beta, _ = self._lam(self._lb + np.exp(fg))
        return self._prior_hyper(kind, alpha, beta, 'grad')

    def global_cross_hessian(self, hyper_par, free_val):
        """The n_global rows of d2 f / d theta d prior^T in FREE coordinates (the N (K - 1) simplex rows are zero): the closed
        form in vector coordinates with the packing Jacobian of the Dirichlet blocks applied on the device."""
        kind = self.hyper_kind(hyper_par)
        if kind == 'weights':
            raise NotImplementedError('the weight cross Hessian of the mixture is not declared')
        fg, _ = self._split(free_val)
        alpha, beta, _ = self._lam(self._lb + np.exp(fg))
        return self.ctx.jac_t_matmul(fg, self._prior_hyper(kind, alpha, beta, 'cross'))

    def cross_hessian(self, hyper_par, val1, val1_is_free=True):
        """Functor protocol (all rows; small N): the global rows over zeros for the simplex logits."""
        if not val1_is_free:
            raise NotImplementedError('the mixture objective is evaluated in free coordinates')
        n_local = self.n_obs * (self.K - 1)
        if n_local * 8 > 2 ** 31:
            raise MemoryError('use global_cross_hessian / global_sensitivity: the simplex rows of the cross Hessian are zero')
        Cg = self.global_cross_hessian(hyper_par, val1)
        return np.vstack([Cg, np.zeros((n_local, Cg.shape[1]))])

    def global_sensitivity(self, hyper_par, free_val):
        """d theta_global / d prior^T = -H_S^-1 C_g: linear response of the Dirichlet parameters to a prior through the Schur
        complement (the N simplex blocks are eliminated on the device; their rows of the cross Hessian are zero)."""
        Cg = self.global_cross_hessian(hyper_par, free_val)
        if self._external_stats is None and self._canonical_order():
            self.global_hessian(free_val, want_host=False)
            self.ctx.chol_factor_last()
        else:
            self.ctx.chol_factor(self.global_hessian(free_val))
        return -self.ctx.chol_solve(Cg)

    def drop_resident_logits(self):
        self._fz_on_device = False

    def _fz_arg(self, fz):
        keep = getattr(self, 'keep_logits_resident', False)
        if keep and getattr(self, '_fz_on_device', False):
            return None
        self._fz_on_device = bool(keep)
        return fz

    def _push_state(self):
        w = self._w_res.changed(self.weights_par)              # O(1) for the objective's own HyperVectorParam
        if w is not None:
            self.ctx.set_weights(w)

    # ---- pieces ------------------------------------------------------------------------------------
    def _split(self, free_val):
        free_val = _hip.as_f64(free_val).ravel()
        if free_val.size != self.n_global + self.n_obs * (self.K - 1):
            raise ValueError('Free value is the wrong length')
        return free_val[:self.n_global], free_val[self.n_global:]

    def _lam(self, eta_g):
        V, K = self.V, self.K
        alpha = eta_g[self._ipi]
        beta = eta_g[self._iphi]
        lam = np.empty((V + 1, K))
        lam[0] = special.digamma(alpha) - special.digamma(np.sum(alpha))
        lam[1:] = special.digamma(beta) - special.digamma(np.sum(beta, axis=0, keepdims=True))
        return alpha, beta, lam

    def _global_terms(self, alpha, beta, C):
        """Value, vector-coordinate gradient (ng,) and Hessian (ng, ng) of everything that depends on
        the Dirichlet parameters, at FIXED responsibilities; C = sum_n w_n x~_n z_n^T.  The K columns of phi are K
        independent Dirichlets: their terms (`_dirichlet_terms`, one column at a time) are evaluated for all columns at once."""
        V, K, ng = self.V, self.K, self.n_global
        g = np.zeros(ng)
        H = np.zeros((ng, ng))
        val, gp, Hp = _dirichlet_terms(alpha, C[0] + self.a0 - 1.0)
        g[self._ipi] = gp
        H[np.ix_(self._ipi, self._ipi)] = Hp
        d = C[1:] + self.b0 - 1.0                                   # (V, K)
        b0 = np.sum(beta, axis=0)                                    # (K,)
        dg, dg0 = special.digamma(beta), special.digamma(b0)
        ent = (np.sum(special.gammaln(beta), axis=0) - special.gammaln(b0) + (b0 - V) * dg0 - np.sum((beta - 1.0) * dg, axis=0))
        e = beta - 1.0 - d
        es = np.sum(e, axis=0)
        p1, p2 = special.polygamma(1, beta), special.polygamma(2, beta)
        p10, p20 = special.polygamma(1, b0), special.polygamma(2, b0)
        val += -np.sum(d * (dg - dg0)) - np.sum(ent)
        g[self._iphi] = e * p1 - es * p10
        idx = self._iphi.T                                           # (K, V): the parameters of column k
        blocks = (-p10 - es * p20)[:, None, None] + np.eye(V)[None] * (p1 + e * p2).T[:, :, None]
        H[idx[:, :, None], idx[:, None, :]] = blocks
        return val, g, H

    def _dlam(self, alpha, beta):
        """d vec(Lam) / d eta_g  ((V + 1) K x ng), vec index j K + k."""
        V, K, ng = self.V, self.K, self.n_global
        DL = np.zeros(((V + 1) * K, ng))
        DL[np.ix_(np.arange(K), self._ipi)] = np.diag(special.polygamma(1, alpha)) - special.polygamma(1, np.sum(alpha))
        rows = ((np.arange(V) + 1) * K)[None, :] + np.arange(K)[:, None]          # (K, V): rows (j + 1) K + k of column k
        idx = self._iphi.T
        p1 = special.polygamma(1, beta)                                          # (V, K)
        p10 = special.polygamma(1, np.sum(beta, axis=0))
        DL[rows[:, :, None], idx[:, None, :]] = np.eye(V)[None] * p1.T[:, :, None] - p10[:, None, None]
        return DL

    def _rows(self, free_val, want_grad, want_schur):
        fg, fz = self._split(free_val)
        fz = self._fz_arg(fz)
        eta_g = self._lb + np.exp(fg)                        # lower-bounded box blocks (Parameters.py:24-54)
        alpha, beta, lam = self._lam(eta_g)
        gz = None
        if self._external_stats is None or want_grad:
            self._push_state()
            val2, gz, S64, R = self.ctx.mixture_rows(self.K, fz, lam, want_grad=want_grad,
                                                     want_schur=want_schur and self._external_stats is None)
        if self._external_stats is not None:
            val2, S64, R = self._unpack_stats(self._external_stats)
        C = S64[:self.V + 1, 32:32 + self.K]
        return fg, eta_g, alpha, beta, val2, gz, C, R

    # ---- statistics that are summed over shards of the observation axis ----------------------------
    def _unpack_stats(self, flat):
        q2k2 = (self.V + 1) ** 2 * self.K ** 2
        if flat.size != 2 + 4096 + q2k2:
            raise ValueError('expected a statistics vector of length {}'.format(2 + 4096 + q2k2))
        return flat[:2], flat[2:2 + 4096].reshape(64, 64), flat[2 + 4096:].reshape((self.V + 1) ** 2, self.K ** 2)

    def local_stats(self, free_val):
        """[val2 (2) | S64 (4096) | R ((V+1)^2 K^2)] at free_val: of THIS process's rows (the buffer a host-side exchange
        all-reduces when observations and their simplex rows are sharded over GPUs), or -- with a reduce hook on the
        context -- already the sums over all ranks, reduced on the device inside the statistics call."""
        self._push_state()
        fg, fz = self._split(free_val)
        _, _, lam = self._lam(self._lb + np.exp(fg))
        val2, _, S64, R = self.ctx.mixture_rows(self.K, self._fz_arg(fz), lam, want_grad=False, want_schur=True)
        return np.concatenate([val2, S64.ravel(), R.ravel()])

    def set_reduced_stats(self, flat):
        """Install statistics summed over all shards (None = use this process's own)."""
        refuse_double_reduction(getattr(self, "ctx", None), flat)
        self._external_stats = None if flat is None else np.asarray(flat, dtype=np.float64).copy()

    # ---- functor protocol ------------------------------------------------------------------------------
    def __call__(self):
        return self.value(np.asarray(self.par.get_free(), dtype=np.float64), True)

    @_hip.host_blas
    def value(self, x, is_free=True):
        if not is_free:
            raise NotImplementedError('the mixture objective is evaluated in free coordinates')
        fg, eta_g, alpha, beta, val2, _, C, _ = self._rows(x, False, False)
        # val2[0] = -tr(Lam^T C) is already inside the Dirichlet terms (d = C + prior - 1)
        return float(val2[1] + self._global_terms(alpha, beta, C)[0])

    @_hip.host_blas
    def grad(self, x, is_free=True):
        if not is_free:
            raise NotImplementedError('the mixture objective is evaluated in free coordinates')
        fg, eta_g, alpha, beta, val2, gz, C, _ = self._rows(x, True, False)
        g_vec = self._global_terms(alpha, beta, C)[1]
        return np.concatenate([g_vec * (eta_g - self._lb), gz.ravel()])

    jacobian = grad

    def hessian(self, x, is_free=True):
        raise MemoryError('the dense Hessian of a mixture has N (K - 1) local rows: use global_hessian() '
                          '(Schur complement onto the Dirichlet block)')

    # ---- arrow structure ---------------------------------------------------------------------------------
    def _canonical_order(self):
        K, V = self.K, self.V
        return (np.array_equal(self._ipi, np.arange(K)) and np.array_equal(self._iphi.ravel(), K + np.arange(V * K))
                and np.all(self._lb == self._lb[0]))

    def _global_hessian_device(self, free_val, want_host):
        """Everything O(n^2) on the device: the statistics call leaves the Schur operand in HBM (summed over the ranks
        there when the context carries a reduce hook), the Dirichlet blocks of d Lam / d alpha and of the global Hessian
        are "diagonal plus a constant per Dirichlet" and are generated by a kernel from O(n) numbers
        (`lrvb_mixture_schur_dirichlet`), and the result stays resident for `chol_factor_last`."""
        V, K = self.V, self.K
        fg, fz = self._split(free_val)
        eta_g = self._lb + np.exp(fg)
        alpha, beta, lam = self._lam(eta_g)
        self._push_state()
        val2, S64 = self.ctx.mixture_stats(K, self._fz_arg(fz), lam, want_schur=True)
        C = S64[:V + 1, 32:32 + K]
        jg = eta_g - self._lb
        # pi
        a0s = np.sum(alpha)
        b0s = np.sum(beta, axis=0)
        # every trigamma / tetragamma of the step in ONE vectorised pass (specfun.polygamma12; scipy's took 0.3-0.7 ms here)
        t1, t2 = polygamma12(np.concatenate([alpha, [a0s], beta.ravel(), b0s]))
        p1, p2, p10, p20 = t1[:K], t2[:K], t1[K], t2[K]
        q1, q2 = t1[K + 1:K + 1 + V * K].reshape(V, K), t2[K + 1:K + 1 + V * K].reshape(V, K)
        q10, q20 = t1[K + 1 + V * K:], t2[K + 1 + V * K:]
        e = alpha - 1.0 - (C[0] + self.a0 - 1.0)
        se = np.sum(e)
        g_pi, hd_pi, hc_pi, dc_pi = e * p1 - se * p10, p1 + e * p2, -p10 - se * p20, -p10
        # phi: K independent Dirichlets over the V rows
        eb = beta - 1.0 - (C[1:] + self.b0 - 1.0)
        es = np.sum(eb, axis=0)
        g_phi, hd_phi = eb * q1 - es * q10, q1 + eb * q2
        g_vec = np.concatenate([g_pi, g_phi.ravel()])
        return self.ctx.mixture_schur_dirichlet(
            K, V + 1, dl_diag=np.concatenate([p1, q1.ravel()]), dl_const=np.concatenate([[dc_pi], -q10]),
            h_diag=np.concatenate([hd_pi, hd_phi.ravel()]), h_const=np.concatenate([[hc_pi], -q10 - es * q20]),
            scale=jg, diag_add=g_vec * jg, want_host=want_host)

    @_hip.host_blas
    def global_hessian(self, free_val, return_parts=False, want_host=True):
        """H_S = H_gg - sum_n H_gn H_nn^-1 H_ng in FREE coordinates ((K + V K) square): the matrix whose
        inverse is the linear-response covariance of the Dirichlet parameters.  want_host=False leaves the result on
        the device (for `ctx.chol_factor_last()`) and returns None."""
        V, K = self.V, self.K
        if self._external_stats is None and not return_parts and self._canonical_order():
            return self._global_hessian_device(free_val, want_host)
        fg, eta_g, alpha, beta, val2, gz, C, R = self._rows(free_val, False, True)
        _, g_vec, Hgg = self._global_terms(alpha, beta, C)
        q = V + 1
        jg = eta_g - self._lb                                 # d alpha / d free (= d2 alpha / d free2)
        # Schur term  DLam^T Rm DLam  with  Rm[(j,k),(j',k')] = R[(j,j'),(k,k')], and the free-coordinate chain
        # rule of the lower-bounded boxes, on the device: two (K q)^3 MFMA GEMMs (lrvb_mixture_schur).  R is
        # taken from the device when this process's own rows produced it, from the host after an all-reduce.
        own = self._external_stats is None
        HS = self.ctx.mixture_schur(K, q, None if own else R, self._dlam(alpha, beta) * jg[None, :], Hgg,
                                    scale=jg, diag_add=g_vec * jg)
        if return_parts:
            Hgg_free = Hgg * jg[:, None] * jg[None, :] + np.diag(g_vec * jg)
            return HS, Hgg_free, Hgg_free - HS
        return HS

    def global_cov(self, free_val, moment_jac=None):
        """Linear-response covariance  M H_S^-1 M^T  of moments of the Dirichlet parameters whose
        free Jacobian is M (default: the free parameters themselves), by the device Cholesky path."""
        if self._external_stats is None and self._canonical_order():
            self.global_hessian(free_val, want_host=False)          # stays on the device: factored where it lies
            self.ctx.chol_factor_last()
        else:
            self.ctx.chol_factor(self.global_hessian(free_val))
        M = np.eye(self.n_global) if moment_jac is None else _hip.as_f64(moment_jac)
        return self.ctx.lrvb_cov(M)

    def e_z(self, free_val):
        """Responsibilities z = softmax([0, f]) on the host (small N; diagnostics and tests)."""
        _, fz = self._split(free_val)
        f = np.hstack([np.zeros((self.n_obs, 1)), fz.reshape(self.n_obs, self.K - 1)])
        f -= f.max(axis=1, keepdims=True)
        e = np.exp(f)
        return e / e.sum(axis=1, keepdims=True)
