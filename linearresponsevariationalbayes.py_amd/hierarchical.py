"""Hierarchical linear mixed model (BASELINE.json config 4; doc/lmm.lyx:77-160 of the reference):

    y_i ~ N(x_i^T beta + u_{g[i]}, tau_y^-1),   u_g ~ N(mu, tau_mu^-1),   i = 1..N,  g = 1..G

with q(beta) = MVNParam(p), q(mu) = UVNParam, q(tau_y), q(tau_mu) = GammaParam,
q(u) = UVNParamVector(G) (LRVB/NormalParams.py:6-76, GammaParams.py:4-16).  The -ELBO is a function
of the weighted sufficient statistics only (doc/lmm.lyx:105-160):

    S = sum_i w_i z_i z_i^T (z = [x; y]),  W = sum_i w_i,   per group: W_g, sum_g w y, sum_g w x

which the GPU produces in two streaming passes per weight vector (`lrvb_weighted_gram`,
`lrvb_group_sums`); across GPUs they are summed with one all-reduce.
The Hessian has ARROW structure -- a dense global block (p + p(p+1)/2 + 6 parameters) bordered by G
diagonal 2 x 2 local blocks -- so besides the dense Hessian the reference would build (used here at
small sizes for parity) the class exposes the Schur complement onto the global block, which is what
linear response needs for global moments at G = 1e4.
"""
import numpy as np
from scipy import special
from scipy import sparse as sp_sparse

from . import _hip
from .models import DeviceContext, DeclaredHypers, sym_to_vech, vech_to_sym, refuse_double_reduction
from .packing import VectorParam, HyperVectorParam, ResidentVector
from .quadform import (duplication_matrix, mvn_prior_hyper_grad, mvn_prior_hyper_cross, gamma_prior_hyper_grad,
                       gamma_prior_hyper_cross)


def _gamma_block(a, b, f_t, f_L):
    """Gradient (2,) and Hessian (2, 2) in (shape, rate) of  f_t * (a/b) + f_L * (psi(a) - log b) - entropy."""
    p1, p2 = special.polygamma(1, a), special.polygamma(2, a)
    g = np.array([f_t / b + f_L * p1 - (1.0 + (1.0 - a) * p1),
                  -f_t * a / b ** 2 - f_L / b + 1.0 / b])
    H = np.array([[f_L * p2 + p1 - (1.0 - a) * p2, -f_t / b ** 2],
                  [-f_t / b ** 2, 2.0 * f_t * a / b ** 3 + f_L / b ** 2 - 1.0 / b ** 2]])
    return g, H


def _gamma_entropy(a, b):
    return a - np.log(b) + special.gammaln(a) + (1.0 - a) * special.digamma(a)


class LMMObjective(DeclaredHypers):
    _lrvb_device_functor = True

    def __init__(self, par, x, y, groups, n_groups, beta_prior_mean=None, beta_prior_info=None,
                 mu_prior_mean=0.0, mu_prior_info=1.0, tau_y_prior=(1.0, 1.0), tau_mu_prior=(1.0, 1.0),
                 names=('beta', 'mu', 'tau_y', 'tau_mu', 'u'), weights=None, device=0):
        self.par = par
        x = _hip.as_f64(x)
        y = _hip.as_f64(y).reshape(-1, 1)
        self.n_obs, self.p = x.shape
        self.G = int(n_groups)
        p, G = self.p, self.G
        self._names = tuple(names)
        self._index(par, names)
        self._declare_priors(beta_prior_mean, beta_prior_info, mu_prior_mean, mu_prior_info, tau_y_prior, tau_mu_prior)
        self.ctx = DeviceContext(par.layout_blocks(), loss='data_only', n_obs=self.n_obs, n_cols=p + 1, device=device)
        self.ctx.set_data(_hip.SLOT_X, np.hstack([x, y]))
        self.ctx.set_groups(groups, G)
        w0 = np.ones(self.n_obs) if weights is None else _hip.as_f64(weights).ravel().copy()
        self._declare_hyper('weights', HyperVectorParam('weights', self.n_obs, val=w0))
        self.tilt_par = None
        self._w_res = ResidentVector()
        self._stats_cache = None
        self._external_stats = None

    # ---- the priors are hyper-parameters (LRVB/ModelSensitivity.py:555-612: prior sensitivity) -------------------------
    def _declare_priors(self, beta_prior_mean, beta_prior_info, mu_prior_mean, mu_prior_info, tau_y_prior, tau_mu_prior):
        """beta_prior_mean_par (p), beta_prior_info_par (symmetric matrix in vector form), mu_prior_par = (mean, information)
        of the normal prior on mu, tau_y_prior_par / tau_mu_prior_par = (shape, rate) of the two gamma priors."""
        p = self.p
        self._declare_hyper('beta_prior_mean', HyperVectorParam('beta_prior_mean', p, val=np.zeros(p) if beta_prior_mean is None else _hip.as_f64(beta_prior_mean).ravel()))
        self._declare_hyper('beta_prior_info', HyperVectorParam('beta_prior_info', p * (p + 1) // 2,
                                                                val=sym_to_vech(np.eye(p) if beta_prior_info is None else beta_prior_info)))
        self._declare_hyper('mu_prior', HyperVectorParam('mu_prior', 2, val=np.array([float(mu_prior_mean), float(mu_prior_info)])))
        self._declare_hyper('tau_y_prior', HyperVectorParam('tau_y_prior', 2, lb=0.0, val=np.array(list(map(float, tau_y_prior)))))
        self._declare_hyper('tau_mu_prior', HyperVectorParam('tau_mu_prior', 2, lb=0.0, val=np.array(list(map(float, tau_mu_prior)))))

    beta0 = property(lambda self: self._hyper_vec('beta_prior_mean'))
    lam0 = property(lambda self: self._hyper_derived('beta_prior_info', vech_to_sym))
    mu0 = property(lambda self: float(self._hyper_vec('mu_prior')[0]))
    kappa0 = property(lambda self: float(self._hyper_vec('mu_prior')[1]))
    a0y = property(lambda self: float(self._hyper_vec('tau_y_prior')[0]))
    b0y = property(lambda self: float(self._hyper_vec('tau_y_prior')[1]))
    a0m = property(lambda self: float(self._hyper_vec('tau_mu_prior')[0]))
    b0m = property(lambda self: float(self._hyper_vec('tau_mu_prior')[1]))

    def _prior_hyper(self, kind, eta_g, want):
        """d f / d eps (Ph,) or the GLOBAL rows of d2 f / d eta d eps^T (n_global x Ph) in vector coordinates; the priors do not
        touch the 2 G group parameters, whose rows are zero."""
        ng = self.n_global
        if kind in ('beta_prior_mean', 'beta_prior_info'):
            m = eta_g[self._ms]
            P = np.linalg.inv(self._sym_from_vech(eta_g[self._ls]))
            sub = kind[len('beta_prior_'):]
            if want == 'grad':
                return mvn_prior_hyper_grad(sub, m - self.beta0, P, self.lam0)
            return mvn_prior_hyper_cross(sub, ng, self._ms, self._ls, m - self.beta0, P, self.lam0)
        if kind == 'mu_prior':
            e_mu, i_mu = eta_g[self._iem], eta_g[self._iim]
            if want == 'grad':
                return np.array([-self.kappa0 * (e_mu - self.mu0), 0.5 * ((e_mu - self.mu0) ** 2 + 1.0 / i_mu)])
            C = np.zeros((ng, 2))
            C[self._iem, 0] = -self.kappa0
            C[self._iem, 1] = e_mu - self.mu0
            C[self._iim, 1] = -0.5 / i_mu ** 2
            return C
        ia, ib = (self._iay, self._iby) if kind == 'tau_y_prior' else (self._iam, self._ibm)
        if kind not in ('tau_y_prior', 'tau_mu_prior'):
            raise NotImplementedError(kind)
        if want == 'grad':
            return gamma_prior_hyper_grad(eta_g[ia], eta_g[ib], special)
        return gamma_prior_hyper_cross(ng, ia, ib, eta_g[ia], eta_g[ib], special)

    def hyper_grad(self, hyper_par, val1, val1_is_free):
        kind = self.hyper_kind(hyper_par)
        if kind == 'weights':
            raise NotImplementedError('d f / d weights of the hierarchical model is not declared')
        return self._prior_hyper(kind, self._eta(val1, val1_is_free)[:self.n_global], 'grad')

    @_hip.host_blas
    def cross_hessian(self, hyper_par, val1, val1_is_free):
        """d2 f / d par1 d hyper^T for the declared priors, all rows (dense protocol, small G): (n1, Ph)."""
        kind = self.hyper_kind(hyper_par)
        if kind == 'weights':
            raise NotImplementedError('the weight cross Hessian of the hierarchical model is not declared')
        val1 = _hip.as_f64(val1).ravel()
        Cg = self.global_cross_hessian(hyper_par, val1, is_free=val1_is_free)
        return np.vstack([Cg, np.zeros((val1.size - self.n_global, Cg.shape[1]))])

    @_hip.host_blas
    def global_cross_hessian(self, hyper_par, val, is_free=True):
        """The n_global rows of the cross Hessian with a prior hyper-parameter (the 2 G local rows are zero): closed form in
        vector coordinates, J_g^T applied on the device."""
        kind = self.hyper_kind(hyper_par)
        ng = self.n_global
        val = _hip.as_f64(val).ravel()
        gc = self._ensure_gctx()
        eta_g = gc.constrain(val[:ng]) if is_free else val[:ng]
        Cv = self._prior_hyper(kind, eta_g, 'cross')
        return gc.jac_t_matmul(val[:ng], Cv) if is_free else Cv

    def global_sensitivity(self, hyper_par, free_val):
        """d theta_global / d hyper^T = -H_S^-1 C_g (n_global x Ph): linear response of the global parameters to a prior
        (LRVB/ModelSensitivity.py:596-602) through the Schur complement of the arrow Hessian -- the local rows of the cross
        Hessian are zero, so the local parameters enter through H_S only."""
        Cg = self.global_cross_hessian(hyper_par, free_val)
        gc = self._ensure_gctx()
        if self._device_path and self._external_stats is None:
            self.global_hessian(free_val, want_host=False)
            gc.chol_factor_last()
        else:
            gc.chol_factor(self.global_hessian(free_val))
        return -gc.chol_solve(Cg)

    def _index(self, par, names):
        p, G = self.p, self.G
        nb, nm, nty, ntm, nu = names
        vi = par.vector_indices_dict
        sub = par[nb]
        self._ms = slice(vi[nb].start + sub.vector_indices_dict['mean'].start, vi[nb].start + sub.vector_indices_dict['mean'].stop)
        self._ls = slice(vi[nb].start + sub.vector_indices_dict['info'].start, vi[nb].start + sub.vector_indices_dict['info'].stop)
        sub = par[nm]
        self._iem = vi[nm].start + sub.vector_indices_dict['mean'].start
        self._iim = vi[nm].start + sub.vector_indices_dict['info'].start
        self._iay = vi[nty].start + par[nty].vector_indices_dict['shape'].start
        self._iby = vi[nty].start + par[nty].vector_indices_dict['rate'].start
        self._iam = vi[ntm].start + par[ntm].vector_indices_dict['shape'].start
        self._ibm = vi[ntm].start + par[ntm].vector_indices_dict['rate'].start
        sub = par[nu]
        self._es = slice(vi[nu].start + sub.vector_indices_dict['mean'].start, vi[nu].start + sub.vector_indices_dict['mean'].stop)
        self._is = slice(vi[nu].start + sub.vector_indices_dict['info'].start, vi[nu].start + sub.vector_indices_dict['info'].stop)
        if self._ms.stop - self._ms.start != p or self._es.stop - self._es.start != G:
            raise ValueError('parameter sizes do not match the data (p = {}, G = {})'.format(p, G))
        self.n_global = self._es.start                     # the local block (u) is pushed last
        if self._is.stop != par.vector_size() or self._is.start != self._es.stop:
            raise ValueError('the group-effect parameter must be pushed last')
        self._dup = duplication_matrix(p)
        self._tril = np.tril_indices(p)
        # the device elimination (lrvb_lmm_group_terms) takes the FREE local parameters as [e_1..e_G | log(i_g - lb)]
        fi = par.free_indices_dict
        fsub = par[nu].free_indices_dict
        self._fe = slice(fi[nu].start + fsub['mean'].start, fi[nu].start + fsub['mean'].stop)
        self._fi = slice(fi[nu].start + fsub['info'].start, fi[nu].start + fsub['info'].stop)
        info = par[nu]['info']
        self._info_lb = float(getattr(info, '_lb', 0.0))
        self._device_path = (self._fe.start == self.n_global and self._fi.start == self._fe.stop
                             and self._fi.stop == par.free_size() and np.isinf(getattr(info, '_ub', np.inf))
                             and self._iby == self._iay + 1 and self._ibm == self._iam + 1 and p <= 57)

    # ---- sufficient statistics (GPU) ---------------------------------------------------------
    def _push_state(self):
        w = self._w_res.changed(self.weights_par)              # O(1) for the objective's own HyperVectorParam
        if w is not None:
            self.ctx.set_weights(w)
            self._stats_cache = None
            self._S_dev = None

    def invalidate_stats(self):
        """Forget the cached statistics (the weights on the device were changed behind this object's back, or a
        benchmark wants the pass over the observations inside every step)."""
        self._stats_cache = None
        self._S_dev = None

    def device_stats(self):
        """S (q x q, host copy) with the group sums left ON THE DEVICE, both summed over the ranks inside the library
        when the context carries a reduce hook (`ShardedObjective`, `native_comm_init`)."""
        self._push_state()
        self._check_hook_epoch()
        if getattr(self, '_S_dev', None) is None:
            self._S_dev = self.ctx.grouped_stats(want_S=True, want_gs=False)[0]
        return self._S_dev

    def _check_hook_epoch(self):
        """A reduce hook installed or removed since the statistics were cached makes them stale (local vs global sums)."""
        if getattr(self, '_epoch', None) != self.ctx.hook_epoch:
            self._stats_cache = None
            self._S_dev = None
            self._epoch = self.ctx.hook_epoch

    def local_stats(self):
        """The statistics as one flat host vector [S (q*q) | group sums (G*(q+1))]: this process's own rows -- the
        buffer a host-side all-reduce (`allreduce_stats`, the gloo tests) sums over shards -- or, when the context carries
        a reduce hook, already the sum over all ranks (one reduction of the device buffer inside `lrvb_grouped_stats`)."""
        self._push_state()
        self._check_hook_epoch()
        if self._stats_cache is None:
            S, gs = self.ctx.grouped_stats(want_S=True, want_gs=True)
            self._stats_cache = np.concatenate([S.ravel(), gs.ravel()])
        return self._stats_cache

    def set_reduced_stats(self, flat):
        """Install statistics summed over all shards (None = use this process's own)."""
        refuse_double_reduction(getattr(self, "ctx", None), flat)
        self._external_stats = None if flat is None else np.asarray(flat, dtype=np.float64).copy()

    def _stats(self):
        flat = self._external_stats if self._external_stats is not None else self.local_stats()
        q = self.p + 1
        S = flat[:q * q].reshape(q, q)
        gs = flat[q * q:].reshape(self.G, q + 1)
        return S, gs

    # ---- closed forms in vector coordinates ------------------------------------------------------
    def _pieces(self, eta):
        p, G = self.p, self.G
        S, gs = self._stats()
        Sxx, Sxy, Syy = S[:p, :p], S[:p, p], S[p, p]
        Wg, sxg, syg = gs[:, 0], gs[:, 1:1 + p], gs[:, 1 + p]
        W = float(np.sum(Wg))
        m = eta[self._ms]
        lam = (self._dup @ eta[self._ls]).reshape(p, p)
        e_mu, i_mu = eta[self._iem], eta[self._iim]
        ay, by, am, bm = eta[self._iay], eta[self._iby], eta[self._iam], eta[self._ibm]
        eg, ig = eta[self._es], eta[self._is]
        P = np.linalg.inv(lam)
        return dict(Sxx=Sxx, Sxy=Sxy, Syy=Syy, Wg=Wg, sxg=sxg, syg=syg, W=W, m=m, lam=lam, P=P, e_mu=e_mu, i_mu=i_mu,
                    ay=ay, by=by, am=am, bm=bm, eg=eg, ig=ig)

    def _scalars(self, q):
        p, G = self.p, self.G
        m, P, eg, ig = q['m'], q['P'], q['eg'], q['ig']
        ty, tm = q['ay'] / q['by'], q['am'] / q['bm']
        Ly = special.digamma(q['ay']) - np.log(q['by'])
        Lm = special.digamma(q['am']) - np.log(q['bm'])
        rg = q['syg'] - q['sxg'] @ m
        rss = q['Syy'] - 2.0 * m @ q['Sxy'] + m @ q['Sxx'] @ m
        Ay = rss + np.sum(q['Sxx'] * P) - 2.0 * eg @ rg + np.sum(q['Wg'] * (eg ** 2 + 1.0 / ig))
        Am = np.sum((eg - q['e_mu']) ** 2 + 1.0 / ig) + G / q['i_mu']
        return ty, tm, Ly, Lm, rg, Ay, Am

    def value_vec(self, eta):
        p, G = self.p, self.G
        q = self._pieces(eta)
        ty, tm, Ly, Lm, rg, Ay, Am = self._scalars(q)
        sign, logdet = np.linalg.slogdet(q['lam'])
        if sign <= 0:
            raise ValueError('Matrix is not positive definite')
        dm = q['m'] - self.beta0
        return float(0.5 * ty * Ay - 0.5 * q['W'] * Ly + 0.5 * tm * Am - 0.5 * G * Lm
                     + 0.5 * (dm @ self.lam0 @ dm + np.sum(self.lam0 * q['P']))
                     + 0.5 * self.kappa0 * ((q['e_mu'] - self.mu0) ** 2 + 1.0 / q['i_mu'])
                     - (self.a0y - 1.0) * Ly + self.b0y * ty - (self.a0m - 1.0) * Lm + self.b0m * tm
                     + 0.5 * logdet - 0.5 * p * (1.0 + np.log(2.0 * np.pi))
                     + 0.5 * np.log(q['i_mu']) - 0.5 * (1.0 + np.log(2.0 * np.pi))
                     + 0.5 * np.sum(np.log(q['ig'])) - 0.5 * G * (1.0 + np.log(2.0 * np.pi))
                     - _gamma_entropy(q['ay'], q['by']) - _gamma_entropy(q['am'], q['bm']))

    def _arrow(self, eta, kron_block=True):
        """Gradient (V,), global Hessian block (ng, ng), cross block and the diagonal of the local block
        (2G,) in vector coordinates.  Only the mean of q(beta) and the five scalar parameters couple to the
        group effects, so the cross block is returned row-sparse: (rows (p + 5,), values (p + 5, 2G)) --
        the dense (ng, 2G) matrix would be 160 MB of zeros at G = 1e4."""
        p, G, ng = self.p, self.G, self.n_global
        q = self._pieces(eta)
        ty, tm, Ly, Lm, rg, Ay, Am = self._scalars(q)
        m, P, eg, ig, Wg, sxg = q['m'], q['P'], q['eg'], q['ig'], q['Wg'], q['sxg']
        Dup = self._dup
        V = eta.size
        g = np.zeros(V)
        Hgg = np.zeros((ng, ng))
        xrows = np.concatenate([np.arange(self._ms.start, self._ms.stop),
                                [self._iem, self._iay, self._iby, self._iam, self._ibm]])
        Hx = np.zeros((p + 5, 2 * G))
        um = q['Sxx'] @ m - q['Sxy'] + sxg.T @ eg
        C = ty * q['Sxx'] + self.lam0
        Gc = P @ C @ P
        PSP = P @ q['Sxx'] @ P
        f_ty, f_tm = 0.5 * Ay + self.b0y, 0.5 * Am + self.b0m
        f_Ly, f_Lm = -0.5 * q['W'] - (self.a0y - 1.0), -0.5 * G - (self.a0m - 1.0)
        tay, tby = 1.0 / q['by'], -q['ay'] / q['by'] ** 2
        tam, tbm = 1.0 / q['bm'], -q['am'] / q['bm'] ** 2
        ms, ls = self._ms, self._ls
        iem, iim, iay, iby, iam, ibm = self._iem, self._iim, self._iay, self._iby, self._iam, self._ibm
        dsum = np.sum(eg - q['e_mu'])
        # gradient
        g[ms] = ty * um + self.lam0 @ (m - self.beta0)
        g[ls] = Dup.T @ (-0.5 * Gc + 0.5 * P).ravel()
        g[iem] = -tm * dsum + self.kappa0 * (q['e_mu'] - self.mu0)
        g[iim] = -0.5 * (tm * G + self.kappa0) / q['i_mu'] ** 2 + 0.5 / q['i_mu']
        gy, Hy = _gamma_block(q['ay'], q['by'], f_ty, f_Ly)
        gm_, Hm_ = _gamma_block(q['am'], q['bm'], f_tm, f_Lm)
        g[iay], g[iby] = gy
        g[iam], g[ibm] = gm_
        dloc = ty * Wg + tm
        g[self._es] = ty * (Wg * eg - rg) + tm * (eg - q['e_mu'])
        g[self._is] = -0.5 * dloc / ig ** 2 + 0.5 / ig
        # global block
        Hgg[ms, ms] = C
        if kron_block:                    # (p(p+1)/2)^2 block through dense duplication-matrix algebra: small p only
            Hgg[ls, ls] = Dup.T @ (0.5 * (np.kron(Gc, P) + np.kron(P, Gc)) - 0.5 * np.kron(P, P)) @ Dup
        else:
            self._kron_factors = (Gc, P)
        Hgg[ms, iay] = Hgg[iay, ms] = um * tay
        Hgg[ms, iby] = Hgg[iby, ms] = um * tby
        gl = Dup.T @ (-0.5 * PSP).ravel()
        Hgg[ls, iay] = Hgg[iay, ls] = gl * tay
        Hgg[ls, iby] = Hgg[iby, ls] = gl * tby
        Hgg[iem, iem] = tm * G + self.kappa0
        Hgg[iem, iam] = Hgg[iam, iem] = -dsum * tam
        Hgg[iem, ibm] = Hgg[ibm, iem] = -dsum * tbm
        Hgg[iim, iim] = (tm * G + self.kappa0) / q['i_mu'] ** 3 - 0.5 / q['i_mu'] ** 2
        Hgg[iim, iam] = Hgg[iam, iim] = -0.5 * G / q['i_mu'] ** 2 * tam
        Hgg[iim, ibm] = Hgg[ibm, iim] = -0.5 * G / q['i_mu'] ** 2 * tbm
        Hgg[np.ix_([iay, iby], [iay, iby])] = Hy
        Hgg[np.ix_([iam, ibm], [iam, ibm])] = Hm_
        # cross block: columns [e_1..e_G | i_1..i_G]
        ce, ci = slice(0, G), slice(G, 2 * G)
        rem, ray, rby, ram, rbm = p, p + 1, p + 2, p + 3, p + 4
        Hx[:p, ce] = ty * sxg.T
        Hx[rem, ce] = -tm
        Hx[ray, ce] = (Wg * eg - rg) * tay
        Hx[rby, ce] = (Wg * eg - rg) * tby
        Hx[ram, ce] = (eg - q['e_mu']) * tam
        Hx[rbm, ce] = (eg - q['e_mu']) * tbm
        Hx[ray, ci] = -0.5 * Wg / ig ** 2 * tay
        Hx[rby, ci] = -0.5 * Wg / ig ** 2 * tby
        Hx[ram, ci] = -0.5 / ig ** 2 * tam
        Hx[rbm, ci] = -0.5 / ig ** 2 * tbm
        dl = np.concatenate([dloc, dloc / ig ** 3 - 0.5 / ig ** 2])
        return g, Hgg, (xrows, Hx), dl

    def _dense_vec(self, eta):
        g, Hgg, (xrows, Hx), dl = self._arrow(eta)
        ng, V = self.n_global, eta.size
        H = np.zeros((V, V))
        H[:ng, :ng] = Hgg
        H[xrows, ng:] = Hx
        H[ng:, xrows] = Hx.T
        H[np.arange(ng, V), np.arange(ng, V)] = dl
        return g, H

    # ---- functor protocol (dense; intended for small G) ---------------------------------------------
    def _eta(self, x, is_free):
        x = _hip.as_f64(x).ravel()
        return self.ctx.constrain(x) if is_free else x

    def __call__(self):
        return self.value(np.asarray(self.par.get_free(), dtype=np.float64), True)

    @_hip.host_blas
    def value(self, x, is_free):
        return self.value_vec(self._eta(x, is_free))

    @_hip.host_blas
    def grad(self, x, is_free):
        g = self._arrow(self._eta(x, is_free))[0]
        return self.ctx.free_to_vector_jac(x).T @ g if is_free else g

    jacobian = grad

    @_hip.host_blas
    def hessian(self, x, is_free):
        if self.par.vector_size() > 8192:
            raise MemoryError('dense Hessian of {} parameters: use global_hessian() (Schur complement)'.format(self.par.vector_size()))
        g, H = self._dense_vec(self._eta(x, is_free))
        return self.ctx.free_hessian_from_vector(x, g, H) if is_free else H

    def hvp(self, x, v, is_free):
        return self.hessian(x, is_free) @ _hip.as_f64(v).ravel()

    # ---- arrow structure: sparse export (SparseObjectives.get_sparse_sub_hessian, :587-594) ------------
    @_hip.host_blas
    def sparse_hessian(self, free_val):
        """The full free-coordinate Hessian as a scipy CSR matrix: dense global block, the cross block
        and the 2G diagonal local entries placed with `get_sparse_sub_matrix` -- the format in which the
        reference hands arrow-structured Hessians to sparse solvers."""
        from .objectives import get_sparse_sub_hessian, get_sparse_sub_matrix
        free_val = _hip.as_f64(free_val).ravel()
        eta = self.ctx.constrain(free_val)
        g, Hgg, (xrows, Hx), dl = self._arrow(eta)
        ng, G, D = self.n_global, self.G, free_val.size
        self._ensure_gctx()
        Hgg_free = self._gctx.free_hessian_from_vector(free_val[:ng], g[:ng], Hgg)
        ig = eta[self._is]
        jl = np.concatenate([np.ones(G), ig])
        dl_free = dl * jl ** 2 + np.concatenate([np.zeros(G), g[self._is] * ig])
        # the coupled rows have element-wise packing maps (identity / exp): their rows of J are diagonal
        jrow = self._global_jac_diag(free_val[:ng])[xrows]
        cross = (jrow[:, None] * Hx) * jl[None, :]
        gi, li = np.arange(ng), np.arange(ng, D)
        H = get_sparse_sub_hessian(Hgg_free, gi, D)
        H = H + get_sparse_sub_matrix(cross, xrows, li, D, D) + get_sparse_sub_matrix(cross.T, li, xrows, D, D)
        return (H + sp_sparse.diags(np.concatenate([np.zeros(ng), dl_free]), format='csr')).tocsr()

    def _global_jac_diag(self, free_g):
        """Diagonal of d eta_g / d free_g from the device packing Jacobian of the global block."""
        return np.diag(self._gctx.free_to_vector_jac(free_g)).copy()

    # ---- arrow structure: Schur complement onto the global block, free coordinates --------------------
    def _ensure_gctx(self):
        """Packing of the global block: a layout that covers only the global parameters."""
        if not hasattr(self, '_gctx'):
            blocks, size = [], 0
            for b in self.par.layout_blocks():
                if size >= self.n_global:
                    break
                blocks.append(b)
                size += b['vec_size']
            assert size == self.n_global
            self._gctx = DeviceContext(blocks, quad_kind=_hip.QUAD_DIAG, device=self.ctx.device)
        return self._gctx

    def _sym_from_vech(self, v):
        p = self.p
        L = np.zeros((p, p))
        L[self._tril] = v
        return L + L.T - np.diag(np.diag(L))

    def _vech_dupT(self, A):
        """Dup^T vec(A) for a symmetric A: its lower triangle with the off-diagonal entries doubled."""
        B = A + A.T
        B[np.diag_indices(self.p)] = np.diag(A)
        return B[self._tril]

    @_hip.host_blas
    def global_hessian(self, free_val, want_host=True):
        """H_S = H_gg - H_gl diag(H_ll)^-1 H_lg in FREE coordinates (n_global x n_global): the
        matrix whose inverse is the linear-response covariance block of the global parameters.

        Device path (this process's own statistics, or statistics reduced inside the library): the group sums never leave
        the GPU -- `lrvb_lmm_group_terms` eliminates the 2 G local parameters there and returns the p + 5 coupled rows of
        the Schur term plus a handful of sums over groups; the host evaluates the closed forms in the p x p matrices and
        the six scalars, the device writes the Kronecker block, converts to free coordinates and keeps the result for
        `chol_factor_last` (want_host=False skips the copy back)."""
        if self._device_path and self._external_stats is None:
            return self._global_hessian_device(_hip.as_f64(free_val).ravel(), want_host)
        return self._global_hessian_host(free_val)

    def _host_pack(self, fv):
        """The theta-only coefficients of the closed forms (`lmm_closed_forms_kernel`, csrc/k_lmm.hip): the global parameters
        are constrained by the host parameter objects -- no device round trip -- and P = Lambda^-1, P Lambda0 P and the four
        polygamma values are formed here; everything that touches the statistics happens on the device."""
        p = self.p
        fi = self.par.free_indices_dict
        sub = []
        for name in self._names[:4]:
            q = self.par[name]
            q.set_free(fv[fi[name].start:fi[name].stop])
            sub.append(q)
        beta, mu, tau_y, tau_mu = sub
        m = np.asarray(beta['mean'].get(), dtype=np.float64).ravel()
        lam = np.asarray(beta['info'].get(), dtype=np.float64)
        e_mu, i_mu = float(np.ravel(mu['mean'].get())[0]), float(np.ravel(mu['info'].get())[0])
        ay, by = float(np.ravel(tau_y['shape'].get())[0]), float(np.ravel(tau_y['rate'].get())[0])
        am, bm = float(np.ravel(tau_mu['shape'].get())[0]), float(np.ravel(tau_mu['rate'].get())[0])
        if np.linalg.slogdet(lam)[0] <= 0:
            raise ValueError('Matrix is not positive definite')
        P = np.linalg.inv(lam)
        lam0 = self.lam0
        hp = np.zeros(32 + 2 * p + 3 * p * p)
        hp[:23] = [ay / by, am / bm, e_mu, i_mu, ay, by, am, bm, 1.0 / by, -ay / by ** 2, 1.0 / bm, -am / bm ** 2,
                   self.kappa0, self.mu0, self.a0y, self.b0y, self.a0m, self.b0m, float(self.G),
                   special.polygamma(1, ay), special.polygamma(2, ay), special.polygamma(1, am), special.polygamma(2, am)]
        o = 32
        for arr in (m, self.beta0, P, lam0, P @ lam0 @ P):
            hp[o:o + arr.size] = np.ravel(arr)
            o += arr.size
        return hp

    def _global_hessian_device(self, fv, want_host):
        """ONE library call (lrvb_lmm_global_hessian, round 4): statistics in one pass, the 2 G local parameters eliminated,
        the closed forms evaluated where the statistics lie (they are affine in them, with theta-only coefficients sent up in
        the call's single upload), Kronecker block, free conversion -- no device-to-host copy inside a step."""
        ng, G = self.n_global, self.G
        if fv.size != ng + 2 * G:
            raise ValueError('Free value is the wrong length')
        gc = self._ensure_gctx()
        self._push_state()
        self._check_hook_epoch()
        idx = [self._ms.start, self._ls.start, self._iem, self._iim, self._iay, self._iby, self._iam, self._ibm]
        want_sums = bool(getattr(self, 'want_diagnostics', False))
        H, sums = self.ctx.lmm_global_hessian(gc, fv, self._host_pack(fv), idx, self._info_lb, want_sums=want_sums, want_host=want_host)
        self._S_dev = None                                       # (the statistics were formed inside the call, not cached here)
        self.last_local_grad_norm = float(np.sqrt(sums[70])) if want_sums else None
        return H

    def _global_hessian_device_stepwise(self, fv, want_host):
        """Round 3's route, call by call (statistics to the host, numpy closed forms, blocks back up): kept as the
        independent path the one-call route is tested against."""
        p, G, ng = self.p, self.G, self.n_global
        if fv.size != ng + 2 * G:
            raise ValueError('Free value is the wrong length')
        gc = self._ensure_gctx()
        S = self.device_stats()
        eta = gc.constrain(fv[:ng])
        Sxx, Sxy, Syy = S[:p, :p], S[:p, p], S[p, p]
        m = eta[self._ms]
        lam = self._sym_from_vech(eta[self._ls])
        e_mu, i_mu = eta[self._iem], eta[self._iim]
        ay, by, am, bm = eta[self._iay], eta[self._iby], eta[self._iam], eta[self._ibm]
        ty, tm = ay / by, am / bm
        tay, tby, tam, tbm = 1.0 / by, -ay / by ** 2, 1.0 / bm, -am / bm ** 2
        par = np.concatenate([[ty, tm, e_mu, tay, tby, tam, tbm, self._info_lb], m])
        sums, M = self.ctx.lmm_group_terms(par, fv[ng:])
        v1, s_eg_rg, s_W_e2, s_d2, dsum, W = sums[:p], sums[64], sums[65], sums[66], sums[67], sums[69]
        self.last_local_grad_norm = float(np.sqrt(sums[70]))
        P = np.linalg.inv(lam)
        SxxP = Sxx @ P
        um = Sxx @ m - Sxy + v1
        rss = Syy - 2.0 * (m @ Sxy) + m @ Sxx @ m
        Ay = rss + np.trace(SxxP) - 2.0 * s_eg_rg + s_W_e2
        Am = s_d2 + G / i_mu
        C = ty * Sxx + self.lam0
        Gc = P @ C @ P
        PSP = P @ SxxP
        f_ty, f_tm = 0.5 * Ay + self.b0y, 0.5 * Am + self.b0m
        f_Ly, f_Lm = -0.5 * W - (self.a0y - 1.0), -0.5 * G - (self.a0m - 1.0)
        ms, ls = self._ms, self._ls
        iem, iim, iay, iby, iam, ibm = self._iem, self._iim, self._iay, self._iby, self._iam, self._ibm
        # gradient of the global parameters (vector coordinates): feeds the second-order packing term
        g = np.zeros(ng)
        g[ms] = ty * um + self.lam0 @ (m - self.beta0)
        g[ls] = self._vech_dupT(-0.5 * Gc + 0.5 * P)
        g[iem] = -tm * dsum + self.kappa0 * (e_mu - self.mu0)
        g[iim] = -0.5 * (tm * G + self.kappa0) / i_mu ** 2 + 0.5 / i_mu
        gy, Hy = _gamma_block(ay, by, f_ty, f_Ly)
        gm_, Hm_ = _gamma_block(am, bm, f_tm, f_Lm)
        g[iay], g[iby] = gy
        g[iam], g[ibm] = gm_
        # the dense part of the global block lives on the p + 6 rows [mean of q(beta) | e_mu, i_mu, a_y, b_y, a_mu, b_mu]
        # (plus the two columns a_y, b_y of the information rows): one scattered block, the Schur term already subtracted
        n6 = p + 6
        B = np.zeros((n6, n6))
        jem, jim, jay, jby, jam, jbm = p, p + 1, p + 2, p + 3, p + 4, p + 5
        B[:p, :p] = C
        B[:p, jay] = B[jay, :p] = um * tay
        B[:p, jby] = B[jby, :p] = um * tby
        B[jem, jem] = tm * G + self.kappa0
        B[jem, jam] = B[jam, jem] = -dsum * tam
        B[jem, jbm] = B[jbm, jem] = -dsum * tbm
        B[jim, jim] = (tm * G + self.kappa0) / i_mu ** 3 - 0.5 / i_mu ** 2
        B[jim, jam] = B[jam, jim] = -0.5 * G / i_mu ** 2 * tam
        B[jim, jbm] = B[jbm, jim] = -0.5 * G / i_mu ** 2 * tbm
        B[jay:jby + 1, jay:jby + 1] = Hy
        B[jam:jbm + 1, jam:jbm + 1] = Hm_
        xr = np.concatenate([np.arange(p), [jem, jay, jby, jam, jbm]])            # the p + 5 coupled rows inside B
        B[np.ix_(xr, xr)] -= M
        rows = np.concatenate([np.arange(ms.start, ms.stop), [iem, iim, iay, iby, iam, ibm]])
        gl = self._vech_dupT(-0.5 * PSP)
        gc.hvec_begin()
        gc.hvec_add_indexed(B, rows, rows)
        gc.hvec_add_block(np.stack([gl * tay, gl * tby], axis=1), ls.start, iay, mirror=True)
        gc.hvec_add_symkron(Gc, P, 0.5, ls.start, ls.start)
        gc.hvec_add_symkron(P, Gc, 0.5, ls.start, ls.start)
        gc.hvec_add_symkron(P, P, -0.5, ls.start, ls.start)
        return gc.hvec_finish(fv[:ng], g, True, want_host=want_host)

    @_hip.host_blas
    def _global_hessian_host(self, free_val):
        free_val = _hip.as_f64(free_val).ravel()
        eta = self.ctx.constrain(free_val)
        g, Hgg, (rows, Hx), dl = self._arrow(eta, kron_block=False)
        Gc, P = self._kron_factors
        ng, G = self.n_global, self.G
        self._ensure_gctx()
        # global block: the dense part from the host, the Kronecker block of q(beta)'s information matrix written
        # by the device (lrvb_hvec_add_symkron), conversion to free coordinates on the resident matrix
        gc = self._gctx
        gc.hvec_begin()
        gc.hvec_add_block(Hgg, 0, 0)
        ls0 = self._ls.start
        gc.hvec_add_symkron(Gc, P, 0.5, ls0, ls0)
        gc.hvec_add_symkron(P, Gc, 0.5, ls0, ls0)
        gc.hvec_add_symkron(P, P, -0.5, ls0, ls0)
        Hgg_free = gc.hvec_finish(free_val[:ng], g[:ng], True)
        # local free coordinates: e_g unconstrained (d eta = 1), i_g = exp(f_g) (d eta = i_g, d2 eta = i_g)
        ig = eta[self._is]
        jl = np.concatenate([np.ones(G), ig])
        dl_free = dl * jl ** 2 + np.concatenate([np.zeros(G), g[self._is] * ig])
        # only the mean of q(beta) and the five scalar parameters couple to the group effects; their packing
        # maps are element-wise (identity / exp), so the free cross block is a row scaling of those rows
        assert not np.any((rows >= self._ls.start) & (rows < self._ls.stop))
        jrow = self._global_jac_diag(free_val[:ng])[rows]               # those rows of J are diagonal
        cross = (jrow[:, None] * Hx) * jl[None, :]
        HS = Hgg_free
        HS[np.ix_(rows, rows)] -= (cross / dl_free[None, :]) @ cross.T
        return HS
