"""Independent torch-fp64 restatement of the packing maps and the declared objective, written
only so that exact AD (torch.func) can check the oracle's ANALYTIC derivatives.  Exact
derivatives are unique, so agreement here pins the oracle's Hessians to what autograd (the
reference's engine, absent from this image) would return for the same function."""
import math
import numpy as np
import torch

torch.set_default_dtype(torch.float64)
BOX, PSD, SIMPLEX = 0, 1, 2


def constrain_block(f, b):
    if b.kind == BOX:
        lo, hi = b.lb, b.ub
        if math.isinf(lo) and math.isinf(hi):
            return f
        if not math.isinf(lo) and math.isinf(hi):
            return torch.exp(f) + lo
        if math.isinf(lo):
            return hi - torch.exp(-f)
        return (hi - lo) * torch.sigmoid(f) + lo
    if b.kind == PSD:
        k = b.dim0
        idx = torch.tril_indices(k, k)
        L = torch.zeros(k, k, dtype=f.dtype)
        L = L.index_put((idx[0], idx[1]), f)
        d = torch.diagonal(L)
        L = L - torch.diag(d) + torch.diag(torch.exp(d))
        A = L @ L.T + b.lb * torch.eye(k)
        return A[idx[0], idx[1]]
    rows, K = b.dim0, b.dim1
    F = f.reshape(rows, K - 1)
    Faug = torch.cat([torch.zeros(rows, 1), F], dim=1)
    return torch.softmax(Faug, dim=1).reshape(-1)


def constrain(theta, layout):
    return torch.cat([constrain_block(theta[b.free_off:b.free_off + b.free_size], b) for b in layout.blocks])


def make_objective(model):
    """Returns f_vec(eta) as a torch function for an oracle.models.DeclaredModel."""
    x = None if model.x is None else torch.tensor(model.x)
    y = None if model.y is None else torch.tensor(model.y)
    w = None if model.w is None else torch.tensor(model.w)
    A = None if model.quad_A is None else torch.tensor(model.quad_A)
    m = torch.tensor(model.quad_m)
    bq = torch.tensor(model.quad_b)

    def f_vec(eta):
        val = torch.zeros((), dtype=eta.dtype)
        if model.loss:
            z = x @ eta[model.glm_off:model.glm_off + model.P]
            if model.loss == 1:
                l = 0.5 * model.lik_info * (y - z) ** 2
            elif model.loss == 2:
                l = torch.nn.functional.softplus(z) - y * z
            else:
                l = torch.exp(z) - y * z
            val = val + torch.sum(w * l)
        if A is not None:
            d = eta - m
            Ad = A * d if A.dim() == 1 else A @ d
            val = val + model.quad_scale * (0.5 * torch.dot(d, Ad) + torch.dot(bq, eta))
        return val

    return f_vec


def make_free_objective(model):
    f_vec = make_objective(model)
    return lambda theta: f_vec(constrain(theta, model.layout))


def example_objective(layout, x, y):
    """Example.ipynb:247-274 in torch; returns f(theta, w)."""
    xt, yt = torch.tensor(x), torch.tensor(y)
    dx, dy = x.shape[1], y.shape[1]
    idx = torch.tril_indices(dy, dy)

    def f(theta, w):
        eta = constrain(theta, layout)
        beta = eta[:dx * dy].reshape(dx, dy)
        lam_l = torch.zeros(dy, dy, dtype=eta.dtype).index_put((idx[0], idx[1]), eta[dx * dy:])
        lam = lam_l + lam_l.T - torch.diag(torch.diagonal(lam_l))
        r = yt - xt @ beta
        y_term = -0.5 * torch.einsum('ni,ij,nj,n', r, lam, r, w)
        return -(y_term + 0.5 * torch.sum(w) * torch.logdet(lam))
    return f


def mvn_regression_objective(x, y, k, mu0, lam0, a0, b0, layout=None):
    """Config 2 (-ELBO of the conjugate-normal regression) in torch as f(point, w); `point` is the
    vector eta = [m (k), tril(Lambda), shape, rate] or, with `layout`, the free vector."""
    xt, yt = torch.tensor(x), torch.tensor(y).reshape(-1)
    mu0t, lam0t = torch.as_tensor(mu0, dtype=torch.float64), torch.as_tensor(lam0, dtype=torch.float64)
    idx = torch.tril_indices(k, k)
    mm = k * (k + 1) // 2

    def f(point, w):
        eta = constrain(point, layout) if layout is not None else point
        m = eta[:k]
        ll = torch.zeros(k, k, dtype=eta.dtype).index_put((idx[0], idx[1]), eta[k:k + mm])
        lam = ll + ll.T - torch.diag(torch.diagonal(ll))
        a, b = eta[k + mm], eta[k + mm + 1]
        sigma = torch.linalg.inv(lam)
        e_tau, e_log_tau = a / b, torch.special.digamma(a) - torch.log(b)
        resid = yt - xt @ m
        quad = torch.einsum('ni,ij,nj->n', xt, sigma, xt)
        e_log_lik = torch.sum(w * (-0.5 * e_tau * (resid ** 2 + quad) + 0.5 * e_log_tau))
        dm = m - mu0t
        mvn_prior = -0.5 * (dm @ lam0t @ dm + torch.trace(lam0t @ sigma))
        gamma_prior = (a0 - 1.0) * e_log_tau - b0 * e_tau
        mvn_entropy = 0.5 * (-torch.logdet(lam) + k + k * math.log(2 * math.pi))
        gamma_entropy = a - torch.log(b) + torch.lgamma(a) + (1.0 - a) * torch.special.digamma(a)
        return -(e_log_lik + mvn_prior + gamma_prior + mvn_entropy + gamma_entropy)
    return f


def wishart_mvn_objective(y, d, mu0, lam0, nu0, w0, layout=None):
    """Config 5 (-ELBO of the Wishart + MVN full-covariance model) in torch as f(point, w); point =
    eta = [m (d), tril(Lambda_mu), nu, tril(V)] or the free vector when `layout` is given."""
    yt = torch.tensor(y)
    mu0t, lam0t, w0t = (torch.as_tensor(t, dtype=torch.float64) for t in (mu0, lam0, w0))
    idx = torch.tril_indices(d, d)
    mm = d * (d + 1) // 2
    half_i = 0.5 * torch.arange(d, dtype=torch.float64)

    def sym(vec):
        ll = torch.zeros(d, d, dtype=vec.dtype).index_put((idx[0], idx[1]), vec)
        return ll + ll.T - torch.diag(torch.diagonal(ll))

    def f(point, w):
        eta = constrain(point, layout) if layout is not None else point
        m = eta[:d]
        lam_mu = sym(eta[d:d + mm])
        nu = eta[d + mm]
        v = sym(eta[d + mm + 1:])
        sigma_mu = torch.linalg.inv(lam_mu)
        mdig = torch.sum(torch.special.digamma(0.5 * nu - half_i))
        mlgam = torch.sum(torch.lgamma(0.5 * nu - half_i)) + 0.25 * math.log(math.pi) * d * (d - 1.0)
        e_log_det = mdig + d * math.log(2.0) + torch.logdet(v)
        r = yt - m
        quad = torch.einsum('ni,ij,nj->n', r, v, r)
        e_log_lik = torch.sum(w * (-0.5 * nu * (quad + torch.trace(v @ sigma_mu)) + 0.5 * e_log_det))
        dm = m - mu0t
        mvn_prior = -0.5 * (dm @ lam0t @ dm + torch.trace(lam0t @ sigma_mu))
        wishart_prior = 0.5 * (nu0 - d - 1.0) * e_log_det - 0.5 * nu * torch.trace(w0t @ v)
        mvn_entropy = 0.5 * (-torch.logdet(lam_mu) + d + d * math.log(2 * math.pi))
        wishart_entropy = (0.5 * (d + 1) * torch.logdet(v) + 0.5 * d * (d + 1) * math.log(2.0) + mlgam
                           - 0.5 * (nu - d - 1.0) * mdig + 0.5 * nu * d)
        return -(e_log_lik + mvn_prior + wishart_prior + mvn_entropy + wishart_entropy)
    return f


def lmm_objective(x, y, gid, G, beta0, lam0, mu0, kappa0, tau_y_prior, tau_mu_prior, layout=None):
    """Config 4 (-ELBO of the hierarchical LMM, doc/lmm.lyx:77-87) in torch as f(point, w); point =
    eta = [m (p), tril(Lambda), e_mu, i_mu, a_y, b_y, a_mu, b_mu, e_1..e_G, i_1..i_G] or the free vector."""
    xt, yt = torch.tensor(x), torch.tensor(y).reshape(-1)
    gt = torch.tensor(gid, dtype=torch.long)
    p = x.shape[1]
    mm = p * (p + 1) // 2
    b0t, l0t = torch.as_tensor(beta0, dtype=torch.float64), torch.as_tensor(lam0, dtype=torch.float64)
    idx = torch.tril_indices(p, p)
    a0y, b0y = tau_y_prior
    a0m, b0m = tau_mu_prior

    def gam_entropy(a, b):
        return a - torch.log(b) + torch.lgamma(a) + (1.0 - a) * torch.special.digamma(a)

    def f(point, w):
        eta = constrain(point, layout) if layout is not None else point
        m = eta[:p]
        ll = torch.zeros(p, p, dtype=eta.dtype).index_put((idx[0], idx[1]), eta[p:p + mm])
        lam = ll + ll.T - torch.diag(torch.diagonal(ll))
        o = p + mm
        e_mu, i_mu, ay, by, am, bm = eta[o], eta[o + 1], eta[o + 2], eta[o + 3], eta[o + 4], eta[o + 5]
        eg, ig = eta[o + 6:o + 6 + G], eta[o + 6 + G:o + 6 + 2 * G]
        sigma = torch.linalg.inv(lam)
        ty, tm = ay / by, am / bm
        Ly, Lm = torch.special.digamma(ay) - torch.log(by), torch.special.digamma(am) - torch.log(bm)
        resid = yt - xt @ m - eg[gt]
        e_sq = resid ** 2 + torch.einsum('ni,ij,nj->n', xt, sigma, xt) + 1.0 / ig[gt]
        e_log_lik = torch.sum(w * (-0.5 * ty * e_sq + 0.5 * Ly))
        e_log_u = torch.sum(-0.5 * tm * ((eg - e_mu) ** 2 + 1.0 / ig + 1.0 / i_mu) + 0.5 * Lm)
        dm = m - b0t
        beta_prior = -0.5 * (dm @ l0t @ dm + torch.trace(l0t @ sigma))
        mu_prior = -0.5 * kappa0 * ((e_mu - mu0) ** 2 + 1.0 / i_mu)
        tau_priors = (a0y - 1.0) * Ly - b0y * ty + (a0m - 1.0) * Lm - b0m * tm
        ent = (0.5 * (-torch.logdet(lam) + p + p * math.log(2 * math.pi))
               + 0.5 * (-torch.log(i_mu) + 1.0 + math.log(2 * math.pi))
               + 0.5 * torch.sum(-torch.log(ig) + 1.0 + math.log(2 * math.pi))
               + gam_entropy(ay, by) + gam_entropy(am, bm))
        return -(e_log_lik + e_log_u + beta_prior + mu_prior + tau_priors + ent)
    return f


def mixture_objective(x, K, a0, b0, lb=0.0):
    """-ELBO of the Dirichlet-multinomial mixture as a function of the FREE vector
    [log(alpha - lb) (K) | log(beta - lb) (V K, row-major) | simplex logits (N (K-1))] and the weights."""
    xt = torch.as_tensor(x, dtype=torch.float64)
    N, V = xt.shape
    a0t = a0 if torch.is_tensor(a0) else torch.as_tensor(np.broadcast_to(a0, (K,)).copy(), dtype=torch.float64)
    b0t = b0 if torch.is_tensor(b0) else torch.as_tensor(np.broadcast_to(b0, (V, K)).copy(), dtype=torch.float64)

    def dir_entropy(al):            # axis 0 is the Dirichlet dimension
        a_sum = al.sum(0)
        M = al.shape[0]
        return (torch.lgamma(al).sum(0) - torch.lgamma(a_sum) + (a_sum - M) * torch.digamma(a_sum)
                - ((al - 1.0) * torch.digamma(al)).sum(0))

    def f(theta, w):
        alpha = lb + torch.exp(theta[:K])
        beta = lb + torch.exp(theta[K:K + V * K]).reshape(V, K)
        logits = torch.cat([torch.zeros(N, 1, dtype=torch.float64), theta[K + V * K:].reshape(N, K - 1)], 1)
        z = torch.softmax(logits, 1)
        elog_pi = torch.digamma(alpha) - torch.digamma(alpha.sum())
        elog_phi = torch.digamma(beta) - torch.digamma(beta.sum(0, keepdim=True))
        s = elog_pi[None, :] + xt @ elog_phi
        lik = (w[:, None] * z * s).sum()
        ent_z = -(w[:, None] * z * torch.log(z)).sum()
        prior = ((a0t - 1.0) * elog_pi).sum() + ((b0t - 1.0) * elog_phi).sum()
        return -(lik + ent_z + prior + dir_entropy(alpha) + dir_entropy(beta).sum())
    return f


def make_hyper_objective(model, kind):
    """f(theta, eps) for an oracle.models.DeclaredModel with ONE of its hyper-parameters as a torch variable (the others
    frozen at the model's values): eps = tilt b | prior_mean m | prior_info diag(A) or vech(A) | quad_scale s | lik_info tau.
    Written independently of oracle/models.py so that torch.func AD of it pins the oracle's closed-form cross Hessians."""
    x = None if model.x is None else torch.tensor(model.x)
    y = None if model.y is None else torch.tensor(model.y)
    w = None if model.w is None else torch.tensor(model.w)
    A0 = None if model.quad_A is None else torch.tensor(model.quad_A)
    m0, b0 = torch.tensor(model.quad_m), torch.tensor(model.quad_b)
    V = model.layout.V
    tri = torch.tril_indices(V, V)

    def f(theta, eps):
        eta = constrain(theta, model.layout)
        tau = eps[0] if kind == 'lik_info' else model.lik_info
        val = torch.zeros((), dtype=eta.dtype)
        if model.loss:
            z = x @ eta[model.glm_off:model.glm_off + model.P]
            if model.loss == 1:
                l = 0.5 * tau * (y - z) ** 2
            elif model.loss == 2:
                l = torch.nn.functional.softplus(z) - y * z
            else:
                l = torch.exp(z) - y * z
            val = val + torch.sum(w * l)
        if A0 is not None:
            m = eps if kind == 'prior_mean' else m0
            b = eps if kind == 'tilt' else b0
            s = eps[0] if kind == 'quad_scale' else model.quad_scale
            A = A0
            if kind == 'prior_info':
                if A0.dim() == 1:
                    A = eps
                else:
                    L = torch.zeros(V, V, dtype=eps.dtype).index_put((tri[0], tri[1]), eps)
                    A = L + L.T - torch.diag(torch.diagonal(L))
            d = eta - m
            Ad = A * d if A.dim() == 1 else A @ d
            val = val + s * (0.5 * torch.dot(d, Ad) + torch.dot(b, eta))
        return val
    return f
