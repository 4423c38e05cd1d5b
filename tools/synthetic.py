"""Seeded synthetic problems shared by bench.py (--config c3 / c4) and the tests.  Data generators only: nothing here
imports the oracle, so the bench's non-baseline legs do not touch it even indirectly."""
import numpy as np


def clustered_problem(N, V, K, seed, a0=1.2, b0=0.9, sweeps=3, trials=20):
    """Clustered multinomial data and a few coordinate-ascent sweeps from a noisy version of the truth
    (vectorised; usable at N = 1e6).  Responsibilities saturate here (p down to 1e-30 and below), the
    regime the scaled local factorisation of the row kernel is written for."""
    from scipy import special
    rng = np.random.default_rng(seed)
    centers = rng.dirichlet(np.ones(V), size=K)
    lab = rng.integers(0, K, size=N)
    x = rng.multinomial(trials, centers[lab]).astype(np.float64)
    w = rng.uniform(0.5, 1.5, N)
    z = np.full((N, K), 0.5 / (K - 1)); z[np.arange(N), lab] = 0.5
    for _ in range(sweeps):
        alpha = a0 + (w[:, None] * z).sum(0)
        beta = b0 + x.T @ (w[:, None] * z)
        lam = np.vstack([special.digamma(alpha) - special.digamma(alpha.sum()),
                         special.digamma(beta) - special.digamma(beta.sum(0, keepdims=True))])
        s = lam[0][None, :] + x @ lam[1:]
        fz = s[:, 1:] - s[:, :1]
        z = np.exp(s - s.max(1, keepdims=True)); z /= z.sum(1, keepdims=True)
    fg = np.concatenate([np.log(alpha), np.log(beta).ravel()])
    return x, w, fg, fz, lam


def lmm_par(vb, p, G):
    """Parameter dictionary of the hierarchical LMM (doc/lmm.lyx shape): q(beta) MVN, q(mu) UVN, two Gamma precisions,
    G group effects."""
    par = vb.ModelParamsDict('params')
    par.push_param(vb.MVNParam('beta', dim=p))
    par.push_param(vb.UVNParam('mu'))
    par.push_param(vb.GammaParam('tau_y'))
    par.push_param(vb.GammaParam('tau_mu'))
    par.push_param(vb.UVNParamVector('u', length=G))
    return par
