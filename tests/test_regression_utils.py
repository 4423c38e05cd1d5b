"""regression helpers against the ridge closed forms the reference's tests use
(LRVB/test_regression.py:112-140 matmul/vecmul, :146-398 cases 1-5: posterior mean
(X^T tau X + Lambda0)^-1 (X^T tau y + Lambda0 mu0))."""
import numpy as np

import lrvb_amd as vb

ru = vb.regression_utils


def test_batched_products():
    rng = np.random.default_rng(43534543)
    a = rng.normal(size=(4, 3, 5, 6)); b = rng.normal(size=(4, 3, 6, 2))
    np.testing.assert_allclose(ru.mat_mul_last2dims(a, b), np.einsum('ijnm,ijmp->ijnp', a, b), rtol=1e-13)
    v = rng.normal(size=(4, 3, 6))
    np.testing.assert_allclose(ru.matvec_mul_last2dims(a, v), np.einsum('ijnm,ijm->ijn', a, v), rtol=1e-13)
    v2 = rng.normal(size=(4, 3, 7, 6))            # extra replicate dimension on y
    np.testing.assert_allclose(ru.matvec_mul_last2dims(a, v2), np.einsum('ijnm,ijrm->ijrn', a, v2), rtol=1e-13)


def test_ridge_closed_forms():
    rng = np.random.default_rng(1)
    n_t, r = 100, 4
    x = rng.normal(size=(n_t, r)); beta = rng.normal(size=r)
    y = x @ beta + 0.1 * rng.normal(size=n_t)
    tau = 3.0
    np.testing.assert_allclose(ru.get_regression_coefficients(y, x, tau), np.linalg.solve(x.T @ x, x.T @ y), rtol=1e-10)
    mu0 = rng.normal(size=r); a = rng.normal(size=(r, r)); lam0 = a @ a.T + np.eye(r)
    mean, info = ru.get_posterior_regression_coefficients(y, x, tau, mu0, lam0)
    np.testing.assert_allclose(info, tau * x.T @ x + lam0, rtol=1e-12)
    np.testing.assert_allclose(mean, np.linalg.solve(tau * x.T @ x + lam0, tau * x.T @ y + lam0 @ mu0), rtol=1e-10)
    # heteroskedastic info matrix and a batch of regressions
    c = rng.normal(size=(n_t, n_t)); infom = c @ c.T / n_t + np.eye(n_t)
    np.testing.assert_allclose(ru.get_regression_coefficients(y, x, infom),
                               np.linalg.solve(x.T @ infom @ x, x.T @ infom @ y), rtol=1e-9)
    xb = rng.normal(size=(5, n_t, r)); yb = np.einsum('bnr,r->bn', xb, beta)
    got = ru.get_regression_coefficients(yb, xb, tau)
    for i in range(5):
        np.testing.assert_allclose(got[i], beta, rtol=1e-8, atol=1e-10)
