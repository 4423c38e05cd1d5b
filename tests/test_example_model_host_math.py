"""The N-independent closed forms of NormalRegressionObjective (product host math) against exact
AD of the restated Example.ipynb closure; the sufficient statistics are formed in numpy here, on
the GPU in tests/test_gpu_example.py."""
import numpy as np
import pytest
import torch

import lrvb_amd as vb
import torch_ref as tr
from oracle import example_model as oex


@pytest.mark.parametrize('dx,dy,N', [(1, 1, 40), (2, 2, 60), (3, 2, 50)])
def test_closed_forms_match_ad(dx, dy, N):
    rng = np.random.default_rng(dx * 10 + dy)
    x = rng.random((N, dx)); y = rng.normal(size=(N, dy)); w = rng.uniform(0.5, 1.5, N)
    lay = oex.layout(dx, dy)
    theta = rng.normal(size=lay.D) * 0.4
    eta = lay.constrain(theta)
    # a NormalRegressionObjective shell without a device context (only the host math is used)
    f = vb.NormalRegressionObjective.__new__(vb.NormalRegressionObjective)
    f.dx, f.dy, f.q = dx, dy, dx + dy
    f._bs, f._ls = range(0, dx * dy), range(dx * dy, lay.V)
    from lrvb_amd.quadform import duplication_matrix
    f._dup = duplication_matrix(dy)
    z = np.hstack([x, y])
    S = z.T @ (w[:, None] * z)
    val, g, H = f._terms(eta, S, float(w.sum()))
    assert abs(val - oex.objective_vec(eta, x, y, w)) < 1e-11 * max(1.0, abs(val))
    # derivatives in vector coordinates: identity packing in the torch restatement
    from oracle import packing as opk
    ident = opk.Layout([opk.box_block(dx * dy), opk.box_block(lay.V - dx * dy)])
    ft = tr.example_objective(ident, x, y)
    te, tw = torch.tensor(eta), torch.tensor(w)
    g_ad = torch.func.grad(ft)(te, tw).numpy()
    H_ad = torch.func.hessian(ft)(te, tw).numpy()
    np.testing.assert_allclose(g, g_ad, rtol=0, atol=1e-11 * np.max(np.abs(g_ad)))
    np.testing.assert_allclose(H, H_ad, rtol=0, atol=1e-11 * np.max(np.abs(H_ad)))
    # per-observation gradient rows: d/d eta of d f / d w_n
    M, c = f._obs_terms(eta)
    G = 0.5 * np.einsum('na,kab,nb->nk', z, M, z) + c[None, :]
    cross = torch.func.jacrev(torch.func.grad(ft, argnums=0), argnums=1)(te, tw).numpy()      # V x N
    np.testing.assert_allclose(G.T, cross, rtol=0, atol=1e-11 * np.max(np.abs(cross)))
    # and the oracle's free-coordinate objective agrees with the torch restatement
    ff = tr.example_objective(lay, x, y)
    assert abs(ff(torch.tensor(theta), tw).item() - oex.objective_free(theta, x, y, w)) < 1e-11 * abs(val)
