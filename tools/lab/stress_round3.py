"""One-off randomized stress of the kernels rewritten in round 3 against numpy / the oracle (test infrastructure: run through
gpurun; the fixed cases of the same comparisons live in tests/).  Shapes are drawn at random -- ragged tails, single rows, odd
widths -- to look for indexing mistakes the parametrised tests do not happen to hit."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'tools'))
import numpy as np
import lrvb_amd as vb
from oracle import mixture as om
from test_mixture_host_math import make_par, problem

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 12345)
def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

worst = {}
def note(k, v):
    worst[k] = max(worst.get(k, 0.0), v)

# 1. mixture rows kernel (two rows per wavefront, MFMA A_n, linear stores, hand-counted waits)
skipped = 0
for it in range(70):
    K = int(rng.integers(2, 33)); V = int(rng.integers(1, 32)); N = int(rng.choice([1, 2, 3, 5, 8, 9, 31, 64, 65, 127, 257, int(rng.integers(1, 400))]))
    x, w, theta = problem(N, V, K, seed=int(rng.integers(1, 10**6)))
    par = make_par(N, V, K)
    fun = vb.MixtureObjective(par, x, pi_prior=1.5, phi_prior=0.8)
    fun.weights_par.set_vector(w)
    fg, fz = theta[:fun.n_global], theta[fun.n_global:]
    _, _, lam = fun._lam(np.exp(fg))
    fun._push_state()
    o_val2, o_g, o_H, o_S64, o_R = om.mixture_rows(fz, x, w, lam)
    pd = all(np.all(np.linalg.eigvalsh(h) > 0) for h in o_H)
    try:
        val2, gz, S64, Rm = fun.ctx.mixture_rows(K, fz, lam)
    except np.linalg.LinAlgError:
        assert not pd, 'the device reported an indefinite block where the oracle has none'
        skipped += 1
        continue
    assert pd, 'the oracle has an indefinite block the device did not report'
    note('rows val', rel(val2, o_val2)); note('rows grad', rel(gz, o_g)); note('rows S64', rel(S64, o_S64)); note('rows R', rel(Rm, o_R))
    fun.ctx.set_tuning(0, 2)
    note('rows R dense path', rel(fun.ctx.mixture_rows(K, fz, lam)[3], o_R))
    fun.ctx.set_tuning(0, 0)
    del fun
# one larger odd size: several grid-stride rounds per wavefront, the last pair half empty
for (N, V, K) in ((20001, 31, 32), (16385, 7, 9)):
    from test_mixture_host_math import near_optimum_problem
    x, w, theta = near_optimum_problem(N, V, K, seed=5)
    par = make_par(N, V, K)
    fun = vb.MixtureObjective(par, x, pi_prior=1.5, phi_prior=0.8)
    fun.weights_par.set_vector(w)
    fg, fz = theta[:fun.n_global], theta[fun.n_global:]
    _, _, lam = fun._lam(np.exp(fg))
    fun._push_state()
    val2, gz, S64, Rm = fun.ctx.mixture_rows(K, fz, lam)
    o_val2, o_g, _, o_S64, o_R = om.mixture_rows(fz, x, w, lam)
    note('rows val', rel(val2, o_val2)); note('rows grad', rel(gz, o_g)); note('rows S64', rel(S64, o_S64)); note('rows R', rel(Rm, o_R))
    del fun
print('mixture rows: ', {k: '%.1e' % v for k, v in worst.items()}, 'indefinite points skipped:', skipped, flush=True)
assert worst['rows val'] < 1e-11 and worst['rows grad'] < 1e-10 and worst['rows S64'] < 1e-11 and worst['rows R'] < 1e-8 and worst['rows R dense path'] < 1e-8

# 2. Cholesky / triangular solves / LRVB covariance (fused diagonal block, fused solve steps)
blocks = [dict(kind=0, free_size=4, vec_size=4, dim0=4, dim1=0, lb=-np.inf, ub=np.inf)]
ctx = vb.DeviceContext(blocks, quad_kind=1)
w2 = {}
for it in range(45):
    n = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 320, int(rng.integers(1, 900))]))
    q = int(rng.choice([1, 2, 17, 63, 64, 65, 130, int(rng.integers(1, 300))]))
    A = rng.normal(size=(n, n + 3))
    H = A @ A.T / n + 0.5 * np.eye(n)
    B = rng.normal(size=(n, q)); M = rng.normal(size=(q, n))
    ctx.chol_factor(H)
    w2['solve'] = max(w2.get('solve', 0), rel(ctx.chol_solve(B), np.linalg.solve(H, B)))
    w2['cov'] = max(w2.get('cov', 0), rel(ctx.lrvb_cov(M), M @ np.linalg.solve(H, M.T)))
print('cholesky:     ', {k: '%.1e' % v for k, v in w2.items()}, flush=True)
assert w2['solve'] < 1e-9 and w2['cov'] < 1e-9

# 3. narrow weighted Gram (unconditional prefetch, aligned / unaligned instantiations)
w3 = 0.0
for it in range(40):
    N = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 1000, 4097, int(rng.integers(1, 20000))])); q = int(rng.integers(1, 65))
    Z = rng.normal(size=(N, q)); w = rng.uniform(0.1, 2.0, size=N)
    c2 = vb.DeviceContext([dict(kind=0, free_size=q, vec_size=q, dim0=q, dim1=0, lb=-np.inf, ub=np.inf)], loss='data_only', n_obs=N, n_cols=q)
    c2.set_data(vb._hip.SLOT_X, Z); c2.set_weights(w)
    w3 = max(w3, rel(c2.weighted_gram(), (Z.T * w) @ Z))
    del c2
print('narrow Gram:   %.1e' % w3, flush=True)
assert w3 < 1e-12
print('stress ok')
