"""Trigamma and tetragamma of positive arguments, vectorised.

`scipy.special.polygamma(n, x)` evaluates a Hurwitz zeta function per element: ~0.35 us per element, which for the 1024
Dirichlet parameters of BASELINE.json's configuration 3 was 0.3-0.7 ms of host time in every Schur-complement build -- as
much as the whole device-side assembly.  Both functions here come from one pass: twelve steps of the recurrence
psi_n(x) = psi_n(x + 1) - (-1)^n n! / x^(n+1), then the asymptotic series at x + 12 >= 12 through the B_14 term (first
neglected term < 1e-17 relative).  Agreement with scipy: < 1e-15 relative on 1e-3 .. 1e6 (tests/test_host_logic.py).
"""
import numpy as np
from scipy import special

_SHIFT = np.arange(12.0)


def polygamma12(x):
    """(psi_1(x), psi_2(x)) elementwise; arguments <= 0 go to scipy."""
    x = np.asarray(x, dtype=np.float64)
    if x.size == 0 or not np.all(x > 0.0):
        return special.polygamma(1, x), special.polygamma(2, x)
    inv = 1.0 / (x[None, ...] + _SHIFT.reshape((12,) + (1,) * x.ndim))
    inv2 = inv * inv
    s2 = inv2[::-1].sum(axis=0)                                 # smallest terms first
    s3 = (inv2 * inv)[::-1].sum(axis=0)
    r = 1.0 / (x + 12.0)
    r2 = r * r
    # psi_1(y) ~ 1/y + 1/(2 y^2) + sum_k B_2k / y^(2k+1);  psi_2 is its derivative term by term
    p1 = r * (1.0 + r * (0.5 + r * (1.0 / 6 + r2 * (-1.0 / 30 + r2 * (1.0 / 42 + r2 * (-1.0 / 30 + r2 * (
        5.0 / 66 + r2 * (-691.0 / 2730 + r2 * (7.0 / 6)))))))))
    p2 = -r2 * (1.0 + r * (1.0 + r * (0.5 + r2 * (-1.0 / 6 + r2 * (1.0 / 6 + r2 * (-0.3 + r2 * (
        5.0 / 6 + r2 * (-691.0 / 210 + r2 * 17.5))))))))
    return p1 + s2, p2 - 2.0 * s3
