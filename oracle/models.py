"""Oracle: the declared objective and its exact derivatives, plain numpy fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).

  f(eta) = sum_n w_n loss(y_n, x_n . eta[off:off+P])  +  s * ( 1/2 (eta-m)^T A (eta-m) + b^T eta )
  f_free(theta) = f(eta(theta))

What the reference computes for such a model (LRVB/SparseObjectives.py):
  Objective.fun_free / fun_vector            :120-129   value
  fun_free_grad / fun_vector_grad            :152-154, 164-166
  fun_free_hessian (autograd.hessian, :103)  :156-158   -> hessian()
  fun_free_hvp (hessian_vector_product)      :105-106, 183-187 -> hvp()
  TwoParameterObjective.fun_hessian_free1_vector2  :429-438 with hyper-parameter = weights
      (Example.ipynb:425-441) -> obs_grad() is its transpose; with hyper-parameter = the tilt b
      (LRVB/test_model_sensitivity.py:56-66) -> cross_hessian_tilt()
The free-coordinate Hessian uses convert_vector_to_free_hessian (LRVB/Parameters.py:397-424).

`hessian_by_hvps` restates the COST STRUCTURE of autograd.hessian -- one gradient plus D
Hessian-vector products against the standard basis, each a full pass over the observations --
and is what bench.py times as the reference-faithful CPU baseline.
"""
import numpy as np
from . import packing

GAUSSIAN, LOGISTIC, POISSON = 1, 2, 3


def loss_terms(loss, y, z, lik_info=1.0):
    """(loss, dloss/dz, d2loss/dz2) elementwise."""
    if loss == GAUSSIAN:
        d = z - y
        return 0.5 * lik_info * d * d, lik_info * d, np.full_like(z, lik_info)
    if loss == LOGISTIC:
        sp = np.logaddexp(0.0, z)
        ez = np.exp(-np.abs(z))
        sig = np.where(z >= 0, 1.0 / (1.0 + ez), ez / (1.0 + ez))
        return sp - y * z, sig - y, sig * (1.0 - sig)
    if loss == POISSON:
        ez = np.exp(z)
        return ez - y * z, ez - y, ez
    raise ValueError('unknown loss')


def sigmoid_derivative_polys(max_order):
    """sigma^(m)(z) as a polynomial in s = sigma(z), ascending coefficients, m = 1..max_order:
    P_1 = s - s^2 and, by the chain rule, P_(m+1) = P_m'(s) (s - s^2)."""
    from numpy.polynomial import polynomial as Pl
    polys = [None, np.array([0.0, 1.0, -1.0])]
    for m in range(1, max_order):
        polys.append(Pl.polymul(Pl.polyder(polys[m]), np.array([0.0, 1.0, -1.0])))
    return polys


def loss_derivative(loss, m, y, z, lik_info=1.0):
    """d^m loss / d z^m elementwise, m >= 1 (what m nested JVPs of the loss along z give)."""
    from numpy.polynomial import polynomial as Pl
    if m <= 2:
        return loss_terms(loss, y, z, lik_info)[m]
    if loss == GAUSSIAN:
        return np.zeros_like(z)
    if loss == POISSON:
        return np.exp(z)
    if loss == LOGISTIC:                       # loss' = sigma - y  ->  loss^(m) = sigma^(m-1)
        ez = np.exp(-np.abs(z))
        sig = np.where(z >= 0, 1.0 / (1.0 + ez), ez / (1.0 + ez))
        return Pl.polyval(sig, sigmoid_derivative_polys(m - 1)[m - 1])
    raise ValueError('unknown loss')


class DeclaredModel(object):
    def __init__(self, layout, loss=0, x=None, y=None, w=None, glm_off=0, lik_info=1.0,
                 quad_A=None, quad_m=None, quad_b=None, quad_scale=1.0):
        self.layout = layout
        self.loss = loss
        self.x = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self.y = None if y is None else np.asarray(y, dtype=np.float64).ravel()
        if self.x is not None:
            self.N, self.P = self.x.shape
            self.w = np.ones(self.N) if w is None else np.asarray(w, dtype=np.float64).ravel()
        else:
            self.N = self.P = 0
            self.w = None
        self.glm_off = glm_off
        self.lik_info = lik_info
        V = layout.V
        self.quad_A = None if quad_A is None else np.asarray(quad_A, dtype=np.float64)
        self.quad_m = np.zeros(V) if quad_m is None else np.asarray(quad_m, dtype=np.float64)
        self.quad_b = np.zeros(V) if quad_b is None else np.asarray(quad_b, dtype=np.float64)
        self.quad_scale = quad_scale

    # ---- vector coordinates -----------------------------------------------------------
    def _A_apply(self, u):
        if self.quad_A is None:
            return np.zeros_like(u)
        return self.quad_A * u if self.quad_A.ndim == 1 else self.quad_A @ u

    def _A_dense(self):
        V = self.layout.V
        if self.quad_A is None:
            return np.zeros((V, V))
        return np.diag(self.quad_A) if self.quad_A.ndim == 1 else self.quad_A

    def _beta(self, eta):
        return eta[self.glm_off:self.glm_off + self.P]

    def value_vec(self, eta):
        eta = np.asarray(eta, dtype=np.float64)
        val = 0.0
        if self.loss:
            z = self.x @ self._beta(eta)
            val += float(np.sum(self.w * loss_terms(self.loss, self.y, z, self.lik_info)[0]))
        if self.quad_A is not None:
            d = eta - self.quad_m
            val += self.quad_scale * (0.5 * float(d @ self._A_apply(d)) + float(self.quad_b @ eta))
        return val

    def grad_vec(self, eta):
        eta = np.asarray(eta, dtype=np.float64)
        g = np.zeros(self.layout.V)
        if self.loss:
            z = self.x @ self._beta(eta)
            l1 = loss_terms(self.loss, self.y, z, self.lik_info)[1]
            g[self.glm_off:self.glm_off + self.P] += self.x.T @ (self.w * l1)
        if self.quad_A is not None:
            g += self.quad_scale * (self._A_apply(eta - self.quad_m) + self.quad_b)
        return g

    def hessian_vec(self, eta):
        eta = np.asarray(eta, dtype=np.float64)
        V = self.layout.V
        H = np.zeros((V, V))
        if self.loss:
            z = self.x @ self._beta(eta)
            l2 = loss_terms(self.loss, self.y, z, self.lik_info)[2]
            s = slice(self.glm_off, self.glm_off + self.P)
            H[s, s] += self.x.T @ ((self.w * l2)[:, None] * self.x)
        if self.quad_A is not None:
            H += self.quad_scale * self._A_dense()
        return H

    def hvp_vec(self, eta, u):
        eta = np.asarray(eta, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        out = np.zeros(self.layout.V)
        if self.loss:
            z = self.x @ self._beta(eta)
            l2 = loss_terms(self.loss, self.y, z, self.lik_info)[2]
            t = self.x @ u[self.glm_off:self.glm_off + self.P]
            out[self.glm_off:self.glm_off + self.P] += self.x.T @ (self.w * l2 * t)
        if self.quad_A is not None:
            out += self.quad_scale * self._A_apply(u)
        return out

    def dk_grad_vec(self, eta, U=None, w_override=None, include_quad=True):
        """D^j g [u_1 .. u_j] of the vector-coordinate gradient g = grad_vec along the rows of U (j x V); j = 0 is g.
        The reference gets this from j nested autograd JVPs of the gradient closure (LRVB/ModelSensitivity.py:38-62,
        221-234); here the closed form: X^T (w o loss^(j+1)(z) o prod_k (X u_k)), plus s A u_1 when j = 1."""
        eta = np.asarray(eta, dtype=np.float64)
        U = np.zeros((0, self.layout.V)) if U is None else np.asarray(U, dtype=np.float64).reshape(-1, self.layout.V)
        j = U.shape[0]
        w = self.w if w_override is None else np.asarray(w_override, dtype=np.float64)
        out = np.zeros(self.layout.V)
        s = slice(self.glm_off, self.glm_off + self.P)
        if self.loss:
            z = self.x @ self._beta(eta)
            coef = w * loss_derivative(self.loss, j + 1, self.y, z, self.lik_info)
            for k in range(j):
                coef = coef * (self.x @ U[k, s])
            out[s] += self.x.T @ coef
        if include_quad and self.quad_A is not None:
            if j == 0:
                out += self.quad_scale * (self._A_apply(eta - self.quad_m) + self.quad_b)
            elif j == 1:
                out += self.quad_scale * self._A_apply(U[0])
        return out

    # ---- free coordinates ---------------------------------------------------------------
    def value(self, theta):
        return self.value_vec(self.layout.constrain(theta))

    def grad(self, theta):
        eta = self.layout.constrain(theta)
        return self.layout.jac(theta).T @ self.grad_vec(eta)

    def hessian(self, theta):
        eta = self.layout.constrain(theta)
        return packing.convert_vector_to_free_hessian(
            self.layout, theta, self.grad_vec(eta), self.hessian_vec(eta))

    def hvp(self, theta, v):
        eta = self.layout.constrain(theta)
        J = self.layout.jac(theta)
        T = self.layout.third_order(theta, self.grad_vec(eta))
        return J.T @ self.hvp_vec(eta, J @ np.asarray(v, dtype=np.float64)) + T @ v

    def obs_grad(self, theta, n0=0, n1=None):
        """Rows of G: G[n, :] = d/dtheta (d f / d w_n) = loss'(y_n, z_n) x_n^T J_slice."""
        n1 = self.N if n1 is None else n1
        eta = self.layout.constrain(theta)
        z = self.x[n0:n1] @ self._beta(eta)
        l1 = loss_terms(self.loss, self.y[n0:n1], z, self.lik_info)[1]
        Jg = self.layout.jac(theta)[self.glm_off:self.glm_off + self.P, :]
        return (l1[:, None] * self.x[n0:n1]) @ Jg

    def obs_grad_vec(self, eta, n0=0, n1=None):
        """Rows of G in vector coordinates ((n1 - n0) x V): loss' x_n^T in the coefficient slice."""
        n1 = self.N if n1 is None else n1
        eta = np.asarray(eta, dtype=np.float64)
        z = self.x[n0:n1] @ self._beta(eta)
        l1 = loss_terms(self.loss, self.y[n0:n1], z, self.lik_info)[1]
        G = np.zeros((n1 - n0, self.layout.V))
        G[:, self.glm_off:self.glm_off + self.P] = l1[:, None] * self.x[n0:n1]
        return G

    def gram(self, theta):
        G = self.obs_grad(theta)
        return G.T @ G

    def cross_hessian_tilt(self, theta):
        """d2 f / d theta d b^T = s J^T  (D x V)."""
        return self.quad_scale * self.layout.jac(theta).T

    # ---- the other hyper-parameters (LRVB/ModelSensitivity.py:555-612 takes any hyper_par) -------------------------
    # eps in {'tilt': b, 'prior_mean': m, 'prior_info': diag(A) or vech(A) (row-major lower triangle,
    # LRVB/MatrixParameters.py:16-41), 'quad_scale': s, 'lik_info': tau of the Gaussian loss}, each in its vector
    # coordinates.  What `jacobian(grad_1, argnum=hyper)` / `grad(argnum=hyper)` of LRVB/SparseObjectives.py:333-339,
    # 381-387 return for this objective, written out.
    def hyper_value(self, kind):
        if kind == 'tilt':
            return self.quad_b.copy()
        if kind == 'prior_mean':
            return self.quad_m.copy()
        if kind == 'prior_info':
            A = self.quad_A
            return A.copy() if A.ndim == 1 else A[np.tril_indices(A.shape[0])].copy()
        if kind == 'quad_scale':
            return np.array([float(self.quad_scale)])
        if kind == 'lik_info':
            return np.array([float(self.lik_info)])
        raise ValueError('unknown hyper-parameter ' + str(kind))

    def set_hyper(self, kind, val):
        val = np.asarray(val, dtype=np.float64).ravel()
        if kind == 'tilt':
            self.quad_b = val.copy()
        elif kind == 'prior_mean':
            self.quad_m = val.copy()
        elif kind == 'prior_info':
            if self.quad_A.ndim == 1:
                self.quad_A = val.copy()
            else:
                V = self.layout.V
                L = np.zeros((V, V))
                L[np.tril_indices(V)] = val
                self.quad_A = L + L.T - np.diag(np.diag(L))
        elif kind == 'quad_scale':
            self.quad_scale = float(val[0])
        elif kind == 'lik_info':
            self.lik_info = float(val[0])
        else:
            raise ValueError('unknown hyper-parameter ' + str(kind))

    def hyper_grad_vec(self, kind, eta):
        """d f / d eps at the vector-coordinate point eta."""
        eta = np.asarray(eta, dtype=np.float64)
        s, r = self.quad_scale, eta - self.quad_m
        if kind == 'tilt':
            return s * eta
        if kind == 'prior_mean':
            return -s * self._A_apply(r)
        if kind == 'prior_info':
            if self.quad_A.ndim == 1:
                return 0.5 * s * r * r
            i, j = np.tril_indices(self.layout.V)
            return s * r[i] * r[j] * np.where(i == j, 0.5, 1.0)
        if kind == 'quad_scale':
            return np.array([0.5 * float(r @ self._A_apply(r)) + float(self.quad_b @ eta)])
        if kind == 'lik_info':
            z = self.x @ self._beta(eta)
            return np.array([float(np.sum(self.w * loss_terms(self.loss, self.y, z, self.lik_info)[0])) / self.lik_info])
        raise ValueError('unknown hyper-parameter ' + str(kind))

    def cross_hessian_hyper_vec(self, kind, eta):
        """d2 f / d eta d eps^T (V x Ph)."""
        eta = np.asarray(eta, dtype=np.float64)
        V, s, r = self.layout.V, self.quad_scale, eta - self.quad_m
        if kind == 'tilt':
            return s * np.eye(V)
        if kind == 'prior_mean':
            return -s * self._A_dense()
        if kind == 'prior_info':
            if self.quad_A.ndim == 1:
                return s * np.diag(r)
            i, j = np.tril_indices(V)
            C = np.zeros((V, i.size))
            cols = np.arange(i.size)
            C[i, cols] += s * r[j]
            off = i != j
            C[j[off], cols[off]] += s * r[i[off]]
            return C
        if kind == 'quad_scale':
            return (self._A_apply(r) + self.quad_b)[:, None]
        if kind == 'lik_info':
            z = self.x @ self._beta(eta)
            g = np.zeros(V)
            g[self.glm_off:self.glm_off + self.P] = self.x.T @ (self.w * loss_terms(self.loss, self.y, z, self.lik_info)[1])
            return (g / self.lik_info)[:, None]
        raise ValueError('unknown hyper-parameter ' + str(kind))

    def hyper_grad(self, kind, theta):
        return self.hyper_grad_vec(kind, self.layout.constrain(theta))

    def cross_hessian_hyper(self, kind, theta):
        """d2 f / d theta d eps^T = J^T (d2 f / d eta d eps^T): the objective is differentiated once in each argument, so no
        second-order packing term enters."""
        return self.layout.jac(theta).T @ self.cross_hessian_hyper_vec(kind, self.layout.constrain(theta))

    # ---- reference-faithful cost structure -------------------------------------------------
    def hessian_by_hvps(self, theta, n_columns=None):
        """autograd.hessian = jacobian(jacobian(f)): one reverse pass per gradient component.
        Each column costs a forward + reverse-over-reverse sweep over all N observations; here
        one column = one analytic HVP (two X passes), which is the cheapest a tape walk can be.
        n_columns < D evaluates a prefix (bench.py extrapolates linearly)."""
        D = self.layout.D
        ncol = D if n_columns is None else int(n_columns)
        eta = self.layout.constrain(theta)
        J = self.layout.jac(theta)
        T = self.layout.third_order(theta, self.grad_vec(eta))
        H = np.empty((D, ncol))
        e = np.zeros(D)
        for j in range(ncol):
            e[j] = 1.0
            H[:, j] = J.T @ self.hvp_vec(eta, J @ e) + T @ e
            e[j] = 0.0
        return H
