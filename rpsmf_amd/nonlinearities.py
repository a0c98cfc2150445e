"""State-transition functions f(theta, x, t) with analytic derivatives.

Call surface of pypsmf/psmf/nonlinearities.py (``__call__(theta, x, t)`` on (p,1) / (r,1)
column vectors, ``n_params``), plus what the filter needs without autograd:

    jac_x(theta, x, t)      -> (r, r)   d f / d x
    jac_theta(theta, x, t)  -> (r, p)   d f / d theta
    device_kind             -> which on-device dynamics (include/psmf_hip.h psmf_dyn_kind)
                               evaluates this function, or None (host-stepped only)

Arbitrary callables are wrapped by `wrap_nonlinearity`, which differentiates them by complex
step (exact to round-off for analytic numpy code) and falls back to central differences.
"""

import numpy as np

from . import _capi

__all__ = ["BaseNonLinearity", "RandomWalk", "CosPhase", "Sinusoid", "FourierBasis", "wrap_nonlinearity"]


def _col(v):
    return np.asarray(v).reshape(-1, 1)


class BaseNonLinearity:
    device_kind = None

    def __init__(self, rank):
        self.rank = int(rank)

    @property
    def n_params(self):
        return 0

    def __call__(self, theta, x, t):
        raise NotImplementedError

    # generic derivatives; subclasses override with closed forms
    def jac_x(self, theta, x, t):
        return _numeric_jac(lambda xx: self(theta, xx, t), np.asarray(x, dtype=float).reshape(-1))

    def jac_theta(self, theta, x, t):
        th = np.asarray(theta, dtype=float).reshape(-1)
        if th.size == 0:
            return np.zeros((np.asarray(x).size, 0))
        return _numeric_jac(lambda tt: self(tt, x, t), th)


def _numeric_jac(fn, v):
    """d fn / d v for fn: (n,) column-ish -> (r,1); complex step, else central differences."""
    n = v.size
    base = np.asarray(fn(_col(v))).reshape(-1)
    J = np.empty((base.size, n))
    try:
        h = 1e-30
        for j in range(n):
            vp = v.astype(complex)
            vp[j] += 1j * h
            out = np.asarray(fn(_col(vp))).reshape(-1)
            if not np.iscomplexobj(out):
                raise TypeError
            J[:, j] = out.imag / h
        return J
    except Exception:
        for j in range(n):
            h = 1e-6 * max(1.0, abs(v[j]))
            vp, vm = v.copy(), v.copy()
            vp[j] += h
            vm[j] -= h
            J[:, j] = (np.asarray(fn(_col(vp))).reshape(-1) - np.asarray(fn(_col(vm))).reshape(-1)) / (2 * h)
        return J


class RandomWalk(BaseNonLinearity):
    """f(x) = x; no parameters."""

    device_kind = _capi.DYN_RANDOM_WALK

    def __init__(self):
        super().__init__(0)

    def __call__(self, theta, x, t):
        return x

    def jac_x(self, theta, x, t):
        return np.eye(np.asarray(x).size)

    def jac_theta(self, theta, x, t):
        return np.zeros((np.asarray(x).size, 0))


class CosPhase(BaseNonLinearity):
    """f = cos(2 pi theta t + x) elementwise, theta in R^r: the dynamics of the synthetic
    experiments (ExperimentSynthetic/synthetic_psmf.py:105-106), evaluated on the device."""

    device_kind = _capi.DYN_COS_PHASE

    @property
    def n_params(self):
        return self.rank

    def __call__(self, theta, x, t):
        return np.cos(2.0 * np.pi * theta * t + x)

    def _arg(self, theta, x, t):
        return 2.0 * np.pi * np.asarray(theta).reshape(-1) * t + np.asarray(x).reshape(-1)

    def jac_x(self, theta, x, t):
        return np.diag(-np.sin(self._arg(theta, x, t)))

    def jac_theta(self, theta, x, t):
        return np.diag(-np.sin(self._arg(theta, x, t)) * (2.0 * np.pi * t))


class Sinusoid(BaseNonLinearity):
    """f = A sin(2 pi b t + c * x) with optional mixing matrix A (scaled) and gains c (phased).
    theta packs [A (r*r, row-major)] [b (r)] [c (r)] in that order."""

    def __init__(self, rank, scaled=True, phased=True):
        super().__init__(rank)
        self.scaled, self.phased = bool(scaled), bool(phased)

    @property
    def n_params(self):
        r = self.rank
        return (r * r if self.scaled else 0) + r + (r if self.phased else 0)

    def _unpack(self, theta):
        r = self.rank
        th = np.asarray(theta).reshape(-1)
        o = 0
        A = None
        if self.scaled:
            A = th[:r * r].reshape(r, r)
            o = r * r
        b = th[o:o + r].reshape(r, 1)
        c = th[o + r:o + 2 * r].reshape(r, 1) if self.phased else 1.0
        return A, b, c

    def __call__(self, theta, x, t):
        A, b, c = self._unpack(theta)
        s = np.sin(2.0 * np.pi * b * t + c * _col(x))
        return A @ s if self.scaled else s


class FourierBasis(BaseNonLinearity):
    """f = sum_n A_n sin(2 pi b_n t + c_n * x) + D_n cos(2 pi e_n t + f_n * x).

    theta packs [A_1, D_1, ..., A_N, D_N (r*r each)] then per n [b_n, c_n, e_n, f_n (r each)],
    the layout of pypsmf/psmf/nonlinearities.py:117-150.  (The reference broadcasts (r,) against
    (r,1) and therefore only works at r = 1, the ExperimentBeijing configuration; here the
    vectors are kept as columns, which is the same thing at r = 1 and well-defined for r > 1.)
    """

    def __init__(self, rank, N=1):
        super().__init__(rank)
        self.N = int(N)

    @property
    def n_params(self):
        r = self.rank
        return self.N * (2 * r * r + 4 * r)

    def __call__(self, theta, x, t):
        r, N = self.rank, self.N
        th = np.asarray(theta).reshape(-1)
        mats = th[:2 * N * r * r].reshape(2 * N, r, r)
        vecs = th[2 * N * r * r:].reshape(4 * N, r, 1)
        xx = _col(x)
        out = 0
        for n in range(N):
            b, c, e, f = vecs[4 * n], vecs[4 * n + 1], vecs[4 * n + 2], vecs[4 * n + 3]
            out = out + mats[2 * n] @ np.sin(2.0 * np.pi * b * t + c * xx)
            out = out + mats[2 * n + 1] @ np.cos(2.0 * np.pi * e * t + f * xx)
        return out


class _Wrapped(BaseNonLinearity):
    def __init__(self, fn, n_params):
        super().__init__(0)
        self._fn = fn
        self._n = int(n_params)

    @property
    def n_params(self):
        return self._n

    def __call__(self, theta, x, t):
        return self._fn(theta, x, t)


def wrap_nonlinearity(fn, n_params):
    """Give a plain callable f(theta, x, t) the derivative interface (complex step)."""
    if isinstance(fn, BaseNonLinearity):
        return fn
    return _Wrapped(fn, n_params)
