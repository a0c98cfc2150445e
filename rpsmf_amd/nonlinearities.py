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

__all__ = ["BaseNonLinearity", "RandomWalk", "CosPhase", "ScaledWalk", "Sinusoid", "FourierBasis", "wrap_nonlinearity"]


def _col(v):
    return np.asarray(v).reshape(-1, 1)


class BaseNonLinearity:
    device_kind = None     # psmf_dyn_kind evaluated inside the device time loop, or None: host-stepped (psmf_step_host)
    device_flags = 0
    device_terms = 0

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


class ScaledWalk(BaseNonLinearity):
    """f = A x (+ b): pypsmf/psmf/nonlinearities.py:59-78.  theta packs [A (r*r, row-major)] [b (r) if bias].
    (The reference declares dims = ["r*r"] and then unpacks two blocks, so its own class raises whatever `bias` is and no
    experiment uses it; this one is the function its docstring describes.)"""

    device_kind = _capi.DYN_SCALED_WALK

    def __init__(self, rank, bias=True):
        super().__init__(rank)
        self.bias = bool(bias)
        self.device_flags = 1 if self.bias else 0

    @property
    def n_params(self):
        r = self.rank
        return r * r + (r if self.bias else 0)

    def _unpack(self, theta):
        r = self.rank
        th = np.asarray(theta).reshape(-1)
        return th[:r * r].reshape(r, r), (th[r * r:r * r + r].reshape(r, 1) if self.bias else 0.0)

    def __call__(self, theta, x, t):
        A, b = self._unpack(theta)
        return A @ _col(x) + b

    def jac_x(self, theta, x, t):
        return np.array(self._unpack(theta)[0], dtype=float)

    def jac_theta(self, theta, x, t):
        r = self.rank
        xx = np.asarray(x, dtype=float).reshape(-1)
        J = np.zeros((r, self.n_params))
        for i in range(r):
            J[i, i * r:(i + 1) * r] = xx            # d f_i / d A_ij = x_j
        if self.bias:
            J[:, r * r:] = np.eye(r)
        return J


class _TrigTerms(BaseNonLinearity):
    """Shared algebra of Sinusoid / FourierBasis: f = sum_t M_t trig_t(2 pi b_t t + c_t o x), M_t a matrix or the identity,
    c_t gains or ones -- the term structure of rpsmf_amd/csrc/psmf_dyn.hip.  Subclasses provide `_terms(theta)` ->
    [(M or None, b (r,), c (r,) or None, is_cos, offsets (m_off, b_off, c_off))]."""

    def _eval(self, theta, x, t):
        xx = np.asarray(x, dtype=float).reshape(-1)
        out = []
        for M, b, c, is_cos, offs in self._terms(theta):
            arg = 2.0 * np.pi * b * t + (c if c is not None else 1.0) * xx
            val, tp = (np.cos(arg), -np.sin(arg)) if is_cos else (np.sin(arg), np.cos(arg))
            out.append((M, b, c, val, tp, offs))
        return xx, out

    def jac_x(self, theta, x, t):
        r = self.rank
        _, terms = self._eval(theta, x, t)
        F = np.zeros((r, r))
        for M, b, c, val, tp, offs in terms:
            dv = tp * (c if c is not None else 1.0)
            F += (M if M is not None else np.eye(r)) * dv[None, :]
        return F

    def jac_theta(self, theta, x, t):
        r = self.rank
        xx, terms = self._eval(theta, x, t)
        J = np.zeros((r, self.n_params))
        for M, b, c, val, tp, (m_off, b_off, c_off) in terms:
            Mm = M if M is not None else np.eye(r)
            if M is not None:
                for i in range(r):
                    J[i, m_off + i * r:m_off + (i + 1) * r] = val      # d f_i / d M_ij = trig(arg_j)
            J[:, b_off:b_off + r] = Mm * (tp * 2.0 * np.pi * t)[None, :]
            if c is not None:
                J[:, c_off:c_off + r] = Mm * (tp * xx)[None, :]
        return J


class Sinusoid(_TrigTerms):
    """f = A sin(2 pi b t + c * x) with optional mixing matrix A (scaled) and gains c (phased).
    theta packs [A (r*r, row-major)] [b (r)] [c (r)] in that order."""

    device_kind = _capi.DYN_SINUSOID

    def __init__(self, rank, scaled=True, phased=True):
        super().__init__(rank)
        self.scaled, self.phased = bool(scaled), bool(phased)
        self.device_flags = (1 if self.scaled else 0) | (2 if self.phased else 0)

    def _terms(self, theta):
        r = self.rank
        th = np.asarray(theta, dtype=float).reshape(-1)
        o = r * r if self.scaled else 0
        A = th[:r * r].reshape(r, r) if self.scaled else None
        c = th[o + r:o + 2 * r] if self.phased else None
        return [(A, th[o:o + r], c, False, (0, o, o + r))]

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


class FourierBasis(_TrigTerms):
    """f = sum_n A_n sin(2 pi b_n t + c_n * x) + D_n cos(2 pi e_n t + f_n * x).

    theta packs [A_1, D_1, ..., A_N, D_N (r*r each)] then per n [b_n, c_n, e_n, f_n (r each)],
    the layout of pypsmf/psmf/nonlinearities.py:117-150.  (The reference broadcasts (r,) against
    (r,1) and therefore only works at r = 1, the ExperimentBeijing configuration; here the
    vectors are kept as columns, which is the same thing at r = 1 and well-defined for r > 1.)
    """

    device_kind = _capi.DYN_FOURIER

    def __init__(self, rank, N=1):
        super().__init__(rank)
        self.N = int(N)
        self.device_terms = self.N
        if self.N > 4:
            self.device_kind = None      # the device evaluates up to 4 sin + 4 cos terms; more: host-stepped

    @property
    def n_params(self):
        r = self.rank
        return self.N * (2 * r * r + 4 * r)

    def _terms(self, theta):
        r, N = self.rank, self.N
        th = np.asarray(theta, dtype=float).reshape(-1)
        out = []
        for t in range(2 * N):
            n, odd = t >> 1, t & 1
            m_off = t * r * r
            b_off = 2 * N * r * r + (4 * n + 2 * odd) * r
            out.append((th[m_off:m_off + r * r].reshape(r, r), th[b_off:b_off + r], th[b_off + r:b_off + 2 * r], bool(odd),
                        (m_off, b_off, b_off + r)))
        return out

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


def wrap_nonlinearity(fn, n_params, rank=None):
    """Give a plain callable f(theta, x, t) the derivative interface.  With `rank` (the device back end passes it): a
    callable that equals one of the closed-form families above on random probes is replaced by that family's object -- it is
    then evaluated inside the device time loop with analytic derivatives (modes.recognise_nonlinearity); any other callable
    is differentiated by complex step and host-stepped."""
    if isinstance(fn, BaseNonLinearity):
        return fn
    if rank is not None:
        from .modes import recognise_nonlinearity

        known = recognise_nonlinearity(fn, n_params, rank)
        if known is not None:
            known.recognised_from = fn
            return known
    return _Wrapped(fn, n_params)
