"""Robust (Student-t) PSMF with the call surface of pypsmf/psmf/rpsmf.py.

`rPSMFIter` adds the per-step scale factors phi_k (dictionary covariance) and omega_k
(coefficient covariance, Q, R) and the growing degrees of freedom lambda_k to `PSMFIter`.
Same two back ends as psmf.py; on the device the scalars live in the serial stage of each step
(rpsmf_amd/csrc/psmf_kernels.hip).  `use_scaling` computes alpha, beta by the same KL
minimisation as the reference (mpmath, host, one-off).
"""

from collections import defaultdict

import numpy as np

from . import psmf as _psmf
from .psmf import PSMFIter, _StateDict

__all__ = ["rPSMFIter", "rPSMFIterMissing", "rPSMFRecursive"]


class rPSMFIter(PSMFIter):
    robust = True

    def __init__(self, theta0, C0, V0, mu0, P0, Q0, R0, lambda0, nonlinearity, fixed_lambda=False,
                 use_scaling=False, optim="adam", **kwargs):
        assert optim in ["adam", "sgd"]
        self.Q0 = Q0
        self.R0 = R0
        self.lambda0 = lambda0
        self.fixed_lambda = fixed_lambda
        self._alpha = 1.0
        self._beta = 1.0
        super().__init__(theta0, C0, V0, mu0, P0, {0: Q0}, {0: R0}, nonlinearity, optim=optim, **kwargs)
        self._lambda = defaultdict(lambda: lambda0) if fixed_lambda else {0: lambda0}
        if use_scaling:
            self._alpha = self.compute_scaling_factor(self._r * self._d, self._d)
            self._beta = self.compute_scaling_factor(self._r, self._d)

    def compute_scaling_factor(self, dim, offset, verbose=False):
        """alpha minimising KL( t_{lambda+offset}(0, I) || t_lambda(0, alpha I) ) in `dim` dimensions
        (rpsmf.py:75-104): stationary point of  B(m/2, (l+d)/2) log(alpha) + (1 + l/m) Q(alpha)."""
        import mpmath as mp

        m, lmd, d = mp.mpf(dim), mp.mpf(self.lambda0), mp.mpf(offset)

        def objective(alpha):
            def integrand(v):
                return (mp.power(v / (1 + v), m / 2) / (v * mp.power(1 + v, (lmd + d) / 2))
                        * mp.log(1 + (lmd + d) / (alpha * lmd) * v))

            return mp.beta(m / 2, (lmd + d) / 2) * mp.log(alpha) + (1 + lmd / m) * mp.quad(integrand, [0, mp.inf])

        return float(mp.findroot(lambda a: mp.diff(objective, a), mp.mpf(1.0), verbose=verbose))

    def step_reset(self):
        super().step_reset()
        self._lambda = defaultdict(lambda: self.lambda0) if self.fixed_lambda else {0: self.lambda0}
        self._R = {0: self.R0}
        self._Q = {0: self.Q0}

    # rPSMF reads the running Q_{k-1}, R_{k-1} (rpsmf.py:123,128,141)
    def _q_at(self, k):
        return self._Q[k - 1]

    def _r_index(self, k):
        return k - 1

    def _update_dictionary_covariance(self, k, Nk, mu_bar, yk):
        lam = self._lambda[k - 1]
        w = self._V[k - 1] @ mu_bar
        e = yk - self._y_pred[k]
        phi_k = float(np.squeeze(lam / (lam + self._d) + (e.T @ e) / ((lam + self._d) * Nk)))
        self._V[k] = self._alpha * phi_k * (self._V[k - 1] - (w @ w.T) / Nk)

    def _update_coefficient_covariance(self, k, Skinv, P_bar, yk):
        lam = self._lambda[k - 1]
        C = self._C[k - 1]
        e = yk - self._y_pred[k]
        omega_k = float(np.squeeze(lam + e.T @ (Skinv @ e))) / (lam + self._d)    # a Python float: R, Q may be scalars / vectors
        self._P[k] = self._beta * omega_k * (P_bar - P_bar @ (C.T @ (Skinv @ C)) @ P_bar)
        self._Q[k] = omega_k * self._Q[k - 1]
        self._R[k] = omega_k * self._R[k - 1]
        if not self.fixed_lambda:
            self._lambda[k] = lam + self._d

    def _grad_f(self, k, yk, eta_k, mu_prev, theta):
        """closed form of d/df of the t-likelihood of rpsmf.py:62-71"""
        lam = self._lambda[k - 1]
        f = self._nl(theta, mu_prev, k)
        C, V = self._C[k - 1], self._V[k - 1]
        u = V @ f
        N = float(np.squeeze(f.T @ u)) + float(np.squeeze(eta_k))
        e = yk - C @ f
        ee = float(np.squeeze(e.T @ e))
        D = lam * N
        return self._d * u / N + 0.5 * (self._d + lam) * (-2.0 * (C.T @ e) / D - 2.0 * lam * ee * u / D**2) / (1.0 + ee / D)

    # ---- device
    def _first_R(self):
        return self.R0

    def _device_lambda0(self):
        return float(self.lambda0)

    def _device_rho_q(self, T=None):
        # the epoch starts from R0, Q0 (step_reset) and runs on the omega-scaled Q_{k-1}, R_{k-1}: no schedules
        return self._rho_of(self.R0), self._q_matrix(self.Q0), None, None

    def _device_kwargs(self):
        kw = super()._device_kwargs()
        kw.update(fixed_lambda=self.fixed_lambda, alpha=self._alpha, beta=self._beta)
        return kw

    def _after_device_epoch(self, s, T):
        self._lambda = defaultdict(lambda: self.lambda0) if self.fixed_lambda else {T: s["lam"]}
        self._Q = {T: s["Q"]}
        # R_T = (product of the omega_k) R0 (rpsmf.py:169); with a uniform R0 the scalar stands for the multiple of the identity
        self._R = {T: s["rho"] if self._row_noise() is None else s["rho"] * np.asarray(self.R0, dtype=float)}


class rPSMFIterMissing(rPSMFIter):
    """Not usable in the reference either (singular S_k whenever a row is missing and a hook called
    with the wrong arity, rpsmf.py:187-287); the working masked semantics are those of
    ExperimentImpute, provided by rpsmf_amd.impute."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("use rpsmf_amd.impute.robust_PSMF (ExperimentImpute semantics)")


class rPSMFRecursive(rPSMFIter):
    run = _psmf.PSMFRecursive.run
    step = _psmf.PSMFRecursive.step
    inner = _psmf.PSMFRecursive.inner
    predict = _psmf.PSMFRecursive.predict
    _reset_gradient = _psmf.PSMFRecursive._reset_gradient
    _carry_theta = _psmf.PSMFRecursive._carry_theta
    _step_hip_recursive = _psmf.PSMFRecursive._step_hip_recursive

    def _host_stepped(self):
        return _psmf._recursive_host_stepped(self, rPSMFIter._host_stepped(self))

    def _device_kwargs(self):
        return _psmf._recursive_kwargs(self, rPSMFIter._device_kwargs(self))


_psmf.rPSMF_BASE = rPSMFIter
_psmf._BASE_CLASSES.extend([rPSMFIter, rPSMFIterMissing, rPSMFRecursive])
