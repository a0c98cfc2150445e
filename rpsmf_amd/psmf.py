"""PSMF filter classes with the call surface of pypsmf (`PSMFIter`, `PSMFRecursive`).

Same constructor, `run / step / inner / predict / optim_* / adam_* / sgd_*` methods, the same
overridable hook names and the same state attributes (`_C, _V, _mu, _P, _theta, _y_pred, _Q, _R,
_gradsum`, `C0, theta0, V0, _d, _r`) as pypsmf/psmf/psmf.py, so experiment subclasses written
against the reference (and its TrackingMixin) run unchanged.  Two execution back ends:

backend="hip"    (default) the whole `for k in 1..T: inner(...)` loop runs on the MI355X through
                 libpsmf_hip.so (include/psmf_hip.h).  Only *recognised* hook configurations can
                 be fused: the class attribute `hip_mode` names one ("full" = the hooks as
                 shipped; "simplified" = the ExperimentSynthetic overrides: P_bar = P_{k-1},
                 eta = tr(R)/d, no coefficient update).  A subclass that overrides compute hooks
                 must declare its `hip_mode`; otherwise the constructor raises.  There is no
                 silent CPU fallback: if the library or the GPU is missing, it raises.
backend="numpy"  the hooks below are executed one by one on the host, exactly as the
                 reference does, but in O(d r^2): no d x d matrix is ever formed (diagonal R,
                 Woodbury in r x r form, S^-1 as an implicit operator).  This is the plumbing
                 path for arbitrary hook overrides and arbitrary `nonlinearity` callables.

Deviations from the reference, all deliberate: `Rs[k]` / `Qs[k]` may be scalars or (d,) vectors
in addition to dense matrices (dense d x d is unusable at d >= 10^4); derivatives of the
nonlinearity are analytic / complex-step instead of autograd.
"""

import numpy as np

from . import _capi
from .learning_rate import BaseLearningRate, ConstantLearningRate
from .linop import InverseInnovation
from .nonlinearities import BaseNonLinearity, wrap_nonlinearity

__all__ = ["PSMFIter", "PSMFIterMissing", "PSMFRecursive"]

COMPUTE_HOOKS = (
    "_predictive_mean", "_predictive_covariance", "_predict_measurement", "_compute_eta_k",
    "_compute_dictionary_innovation", "_update_dictionary_mean", "_update_dictionary_covariance",
    "_compute_inverse_coefficient_innovation", "_update_coefficient_mean",
    "_update_coefficient_covariance", "_store_gradient",
)

HIP_MODES = {
    # name: (coef_update, eta_full, pbar_predict)
    "full": (True, True, True),
    "simplified": (False, False, False),
}
# compute hooks a class may override under each declared mode (the overrides ARE the mode: synthetic_psmf.py:82-98,
# synthetic_rpsmf.py:86-114); anything else would be skipped silently by the fused device loop
HIP_MODE_HOOKS = {
    "full": frozenset(),
    "simplified": frozenset({"_predictive_covariance", "_compute_eta_k", "_compute_inverse_coefficient_innovation",
                             "_update_coefficient_mean", "_update_coefficient_covariance"}),
}
# drive-level methods the fused loop replaces as a whole
FUSED_METHODS = ("inner", "_store_gradient", "_set_gradient", "_reset_gradient", "_carry_theta")


class _Lazy:
    """A value that lives on the device until somebody asks for it."""

    def __init__(self, owner, name, epoch):
        self.owner, self.name, self.epoch = owner, name, epoch

    def fetch(self):
        return self.owner._fetch_device(self.name, self.epoch)


class _StateDict(dict):
    """dict k -> array whose values may be device-resident (_Lazy) until first read."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, _Lazy):
            v = v.fetch()
            dict.__setitem__(self, k, v)
        return v

    def raw(self, k):
        return dict.__getitem__(self, k)


class _MuHist(_StateDict):
    """`_mu[k]` -> (r, 1) for k = 0..T after a device epoch: served from the device's mean history
    (the reference only keeps them when the experiment's `_prune` override spares `_mu`)."""

    def __init__(self, owner, T, last):
        super().__init__({T: last})
        self._owner, self._T, self._host = owner, T, None

    def __missing__(self, k):
        if 0 <= k <= self._T:
            if self._host is None:
                self._host = self._owner._dev.mu_history(0, self._T + 1)
            return self._host[k].reshape(-1, 1)
        raise KeyError(k)

    def __contains__(self, k):
        return 0 <= k <= self._T


class _YPred(dict):
    """`_y_pred[k]` -> (d, 1).  Steps 1..T are served from the device buffer (fetched once, whole)."""

    def __init__(self, owner=None, T=0):
        super().__init__()
        self._owner, self._T, self._host = owner, T, None

    def _block(self):
        if self._host is None:
            self._host = self._owner._dev.y_pred(0, self._T)
        return self._host

    def __missing__(self, k):
        if self._owner is not None and 1 <= k <= self._T:
            return self._block()[k - 1].reshape(-1, 1)
        raise KeyError(k)

    def __contains__(self, k):
        return dict.__contains__(self, k) or (self._owner is not None and 1 <= k <= self._T)


def _diag_of(R, d):
    """diag(R) as scalar or (d,) vector, or None if R is a non-diagonal d x d matrix.  Accepted shapes: scalar, (1,),
    (1, 1), (d,), (d, 1), (1, d), (d, d); anything else raises (a silently broadcast R would give wrong numbers)."""
    if np.ndim(R) == 0:
        return float(R)
    R = np.asarray(R)
    if R.size == 1:
        return float(R.reshape(()))
    if R.shape in ((d,), (d, 1), (1, d)):
        return R.reshape(-1).astype(float)
    if R.shape != (d, d):
        raise ValueError(f"R has shape {R.shape}: expected a scalar, a (d,) diagonal or a (d, d) matrix with d = {d}")
    dg = np.diagonal(R)
    if np.count_nonzero(R) != np.count_nonzero(dg):
        return None
    return dg.astype(float)


def _content_hash(a):
    try:
        import xxhash

        return xxhash.xxh3_128_hexdigest(a.data)
    except ImportError:
        import hashlib

        return hashlib.blake2b(a.data, digest_size=16).hexdigest()


def _recognise_default():
    import os

    return os.environ.get("PSMF_RECOGNISE", "1").strip().lower() not in ("0", "false", "off", "no")


def _as_scalar_if_uniform(rho):
    if np.ndim(rho) == 0:
        return float(rho)
    rho = np.asarray(rho)
    return float(rho[0]) if np.all(rho == rho[0]) else None


class PSMFIter:
    """Iterative (epochs over a fixed series) PSMF.  See the module docstring."""

    hip_mode = "full"
    robust = False

    def __init__(self, theta0, C0, V0, mu0, P0, Qs, Rs, nonlinearity, optim="adam", backend="hip",
                 device=0, storage="auto", gram_refresh=0, engine="auto", recognise=None):
        assert optim in ["adam", "sgd"]
        if backend not in ("hip", "numpy"):
            raise ValueError("backend must be 'hip' or 'numpy'")
        self.optim = optim
        self.backend = backend
        self.theta0 = theta0
        self.C0 = C0
        self.V0 = V0
        self.mu0 = mu0
        self.P0 = P0
        self._d, self._r = C0.shape
        self.nonlinearity = nonlinearity
        # `recognise` (default: on, unless the environment says PSMF_RECOGNISE=0): may the constructor work out by PROBING what
        # an undeclared hook override / a plain-function nonlinearity computes (modes.py)?  False = the strict behaviour:
        # undeclared overrides raise TypeError, plain functions are host-stepped (correct for any callable).
        self.recognise = _recognise_default() if recognise is None else bool(recognise)
        # (backend "hip": a plain function that IS one of the closed-form families runs inside the device loop, modes.py)
        self._nl = wrap_nonlinearity(nonlinearity, np.asarray(theta0).size,
                                     rank=self._r if (backend == "hip" and self.recognise) else None)
        self._C = _StateDict()
        self._P = _StateDict()
        self._V = _StateDict()
        self._mu = _StateDict()
        self._Q = Qs
        self._R = Rs
        self._theta = {0: theta0}
        self._y_pred = {}
        self._gradsum = np.zeros(np.asarray(theta0).shape)
        self._dev = None
        self._dev_opts = dict(device=device, storage=storage, gram_refresh=gram_refresh, engine=engine)
        self._dev_epoch = 0
        self._series_key = None
        if backend == "hip":
            self._check_hip_configuration()

    # ------------------------------------------------------------------ drive methods
    def run(self, y, T, n_iter, n_pred):
        self.optim_init()
        for i in range(1, n_iter + 1):
            self.step(y, i, T)
            self.predict(i, T, n_pred)
            self.optim_update(i)

    def step_reset(self):
        """Epoch start: carry the last C, mu, P, V (or the initial values); zero the gradient."""
        def latest(D, init):
            if not D:
                return init
            k = max(D.keys())
            return D.raw(k) if isinstance(D, _StateDict) else D[k]

        self._C = _StateDict({0: latest(self._C, self.C0)})
        self._mu = _StateDict({0: latest(self._mu, self.mu0)})
        self._P = _StateDict({0: latest(self._P, self.P0)})
        self._V = _StateDict({0: latest(self._V, self.V0)})
        self._gradsum = np.zeros(np.asarray(self.theta0).shape)

    def step(self, y, i, T):
        if self.backend == "hip":
            return self._step_hip(y, i, T)
        self.step_reset()
        for k in range(1, T + 1):
            self.inner(i, k, np.asarray(y[k]).reshape(-1, 1))     # y[k], k = 1..T: dict or array-like (row 0 unused)

    def inner(self, i, k, yk):
        mu_bar = self._predictive_mean(i, k)
        P_bar = self._predictive_covariance(i, k)
        self._y_pred[k] = self._predict_measurement(k, mu_bar)
        eta_k = self._compute_eta_k(k, P_bar)
        Nk = self._compute_dictionary_innovation(k, eta_k, mu_bar, P_bar)
        self._update_dictionary_mean(k, yk, Nk, mu_bar)
        self._update_dictionary_covariance(k, Nk, mu_bar, yk)
        Skinv = self._compute_inverse_coefficient_innovation(k, mu_bar, P_bar)
        self._update_coefficient_mean(k, yk, Skinv, mu_bar, P_bar)
        self._update_coefficient_covariance(k, Skinv, P_bar, yk)
        self._store_gradient(i, k, yk, eta_k)
        self._prune(k)

    # ------------------------------------------------------------------ hooks (numpy backend)
    def _q_at(self, k):
        return self._Q[k]

    def _r_index(self, k):
        return k

    def _rho_at(self, k):
        """diag(R_k) as (d,) vector; raises for a non-diagonal R (use the dense route)."""
        dg = _diag_of(self._R[self._r_index(k)], self._d)
        if dg is None:
            return None
        return np.full(self._d, dg) if np.ndim(dg) == 0 else dg

    def _predictive_mean(self, i, k):
        return self._nl(self._theta[i - 1], self._mu[k - 1], k)

    def _predictive_covariance(self, i, k):
        F = self._nl.jac_x(self._theta[i - 1], self._mu[k - 1], k)
        return F @ self._P[k - 1] @ F.T + self._q_at(k)

    def _predict_measurement(self, k, mu_bar):
        return self._C[k - 1] @ mu_bar

    def _compute_eta_k(self, k, P_bar):
        C = self._C[k - 1]
        rho = self._rho_at(k)
        tr_R = np.trace(self._R[self._r_index(k)]) if rho is None else rho.sum()
        return (tr_R + np.sum((C.T @ C) * P_bar)) / self._d

    def _compute_dictionary_innovation(self, k, eta_k, mu_bar, P_bar):
        return mu_bar.T @ self._V[k - 1] @ mu_bar + eta_k

    def _update_dictionary_mean(self, k, yk, Nk, mu_bar):
        w = self._V[k - 1] @ mu_bar
        self._C[k] = self._C[k - 1] + (yk - self._y_pred[k]) @ w.T / Nk

    def _update_dictionary_covariance(self, k, Nk, mu_bar, yk):
        w = self._V[k - 1] @ mu_bar
        self._V[k] = self._V[k - 1] - (w @ w.T) / Nk

    def _compute_inverse_coefficient_innovation(self, k, mu_bar, P_bar):
        C = self._C[k - 1]
        s = float(np.squeeze(mu_bar.T @ self._V[k - 1] @ mu_bar))
        rho = self._rho_at(k)
        if rho is not None and (rho + s).sum() > 0:
            w = 1.0 / (rho + s)
            U = w[:, None] * C
            K = np.linalg.inv(np.linalg.inv(P_bar) + C.T @ U)
            return InverseInnovation(w, U, K)
        Rbar = np.asarray(self._R[self._r_index(k)], dtype=float) + s * np.eye(self._d)
        return np.linalg.inv(C @ P_bar @ C.T + Rbar)

    def _update_coefficient_mean(self, k, yk, Skinv, mu_bar, P_bar):
        self._mu[k] = mu_bar + P_bar @ (self._C[k - 1].T @ (Skinv @ (yk - self._y_pred[k])))

    def _update_coefficient_covariance(self, k, Skinv, P_bar, yk):
        C = self._C[k - 1]
        self._P[k] = P_bar - P_bar @ (C.T @ (Skinv @ C)) @ P_bar

    def _grad_f(self, k, yk, eta_k, mu_prev, theta):
        """d(incremental likelihood)/d f at the pre-update state (closed form of psmf.py:57-64)."""
        f = self._nl(theta, mu_prev, k)
        C, V = self._C[k - 1], self._V[k - 1]
        u = V @ f
        N = float(np.squeeze(f.T @ u)) + float(np.squeeze(eta_k))
        e = yk - C @ f
        ee = float(np.squeeze(e.T @ e))
        return self._d * u / N - (C.T @ e) / N - ee * u / N**2

    def _store_gradient(self, i, k, yk, eta_k):
        theta = self._theta[i - 1]
        if np.asarray(theta).size == 0:
            return
        Jt = self._nl.jac_theta(theta, self._mu[k - 1], k)
        g = Jt.T @ self._grad_f(k, yk, eta_k, self._mu[k - 1], theta)
        self._gradsum = self._gradsum + g.reshape(self._gradsum.shape)

    def _prune(self, k):
        del self._C[k - 1], self._V[k - 1], self._mu[k - 1], self._P[k - 1]

    def predict(self, i, T, n_pred):
        if self.backend == "hip" and self._dev is not None and isinstance(self._y_pred, _YPred):
            return self._predict_hip(i, T, n_pred)
        self._mu_pred = {T: self._mu[T]}
        for k in range(T + 1, T + n_pred + 1):
            self._mu_pred[k] = self._nl(self._theta[i - 1], self._mu_pred[k - 1], k)
            self._y_pred[k] = self._C[T] @ self._mu_pred[k]

    # ------------------------------------------------------------------ optimiser (host, p-sized)
    def adam_init(self, gam=1e-3, b1=0.9, b2=0.999):
        self.adam_gam = gam if isinstance(gam, BaseLearningRate) else ConstantLearningRate(gam)
        self.adam_b1, self.adam_b2 = b1, b2
        shape = np.asarray(self.theta0).shape
        self.adam_m, self.adam_v = np.zeros(shape), np.zeros(shape)
        self.adam_m_hat, self.adam_v_hat = np.zeros(shape), np.zeros(shape)

    def sgd_init(self, gam=1e-3):
        self.sgd_gam = gam if isinstance(gam, BaseLearningRate) else ConstantLearningRate(gam)

    def optim_init(self, gam=1e-3):
        if self.optim == "adam":
            self.adam_init(gam=gam)
        else:
            self.sgd_init(gam=gam)

    def optim_update(self, i, project=True):
        if self.optim == "adam":
            return self.adam_update(i, project=project)
        return self.sgd_update(i, project=project)

    def adam_update(self, i, project=True):
        g = self._gradsum
        self.adam_m = self.adam_b1 * self.adam_m + (1 - self.adam_b1) * g
        self.adam_v = self.adam_b2 * self.adam_v + (1 - self.adam_b2) * g * g
        self.adam_m_hat = self.adam_m / (1 - self.adam_b1**i)
        self.adam_v_hat = self.adam_v / (1 - self.adam_b2**i)
        step = self.adam_gam.get(i) * self.adam_m_hat / (np.sqrt(self.adam_v_hat) + 1e-8)
        self._theta[i] = self._theta[i - 1] - step
        if project:
            self._theta[i] = np.maximum(self._theta[i], 0)

    def sgd_update(self, i, project=True):
        self._theta[i] = self._theta[i - 1] - self.sgd_gam.get(i) * self._gradsum
        if project:
            self._theta[i] = np.maximum(self._theta[i], 0)

    # ------------------------------------------------------------------ device back end
    def _overridden_hooks(self):
        base = rPSMF_BASE if self.robust else PSMFIter
        return [h for h in COMPUTE_HOOKS if getattr(type(self), h) is not getattr(base, h)]

    def _overrides(self, name):
        """True if the class replaces `name` of the library class it derives from (PSMFIter / PSMFRecursive / rPSMF*)."""
        mine = getattr(type(self), name, None)
        for base in type(self).__mro__:
            if base in _BASE_CLASSES and base is not object and name in vars(base):
                return mine is not vars(base)[name]
        return False

    def _check_hip_configuration(self):
        over = self._overridden_hooks()
        declared = any("hip_mode" in vars(c) for c in type(self).__mro__ if c not in _BASE_CLASSES)
        recognised = False
        if over and not declared:
            # an experiment subclass written against the reference (synthetic_psmf.py:78-100): find out WHAT its hooks compute
            # by running them on small probe problems next to the library's statement of every device mode (modes.py)
            from .modes import HookRecognitionWarning, recognise_hip_mode

            if not self.recognise:
                raise TypeError(
                    f"{type(self).__name__} overrides {over} without declaring hip_mode, and recognition by probing is switched "
                    "off (recognise=False / PSMF_RECOGNISE=0).  Declare the class attribute hip_mode, or use backend='numpy'.")
            mode = recognise_hip_mode(self)
            if mode is not None:
                import warnings

                warnings.warn(
                    f"{type(self).__name__}: the overridden hooks {over} were matched to the device mode {mode!r} by running them on "
                    "probe problems (several sizes, step indices up to 10^5); the fused device loop executes that mode, NOT the "
                    "Python hooks.  A hook whose behaviour depends on something the probes do not vary is not detected: declare "
                    "`hip_mode` on the class to state the intent, or pass recognise=False / backend='numpy'.",
                    HookRecognitionWarning, stacklevel=4)
            if mode is None:
                raise TypeError(
                    f"{type(self).__name__} overrides {over}: the device back end can only fuse recognised hook "
                    f"configurations, and on a probe problem these hooks match none of {sorted(HIP_MODES)}.  Declare the class "
                    "attribute hip_mode if they are meant to, or construct with backend='numpy' (which executes the hooks).")
            self.hip_mode = mode
            recognised = True
        self.hip_mode_recognised = recognised
        if self.hip_mode not in HIP_MODES:
            raise ValueError(f"unknown hip_mode {self.hip_mode!r}")
        extra = [] if recognised else sorted(set(over) - HIP_MODE_HOOKS[self.hip_mode])
        if extra:
            raise TypeError(
                f"{type(self).__name__} declares hip_mode={self.hip_mode!r} but also overrides {extra}, which that mode "
                "does not cover: the fused device loop would skip them.  Use backend='numpy'.")
        lost = [m for m in FUSED_METHODS if m not in COMPUTE_HOOKS and self._overrides(m)]
        if lost:
            raise TypeError(
                f"{type(self).__name__} overrides {lost}: with backend='hip' the whole time loop runs on the device and "
                "these methods are never called.  Use backend='numpy'.")
        if self._r > _capi.RMAX:
            raise ValueError(f"r = {self._r} > {_capi.RMAX}")

    def _rho_of(self, Rk):
        dg = _diag_of(Rk, self._d)
        rho = None if dg is None else _as_scalar_if_uniform(dg)
        if rho is None:
            if dg is not None and self._dense_noise() is None and np.array_equal(dg, self._row_noise()):
                return 1.0          # the scalar in front of diag(rho_rows) (psmf_set_row_noise)
            if dg is None and self._dense_noise() is not None and (Rk is self._first_R() or np.array_equal(Rk, self._first_R())):
                return 1.0          # the scalar in front of U diag(lam) U^T (psmf_set_noise_rotation)
            raise NotImplementedError("the device path needs R_k = rho_k * I, or ONE constant matrix R (a non-uniform diagonal or a "
                                      "symmetric positive semi-definite dense matrix); use backend='numpy'")
        return rho

    def _dense_noise(self):
        """(lam, U) of a NON-DIAGONAL R = U diag(lam) U^T (the reference's dense branch, psmf.py:150-152) or None.  The device
        keeps the series and C in the eigenbasis of R, where the step is the non-uniform-diagonal one (psmf_set_noise_rotation);
        the O(d^3) symmetric eigen-decomposition is done once, here, by LAPACK."""
        if not hasattr(self, "_dense_noise_cache"):
            R = self._first_R()
            out = None
            if np.ndim(R) == 2 and np.shape(R) == (self._d, self._d) and self._d > 1 and _diag_of(R, self._d) is None:
                R = np.asarray(R, dtype=float)
                if not np.allclose(R, R.T, rtol=1e-12, atol=1e-14 * np.max(np.abs(R))):
                    raise NotImplementedError("the device path needs a symmetric R (a covariance); use backend='numpy'")
                lam, U = np.linalg.eigh(0.5 * (R + R.T))
                if lam[0] < -1e-12 * max(lam[-1], 0.0) or not lam[-1] > 0.0:
                    raise NotImplementedError("the device path needs a positive semi-definite R (a covariance); use backend='numpy'")
                out = (np.maximum(lam, 0.0), np.ascontiguousarray(U))
            self._dense_noise_cache = out
        return self._dense_noise_cache

    def _first_R(self):
        R = self._R
        return R[1 if 1 in R else min(R.keys())] if isinstance(R, dict) else R

    def _row_noise(self):
        """diag(R) as a (d,) vector when R is a NON-uniform diagonal (the device then weights every row, per-step engine), else None."""
        if not hasattr(self, "_row_noise_cache"):
            dn = self._dense_noise()
            if dn is not None:          # a non-diagonal R: diagonal in its eigenbasis, where the device works
                self._row_noise_cache = dn[0]
            else:
                dg = _diag_of(self._first_R(), self._d)
                self._row_noise_cache = None if (dg is None or np.ndim(dg) == 0 or _as_scalar_if_uniform(dg) is not None) else np.asarray(dg, dtype=float)
        return self._row_noise_cache

    def _q_matrix(self, Qk):
        Q = np.asarray(Qk, dtype=float)
        return float(Q) * np.eye(self._r) if Q.ndim == 0 else Q.reshape(self._r, self._r)

    def _device_rho_q(self, T=None):
        """(rho, Q, rho_sched, q_sched) for the device: PSMFIter reads R[k], Q[k] of step k (psmf.py:115,123,141).
        Constant dictionaries give (rho, Q, None, None); R_k = rho_k I and Q_k = q_k Q_1 give per-step scalar schedules
        (index k, entry 0 unused); a Q[k] that is not a multiple of Q[1] returns q_sched = "host" (the host-stepped mode
        forms P_bar itself and takes any Q[k])."""
        R, Q = self._R, self._Q
        ks = list(range(1, (T or 0) + 1))
        if not isinstance(R, dict):
            R = {k: R for k in [0] + ks}
        if not isinstance(Q, dict):
            Q = {k: Q for k in [0] + ks}
        # which R a step reads: the full filter R[k] (psmf.py:123,141); the simplified hooks R[k - 1]
        # (synthetic_psmf.py:86-87: eta = tr(R[k-1]) / d).  The device applies rho_sched[k] at step k either way.
        off = 1 if self.hip_mode == "simplified" else 0
        k1 = (1 - off) if (1 - off) in R else min(R.keys())
        rho1 = self._rho_of(R[k1])
        Q1 = self._q_matrix(Q[1 if 1 in Q else min(Q.keys())])
        rho_s = q_s = None
        seen_R, seen_Q = {id(R[k1]): rho1}, {}
        kq1 = 1 if 1 in Q else min(Q.keys())
        for k in ks:
            Rk = R.get(k - off, R[k1])    # (a dictionary without the step's key: the constant it was built from)
            if id(Rk) not in seen_R:
                seen_R[id(Rk)] = self._rho_of(Rk)
            rk = seen_R[id(Rk)]
            if rk != rho1 and rho_s is None:
                rho_s = np.full(len(ks) + 1, rho1)
            if rho_s is not None:
                rho_s[k] = rk
            Qk = Q.get(k, Q[kq1])
            if id(Qk) not in seen_Q:
                Qm = self._q_matrix(Qk)
                if np.array_equal(Qm, Q1):
                    seen_Q[id(Qk)] = 1.0
                else:
                    nz = np.flatnonzero(Q1)
                    c = Qm.reshape(-1)[nz[0]] / Q1.reshape(-1)[nz[0]] if nz.size else np.nan
                    seen_Q[id(Qk)] = float(c) if np.isfinite(c) and np.allclose(Qm, c * Q1, rtol=1e-14, atol=0.0) else "host"
            qk = seen_Q[id(Qk)]
            if qk == "host":
                q_s = "host"
            elif not isinstance(q_s, str):
                if qk != 1.0 and q_s is None:
                    q_s = np.ones(len(ks) + 1)
                if q_s is not None:
                    q_s[k] = qk
        return rho1, Q1, rho_s, q_s

    def _host_stepped(self):
        """True when f is evaluated on the host, one device step at a time (psmf_step_host): arbitrary callables and what
        psmf_dyn.hip does not hold (a Fourier basis of more than 4 + 4 terms), or a Q[k] schedule of matrices too large to
        upload.  The recognised kinds run inside the device time loop on either engine (scaled walk / sinusoid / Fourier at
        r > 32 or with a non-uniform R: the per-step engine's serial stage)."""
        return self._nl.device_kind is None or getattr(self, "_force_host_stepped", False)

    def _device_kwargs(self):
        coef, eta_full, pbar = HIP_MODES[self.hip_mode]
        kw = dict(robust=self.robust, coef_update=coef, eta_full=eta_full, pbar_predict=pbar, **self._dev_opts)
        if self._host_stepped():
            kw.update(dyn_kind=_capi.DYN_HOST, engine="step")
        else:
            kw.update(dyn_kind=self._nl.device_kind, dyn_flags=self._nl.device_flags, dyn_terms=self._nl.device_terms)
        if self._row_noise() is not None or getattr(self, "_q_matrix_sched", False):
            kw.update(engine="step")
        if self._row_noise() is not None:
            kw.update(nonuniform_R=True)
        return kw

    # a Q[k] schedule of matrices (not multiples of Q[1]) goes to the device as [T + 1, r, r] up to this size, else host-stepped
    _Q_MATRIX_SCHEDULE_MAX_BYTES = 1 << 31

    def _q_matrices(self, T):
        Q = self._Q
        k1 = 1 if 1 in Q else min(Q.keys())
        out = np.empty((T + 1, self._r, self._r))
        out[0] = self._q_matrix(Q[k1])
        for k in range(1, T + 1):
            out[k] = self._q_matrix(Q.get(k, Q[k1]))
        return out

    def _ensure_device(self):
        if self._dev is None:
            self._dev = _capi.DeviceFilter(self._d, self._r, **self._device_kwargs())
            if self._dense_noise() is not None:
                lam, U = self._dense_noise()
                self._dev.set_noise_rotation(U, lam)
            elif self._row_noise() is not None:
                self._dev.set_row_noise(self._row_noise())
        return self._dev

    def _fetch_device(self, name, epoch):
        if epoch != self._dev_epoch:
            raise RuntimeError(f"stale device reference to {name} (the device state has moved on)")
        if name == "C":
            return self._dev.get_state(want_C=True)["C"]
        raise KeyError(name)

    def _upload_series(self, y, T):
        """y[k], k = 1..T, onto the device -- dict {k: (d, 1)} as in the reference, or any array-like indexed the same way
        (row 0 unused), exactly what the numpy back end reads.  The resident copy is reused only if the CONTENT is
        unchanged (a checksum of the T observations; object identity says nothing: ids are recycled and containers are
        refilled in place)."""
        if isinstance(y, dict):
            Y = np.concatenate([np.asarray(y[k]).reshape(1, -1) for k in range(1, T + 1)], axis=0)
        else:
            Y = np.asarray(y)[1:T + 1].reshape(T, -1)
        if Y.shape != (T, self._d):
            raise ValueError(f"series: expected {T} observations of length {self._d}, got an array of shape {Y.shape}")
        Y = np.ascontiguousarray(Y)
        key = (T, Y.dtype.str, _content_hash(Y))
        if self._series_key == key:
            return
        self._dev.upload_series(Y, t0=0, T_total=T)
        self._series_key = key

    def _push_state(self, i, T=None):
        dev = self._dev
        C0 = self._C.raw(0)
        same_C = isinstance(C0, _Lazy) and C0.owner is self and C0.epoch == self._dev_epoch
        rho, Q, rho_s, q_s = self._device_rho_q(T)
        theta = np.asarray(self._theta[i - 1], dtype=float).reshape(-1)
        dev.set_state(None if same_C else np.asarray(C0, dtype=float), self._V[0], self._P[0], Q,
                      np.asarray(self._mu[0]).reshape(-1), rho=rho, lambda0=self._device_lambda0(),
                      theta=theta if (theta.size and dev.n_theta) else None)
        sched = (None if rho_s is None else tuple(rho_s), None if (q_s is None or isinstance(q_s, str)) else tuple(q_s))
        if sched != getattr(self, "_sched_key", (None, None)):
            dev.set_schedules(rho_s, None if isinstance(q_s, str) else q_s)
            self._sched_key = sched
        if getattr(self, "_q_matrix_sched", False) and not self._host_stepped():
            Qm = self._q_matrices(T)
            key = (Qm.shape, _content_hash(Qm))
            if key != getattr(self, "_qmat_key", None):
                dev.set_q_matrix_schedule(Qm)
                self._qmat_key = key

    def _device_lambda0(self):
        return 0.0

    def _pull_state(self, T):
        s = self._dev.get_state(want_C=False)
        self._dev_epoch += 1
        ep = self._dev_epoch
        self._C = _StateDict({T: _Lazy(self, "C", ep)})
        self._V = _StateDict({T: s["V"]})
        self._P = _StateDict({T: s["P"]})
        self._mu = _MuHist(self, T, s["mu"].reshape(-1, 1))
        if s["gradsum"].size == np.asarray(self.theta0).size and s["gradsum"].size:
            self._gradsum = s["gradsum"].reshape(np.asarray(self.theta0).shape)
        self._y_pred = _YPred(self, T)
        return s

    def _step_hip(self, y, i, T):
        self.step_reset()
        if (not self._host_stepped() and not self.robust and not getattr(self, "_q_matrix_sched", False)
                and isinstance(self._device_rho_q(T)[3], str)):
            # Q[k] is not a scalar multiple of Q[1]: uploaded matrix by matrix, P_bar = F P F^T + Q_k formed in the per-step
            # engine's serial stage (psmf_set_q_matrix_schedule) -- or, beyond _Q_MATRIX_SCHEDULE_MAX_BYTES, by the host, one
            # device step at a time.  Either way another kind of handle.
            if (T + 1) * self._r * self._r * 8 <= self._Q_MATRIX_SCHEDULE_MAX_BYTES:
                self._q_matrix_sched = True
            else:
                self._force_host_stepped = True
            if self._dev is not None:
                self._dev.close()
                self._dev, self._series_key, self._sched_key = None, None, (None, None)
        self._ensure_device()
        self._upload_series(y, T)
        self._push_state(i, T)
        if self._host_stepped():
            return self._after_device_epoch(self._run_host_stepped(i, T), T)
        self._dev.zero_gradsum()
        self._dev.run(0, T)
        self._after_device_epoch(self._pull_state(T), T)
        self._verify_recognised_nonlinearity(self._theta[i - 1], T)

    def _q_for_step(self, k, Q_running):
        """Q entering P_bar of step k: PSMFIter reads Q[k] (psmf.py:115); rPSMFIter its running Q_{k-1} (rpsmf.py:123)."""
        if self.robust:
            return Q_running
        Q = self._Q[k] if isinstance(self._Q, dict) else self._Q
        return self._q_matrix(Q)

    def _run_host_stepped(self, i, T, recursive=False):
        """The time loop with f on the host: per step the host evaluates mu_bar = f(theta, mu, k), F = df/dx and
        P_bar = F P F^T + Q_k (r-sized, psmf.py:104-115), the device does everything d-sized of inner() (psmf_step_host),
        and the theta gradient is accumulated as J_theta^T g_f from the g_f the device returns (psmf.py:167-177)."""
        dev, nl, r = self._dev, self._nl, self._r
        pbar_predict = HIP_MODES[self.hip_mode][2]
        theta = self._theta[i - 1] if not recursive else self._theta[0]
        n_th = np.asarray(theta).size
        mu = np.asarray(self._mu[0], dtype=float).reshape(-1, 1)
        P = np.asarray(self._P[0], dtype=float)
        Qrun = self._q_matrix(self.Q0) if self.robust else None
        grad = np.zeros(np.asarray(self.theta0).shape)
        for k in range(1, T + 1):
            mu_bar = np.asarray(nl(theta, mu, k), dtype=float).reshape(-1)
            if pbar_predict:
                F = nl.jac_x(theta, mu, k)
                P_bar = F @ P @ F.T + self._q_for_step(k, Qrun)
            else:
                P_bar = P
            mu_new, gf, P, Qrun_dev = dev.step_host(k - 1, mu_bar, 0.5 * (P_bar + P_bar.T))
            if self.robust:
                Qrun = Qrun_dev
            if n_th:
                Jt = nl.jac_theta(theta, mu, k)
                grad = grad + (Jt.T @ gf).reshape(grad.shape)
            mu = mu_new.reshape(-1, 1)
            if recursive:
                self._gradsum = grad
                if k % self._update_every == 0:
                    self.optim_update(k)
                    grad = np.zeros_like(grad)
                else:
                    self._carry_theta(k)
                theta = self._theta[k]
        s = self._pull_state(T)
        self._gradsum = grad
        s["gradsum"] = np.zeros(0)
        s["theta"] = np.asarray(theta, dtype=float).reshape(-1)
        return s

    def _after_device_epoch(self, s, T):
        pass

    def _verify_recognised_nonlinearity(self, theta, T, ks=None):
        """A plain callable that the constructor matched to a closed-form family (modes.recognise_nonlinearity) ran inside
        the device loop as that family.  Re-check the match where it matters: at (theta, mu_{k-1}, k) for step indices spread
        over the epoch just filtered (the mean history is on the device).  A mismatch means the epoch was filtered with the
        wrong f: raise, naming the way out."""
        fn = getattr(self._nl, "recognised_from", None)
        if fn is None or T < 1:
            return
        from .modes import nonlinearity_mismatch

        if ks is None:
            ks = sorted({1, 2, T, *np.linspace(1, T, num=min(T, 24), dtype=int).tolist()})
        for k in ks:
            if nonlinearity_mismatch(fn, self._nl, theta, self._mu[k - 1], k):
                raise RuntimeError(
                    f"the nonlinearity {getattr(fn, '__name__', fn)!r} was recognised as {type(self._nl).__name__} on probe inputs "
                    f"but differs from it at step {k} of this run: the device evaluated the wrong function.  Construct with "
                    "recognise=False (or PSMF_RECOGNISE=0) so that the callable is host-stepped.")

    def _predict_hip(self, i, T, n_pred):
        if self._host_stepped():
            theta = self._theta[i - 1] if (i - 1) in self._theta else self._theta[max(self._theta)]
            mu, roll = self._mu[T], []
            for k in range(T + 1, T + n_pred + 1):      # psmf.py:183-187 on the host (r-sized), C mu_pred on the device
                mu = np.asarray(self._nl(theta, mu, k), dtype=float).reshape(-1, 1)
                roll.append(mu.reshape(-1))
            out = self._dev.project(np.array(roll)) if roll else np.zeros((0, self._d))
        else:
            out = self._dev.predict(T, n_pred)
        for q in range(n_pred):
            dict.__setitem__(self._y_pred, T + q + 1, out[q].reshape(-1, 1))

    def sq_errors(self, T):
        """sum_k ||y_hat_k - y_k||^2 over the filtered steps, reduced on the device
        (what TrackingMixin.errors_update computes from `_y_pred` on the host)."""
        return self._dev.sq_error(0, T)


def _device_optimiser(obj):
    """Keyword arguments of the in-loop optimiser the device implements -- Adam (psmf.py:224-242) or plain SGD (psmf.py:244-248),
    each with a constant or an exponentially decaying learning rate (learning_rate.py:13-27) -- or None: custom learning-rate
    schedules keep theta and its optimiser on the host (host-stepped device loop)."""
    if obj.optim == "adam":
        gam, kind = getattr(obj, "adam_gam", ConstantLearningRate(1e-3)), 1
    elif obj.optim == "sgd":
        gam, kind = getattr(obj, "sgd_gam", ConstantLearningRate(1e-3)), 2
    else:
        return None
    if isinstance(gam, ConstantLearningRate):
        return dict(recursive=kind, adam_lr=gam.lr)
    if hasattr(gam, "lr_start") and hasattr(gam, "lr_end") and hasattr(gam, "steps"):
        return dict(recursive=kind, adam_lr=gam.lr_start, adam_lr_end=gam.lr_end, adam_lr_steps=gam.steps)
    return None


def _recursive_kwargs(obj, kw):
    """Adds the in-loop optimiser's configuration (psmf.py:224-248,299-304) to the device options: recursive = 1 Adam, 2 SGD."""
    if kw.get("dyn_kind") == _capi.DYN_HOST:
        return kw               # host-stepped: theta and its optimiser (any schedule) stay on the host
    kw.update(update_every=getattr(obj, "_update_every", 1),
              adam_b1=getattr(obj, "adam_b1", 0.9), adam_b2=getattr(obj, "adam_b2", 0.999), **_device_optimiser(obj))
    return kw


def _recursive_host_stepped(obj, base):
    """Recursive classes: also host-stepped when the in-loop optimiser is not one the device implements."""
    return base or (np.asarray(obj.theta0).size > 0 and _device_optimiser(obj) is None)


class PSMFIterMissing(PSMFIter):
    def __init__(*args, **kwargs):
        # unfinished in the reference as well (pypsmf/psmf/psmf.py:251-254); the working masked
        # filter is rpsmf_amd.impute (ExperimentImpute semantics)
        raise NotImplementedError


class PSMFRecursive(PSMFIter):
    """Online variant: theta takes an optimiser step inside the time loop every `update_every`
    observations (pypsmf/psmf/psmf.py:275-331)."""

    def run(self, y, T, n_pred, update_every=1):
        self._update_every = update_every
        self.optim_init()
        self.step(y, T)
        self.predict(T, n_pred)

    def step(self, y, T):
        if self.backend == "hip":
            return self._step_hip_recursive(y, T)
        self.step_reset()
        for k in range(1, T + 1):
            self.inner(k, np.asarray(y[k]).reshape(-1, 1))

    def inner(self, k, yk):
        mu_bar = self._predictive_mean(k, k)
        P_bar = self._predictive_covariance(k, k)
        self._y_pred[k] = self._predict_measurement(k, mu_bar)
        eta_k = self._compute_eta_k(k, P_bar)
        Nk = self._compute_dictionary_innovation(k, eta_k, mu_bar, P_bar)
        self._update_dictionary_mean(k, yk, Nk, mu_bar)
        self._update_dictionary_covariance(k, Nk, mu_bar, yk)
        Skinv = self._compute_inverse_coefficient_innovation(k, mu_bar, P_bar)
        self._update_coefficient_mean(k, yk, Skinv, mu_bar, P_bar)
        self._update_coefficient_covariance(k, Skinv, P_bar, yk)
        self._store_gradient(k, k, yk, eta_k)
        if k % self._update_every == 0:
            self.optim_update(k)
            self._reset_gradient()
        else:
            self._carry_theta(k)

    def _reset_gradient(self):
        self._gradsum = np.zeros(np.asarray(self.theta0).shape)

    def _carry_theta(self, i):
        self._theta[i] = self._theta[i - 1]

    def _set_gradient(self, k, yk, eta_k):
        self._reset_gradient()
        self._store_gradient(k, k, yk, eta_k)

    def predict(self, T, n_pred):
        if self.backend == "hip" and self._dev is not None and isinstance(self._y_pred, _YPred):
            return self._predict_hip(T + 1, T, n_pred)        # rolls forward with theta_T (psmf.py:324-331)
        last_theta = self._theta[T]
        self._mu_pred = {T: self._mu[T]}
        for k in range(T + 1, T + n_pred + 1):
            self._mu_pred[k] = self._nl(last_theta, self._mu_pred[k - 1], k)
            self._y_pred[k] = self._C[T] @ self._mu_pred[k]

    # device: Adam and plain SGD run inside the time loop of either engine; custom learning-rate schedules: host-stepped
    def _host_stepped(self):
        return _recursive_host_stepped(self, super()._host_stepped())

    def _device_kwargs(self):
        return _recursive_kwargs(self, super()._device_kwargs())

    def _step_hip_recursive(self, y, T):
        self.step_reset()
        self._ensure_device()
        self._upload_series(y, T)
        self._theta = {0: self._theta[0]}
        self._push_state(1, T)
        if self._host_stepped():
            return self._after_device_epoch(self._run_host_stepped(1, T, recursive=True), T)
        self._dev.zero_gradsum()
        if self.optim == "adam":
            self._dev.set_adam(self.adam_m.reshape(-1), self.adam_v.reshape(-1))
        self._dev.run(0, T)
        s = self._pull_state(T)
        self._theta[T] = s["theta"].reshape(np.asarray(self.theta0).shape)
        self._after_device_epoch(s, T)
        self._verify_recognised_nonlinearity(self._theta[0], T, ks=[1])      # theta moves inside the loop: only step 1 has a known theta


rPSMF_BASE = PSMFIter          # rebound by rpsmf.py once rPSMFIter exists
_BASE_CLASSES = [PSMFIter, PSMFRecursive, PSMFIterMissing, object]
