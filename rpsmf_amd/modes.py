"""Recognising what a user's subclass computes, so that it can run on the device unchanged.

The reference's plug-in API is the template method: an experiment subclasses `PSMFIter` / `rPSMFIter` and replaces hooks
(ExperimentSynthetic/synthetic_psmf.py:78-100, synthetic_rpsmf.py:82-118) and passes a plain function as the
nonlinearity (synthetic_psmf.py:105-106).  The device runs the whole time loop as fused kernels, so it can only execute hook
configurations it has kernels for (`HIP_MODES`).  Instead of asking the user to annotate the class, the constructor PROBES it:

* `recognise_hip_mode(obj)`: the subclass's hooks are executed, on the host, on small random problems (probe objects of
  the same class carrying the user object's own attributes; two row counts; ten step indices from 1 to 10^5 with R_k and
  Q_k different at every step), and so are the library's own classes for every entry of `HIP_MODES`; the mode whose C, V,
  mu, P, y_hat, gradient (and rPSMF's R, Q, lambda) agree to 1e-12 on ALL of them is the one the device runs.  No match:
  the constructor raises, as before (or `backend="numpy"` executes the hooks themselves).  A match is announced with a
  `HookRecognitionWarning`: the device then runs the mode's kernels, not the Python hooks, and a finite set of probes
  cannot exclude a hook whose behaviour changes outside it.
* `recognise_nonlinearity(fn, n_params, rank)`: a plain callable is compared with the closed-form families the device
  evaluates (nonlinearities.py) on 30 random (theta, x, t) -- theta inside and outside [0, 1), |x| up to a few hundred,
  t up to 10^6; a match makes it run inside the device loop with analytic derivatives, otherwise it stays host-stepped
  (psmf_step_host), which is correct for any callable.  After every device epoch the filter classes compare the
  callable with the family again on states the run actually visited and raise on a mismatch (psmf.py).

Both are switched off by `recognise=False` (constructor keyword) or `PSMF_RECOGNISE=0` (environment): undeclared overrides
then raise and plain functions are host-stepped.  Declaring `hip_mode` on the class / passing a `BaseNonLinearity` object
states the intent and involves no probing.
Nothing here touches the GPU or the oracle; the probes are r-sized host arithmetic, run once per construction.
"""

import copy

import numpy as np

from . import nonlinearities as NL
from .psmf import HIP_MODES, PSMFIter
from .rpsmf import rPSMFIter

__all__ = ["SimplifiedPSMF", "SimplifiedRPSMF", "recognise_hip_mode", "recognise_nonlinearity", "nonlinearity_mismatch", "mode_class",
           "HookRecognitionWarning"]

PROBE_TOL = 1e-12


# ---------------------------------------------------------------------------------------------------------------------
# the library's own statement of the "simplified" mode (SURVEY App. A mode table), in O(d r) form
# ---------------------------------------------------------------------------------------------------------------------
class _SimplifiedHooks:
    """P_bar = P_{k-1}; eta = tr(R_{k-1}) / d; no coefficient update (mu_k = mu_bar, P_k = P_bar)."""

    hip_mode = "simplified"

    def _noise_diag_prev(self, k):
        from .psmf import _diag_of

        dg = _diag_of(self._R[k - 1], self._d)
        if dg is None:
            raise NotImplementedError("simplified mode: diagonal R only")
        return np.full(self._d, dg) if np.ndim(dg) == 0 else dg

    def _predictive_covariance(self, i, k):
        return self._P[k - 1]

    def _compute_eta_k(self, k, P_bar):
        return float(np.sum(self._noise_diag_prev(k))) / self._d

    def _compute_inverse_coefficient_innovation(self, k, mu_bar, P_bar):
        return None

    def _update_coefficient_mean(self, k, yk, Skinv, mu_bar, P_bar):
        self._mu[k] = mu_bar

    def _update_coefficient_covariance(self, k, Skinv, P_bar, yk):
        self._P[k] = P_bar


class SimplifiedPSMF(_SimplifiedHooks, PSMFIter):
    """PSMF with the ExperimentSynthetic simplifications (synthetic_psmf.py:82-98) as a ready-made class."""


class SimplifiedRPSMF(_SimplifiedHooks, rPSMFIter):
    """rPSMF with the ExperimentSynthetic simplifications (synthetic_rpsmf.py:86-114): omega from (R + s I)^-1 alone, R scaled
    by omega, Q and P carried."""

    def _update_coefficient_covariance(self, k, Skinv, P_bar, yk):
        lam = self._lambda[k - 1]
        mu_bar = self._mu[k]                                   # = mu_bar of this step (no coefficient update)
        s = float(np.squeeze(mu_bar.T @ self._V[k - 1] @ mu_bar))
        e = np.asarray(yk - self._y_pred[k]).reshape(-1)
        quad = float(np.sum(e * e / (self._noise_diag_prev(k) + s)))
        omega = (lam + quad) / (lam + self._d)
        self._P[k] = P_bar
        self._Q[k] = self._Q[k - 1]
        self._R[k] = omega * self._R[k - 1]
        if not self.fixed_lambda:
            self._lambda[k] = lam + self._d


def mode_class(mode, robust):
    """The library class that states `mode` on the host."""
    if mode == "full":
        return rPSMFIter if robust else PSMFIter
    if mode == "simplified":
        return SimplifiedRPSMF if robust else SimplifiedPSMF
    raise KeyError(mode)


# ---------------------------------------------------------------------------------------------------------------------
# hook probing
# ---------------------------------------------------------------------------------------------------------------------
class HookRecognitionWarning(UserWarning):
    """Emitted when an undeclared hook override has been matched to a device mode by probing (see recognise_hip_mode)."""


# Step indices the hooks are run at: the first steps, then pairs spread up to 10^5 (a hook that switches behaviour at
# some k shows unless the switch lies beyond); sizes: two different d (a hook that depends on d shows).  This is a
# finite experiment, not a proof -- hence the warning the constructor emits and the `recognise=False` opt-out.
PROBE_STEPS = (1, 2, 3, 17, 64, 65, 1000, 4097, 25000, 100000)
PROBE_ROWS = (7, 12)


def _rho_at(k):
    return 0.8 + 0.5 * ((k * 0.6180339887498949) % 1.0)       # different at every step, so an index off by one shows


def _q_at(k):
    return 1.0 + 0.25 * (k % 7)


def _probe_problem(r, robust, fixed_lambda, d):
    rng = np.random.default_rng(0x5EED0 + 131 * d + r)
    A, B = rng.standard_normal((r, r)), rng.standard_normal((r, r))
    return dict(
        d=d, steps=PROBE_STEPS,
        theta0=0.05 + 0.1 * rng.random((r, 1)),
        C0=0.3 * rng.standard_normal((d, r)),
        V0=0.2 * np.eye(r) + 0.02 * (A @ A.T) / r,
        mu0=0.3 * rng.standard_normal((r, 1)),
        P0=0.5 * np.eye(r) + 0.05 * (B @ B.T) / r,
        Q=0.07 * np.eye(r),
        lambda0=2.3,
        Y=rng.standard_normal((len(PROBE_STEPS), d)),
        fixed_lambda=fixed_lambda,
    )


def _make_probe(cls, robust, pb, alpha, beta, like=None):
    """An object of class `cls` initialised by the LIBRARY initialiser (numpy back end) on the probe problem.  `like`: the
    user's object -- attributes its own __init__ set (which object.__new__ bypasses) are carried over."""
    d, r = pb["C0"].shape
    nl = NL.CosPhase(r)
    obj = object.__new__(cls)
    keys = sorted({k for s in pb["steps"] for k in (s - 1, s)} | {0})
    if robust:
        rPSMFIter.__init__(obj, pb["theta0"], pb["C0"], pb["V0"], pb["mu0"], pb["P0"], pb["Q"], _rho_at(0) * np.eye(d),
                           pb["lambda0"], nl, fixed_lambda=pb["fixed_lambda"], backend="numpy")
        obj._alpha, obj._beta = alpha, beta
    else:
        Qs = {k: _q_at(k) * pb["Q"] for k in keys}
        Rs = {k: _rho_at(k) * np.eye(d) for k in keys}
        PSMFIter.__init__(obj, pb["theta0"], pb["C0"], pb["V0"], pb["mu0"], pb["P0"], Qs, Rs, nl, backend="numpy")
    if like is not None:
        # COPIES of the user's attributes: hooks that keep counters, caches or dicts of their own would otherwise leave the
        # probe's 2 x 10 steps in the live object before its real run.  (Attributes must exist when PSMFIter.__init__ runs:
        # a subclass that sets them after super().__init__() is probed without them.)
        for name, val in vars(like).items():
            if name not in vars(obj):
                try:
                    val = copy.deepcopy(val)
                except Exception:
                    try:
                        val = copy.copy(val)
                    except Exception:
                        pass              # not copyable (an open file, a lock, ...): shared, as before
                setattr(obj, name, val)
    return obj


def _run_probe(obj, pb):
    """The hook sequence of `inner` (psmf.py:90-102) for epoch 1 at the step indices pb["steps"] (not contiguous: the state a
    step leaves is what the next probed step finds at its k - 1), without pruning; returns everything a step produces."""
    PSMFIter.step_reset(obj) if not obj.robust else rPSMFIter.step_reset(obj)
    out = []
    prev = 0
    for n, k in enumerate(pb["steps"]):
        if k - 1 != prev:                 # carry the state to where step k reads it
            for name in ("_C", "_V", "_mu", "_P"):
                D = getattr(obj, name)
                D[k - 1] = D[prev]
            if obj.robust:
                obj._R[k - 1], obj._Q[k - 1] = obj._R[prev], obj._Q[prev]
                if not obj.fixed_lambda:
                    obj._lambda[k - 1] = obj._lambda[prev]
        yk = pb["Y"][n].reshape(-1, 1)
        mu_bar = obj._predictive_mean(1, k)
        P_bar = obj._predictive_covariance(1, k)
        obj._y_pred[k] = obj._predict_measurement(k, mu_bar)
        eta = obj._compute_eta_k(k, P_bar)
        Nk = obj._compute_dictionary_innovation(k, eta, mu_bar, P_bar)
        obj._update_dictionary_mean(k, yk, Nk, mu_bar)
        obj._update_dictionary_covariance(k, Nk, mu_bar, yk)
        Sk = obj._compute_inverse_coefficient_innovation(k, mu_bar, P_bar)
        obj._update_coefficient_mean(k, yk, Sk, mu_bar, P_bar)
        obj._update_coefficient_covariance(k, Sk, P_bar, yk)
        obj._store_gradient(1, k, yk, eta)
        vals = [obj._y_pred[k], obj._C[k], obj._V[k], obj._mu[k], obj._P[k], obj._gradsum, np.asarray(eta), np.asarray(Nk)]
        if obj.robust:
            vals += [np.asarray(obj._R[k]), np.asarray(obj._Q[k]), np.asarray(obj._lambda[k])]
        out.append([np.asarray(v, dtype=float) for v in vals])
        prev = k
    return out


def _agree(a, b, tol):
    for sa, sb in zip(a, b):
        for x, y in zip(sa, sb):
            if x.size != y.size:
                return False
            x, y = x.reshape(-1), y.reshape(-1)
            if not (np.all(np.isfinite(x)) and np.all(np.isfinite(y))):
                return False
            if np.max(np.abs(x - y), initial=0.0) > tol * max(1.0, float(np.max(np.abs(y), initial=0.0))):
                return False
    return True


def recognise_hip_mode(obj, tol=PROBE_TOL):
    """Name of the `HIP_MODES` entry that the hooks of type(obj) compute on EVERY probe problem, or None.  See the module
    docstring; the probes vary the step index (PROBE_STEPS), the number of rows (PROBE_ROWS), R_k and Q_k per step."""
    cls, robust = type(obj), bool(obj.robust)
    alpha, beta = getattr(obj, "_alpha", 1.0), getattr(obj, "_beta", 1.0)
    found = None
    for d in PROBE_ROWS:
        pb = _probe_problem(obj._r, robust, bool(getattr(obj, "fixed_lambda", False)), d)
        try:
            mine = _run_probe(_make_probe(cls, robust, pb, alpha, beta, like=obj), pb)
        except Exception:
            return None                       # hooks that cannot run on the probe cannot be recognised
        match = None
        for mode in HIP_MODES:
            ref = _run_probe(_make_probe(mode_class(mode, robust), robust, pb, alpha, beta), pb)
            if _agree(mine, ref, tol):
                match = mode
                break
        if match is None or (found is not None and match != found):
            return None
        found = match
    return found


# ---------------------------------------------------------------------------------------------------------------------
# nonlinearity probing
# ---------------------------------------------------------------------------------------------------------------------
def _candidates(n_params, r):
    out = []
    if n_params == 0:
        out.append(NL.RandomWalk())
    if n_params == r:
        out += [NL.CosPhase(r), NL.Sinusoid(r, scaled=False, phased=False)]
    if n_params == 2 * r:
        out.append(NL.Sinusoid(r, scaled=False, phased=True))
    if n_params == r * r:
        out.append(NL.ScaledWalk(r, bias=False))
    if n_params == r * r + r:
        out += [NL.ScaledWalk(r, bias=True), NL.Sinusoid(r, scaled=True, phased=False)]
    if n_params == r * r + 2 * r:
        out.append(NL.Sinusoid(r, scaled=True, phased=True))
    for N in range(1, 5):
        if n_params == N * (2 * r * r + 4 * r):
            out.append(NL.FourierBasis(r, N=N))
    return out


def _nl_probes(n_params, rank):
    """(theta, x, t) triples: theta in and outside [0, 1), |x| up to ~300, t from 1 to beyond any series length the
    experiments use -- so that a clip, a switch at some t, a restriction of theta shows as a mismatch."""
    rng = np.random.default_rng(0xF00D + rank)
    ts = (1, 2, 7, 113, 1000, 4999, 10001, 65537, 250003, 1000003)
    out = []
    for n, t in enumerate(ts):
        for th_lo, th_hi, x_scale in ((0.0, 1.0, 1.0), (-2.0, 3.0, 10.0), (0.0, 8.0, 100.0)):
            th = th_lo + (th_hi - th_lo) * rng.random((n_params, 1))
            out.append((th, x_scale * rng.standard_normal((rank, 1)), t))
    return out


def nonlinearity_mismatch(fn, known, theta, x, t, tol=1e-13):
    """True if the plain callable `fn` and the library family `known` differ at (theta, x, t) by more than rounding (the
    tolerance scales with the size of the phase 2 pi theta t + x, whose rounding is what two spellings of the same formula
    differ by)."""
    theta, x = np.asarray(theta, dtype=float).reshape(-1, 1), np.asarray(x, dtype=float).reshape(-1, 1)
    v = np.asarray(fn(theta.copy(), x.copy(), t), dtype=float).reshape(-1)
    w = np.asarray(known(theta, x, t), dtype=float).reshape(-1)
    if v.size != w.size or not np.all(np.isfinite(v)):
        return True
    scale = 1.0 + float(np.max(np.abs(x), initial=0.0)) + 2.0 * np.pi * abs(t) * float(np.max(np.abs(theta), initial=0.0))
    return bool(np.max(np.abs(w - v)) > tol * scale * max(1.0, float(np.max(np.abs(w)))))


def recognise_nonlinearity(fn, n_params, rank, tol=1e-13):
    """A library nonlinearity (device-evaluated, analytic derivatives) that equals the plain callable `fn` on the probe set
    (_nl_probes), or None.  The filter classes re-check the match on the states a run actually visits
    (PSMFIter._verify_recognised_nonlinearity) and raise if the two functions differ there."""
    if rank is None or rank < 1:
        return None
    cands = _candidates(int(n_params), int(rank))
    if not cands:
        return None
    probes = _nl_probes(int(n_params), int(rank))
    for c in cands:
        try:
            if not any(nonlinearity_mismatch(fn, c, th, x, t, tol) for th, x, t in probes):
                return c
        except Exception:
            return None
    return None
