"""CPU oracle for the per-timestep PSMF / rPSMF filter  --  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product (``rpsmf_amd``) never does;
its device path fails loudly when the HIP library is missing.

Parity status: PINNED.  This restatement is checked
  * against the reference classes themselves (imported from /root/reference in
    the build container by ``tests/golden/make_golden.py``; the outputs are the
    committed ``tests/golden/*.npz`` fixtures), and
  * against the reference's own known-answer data for the masked path
    (``ExperimentImpute/output/*_PSMF.json`` / ``*_rPSMF.json`` results, see
    ``oracle/impute_oracle.py`` and ``tests/test_oracle_kat.py``).

Two restatements of one step are provided:

``lowrank_step``  O(d r^2): the algebra the HIP kernels implement.  Every d x d
    object of the reference is removed (diagonal R, Woodbury in r x r form).
``literal_step``  O(d^2 r): forms the same d x d matrices the reference forms
    (dense R, dense inverse innovation).  Only usable for d <~ 2000; it is the
    "reference algorithm" CPU timing and a cross-check of ``lowrank_step``.

Reference equations followed (paths relative to /root/reference):
  predictive mean / covariance      pypsmf/psmf/psmf.py:104-115, rpsmf.py:116-123
  measurement prediction            pypsmf/psmf/psmf.py:117-119
  eta_k                             pypsmf/psmf/psmf.py:121-125, rpsmf.py:125-131
  N_k                               pypsmf/psmf/psmf.py:127-128
  dictionary mean / covariance      pypsmf/psmf/psmf.py:130-138, rpsmf.py:133-138
  inverse innovation (Woodbury)     pypsmf/psmf/psmf.py:140-153
  coefficient mean / covariance     pypsmf/psmf/psmf.py:155-165, rpsmf.py:155-171
  theta gradient                    pypsmf/psmf/psmf.py:48-66,167-177, rpsmf.py:53-73,173-184
  predict roll-out                  pypsmf/psmf/psmf.py:182-188
  Adam / SGD                        pypsmf/psmf/psmf.py:224-248
  "simplified" experiment modes     ExperimentSynthetic/synthetic_psmf.py:78-100,
                                    ExperimentSynthetic/synthetic_rpsmf.py:82-118
"""

from __future__ import annotations

import dataclasses
from typing import Callable, Optional

import numpy as np

__all__ = [
    "Mode",
    "State",
    "Dynamics",
    "RandomWalkDyn",
    "CosPhaseDyn",
    "CallableDyn",
    "lowrank_step",
    "literal_step",
    "run_epoch",
    "predict_rollout",
    "adam_update",
    "sgd_update",
    "synthetic_series",
]


# --------------------------------------------------------------------------
# configuration / state
# --------------------------------------------------------------------------
@dataclasses.dataclass
class Mode:
    """Which hook configuration of the reference is being restated.

    robust       False: PSMFIter (psmf.py:90-102); True: rPSMFIter (rpsmf.py:116-184)
    coef_update  True: full Kalman update of (mu, P); False: mu_k = mu_bar,
                 P_k = P_bar (synthetic_psmf.py:94-98, synthetic_rpsmf.py:100-114)
    eta_full     True: eta = tr(R + C Pbar C^T)/d; False: eta = tr(R)/d
                 (synthetic_psmf.py:86-87)
    pbar_predict True: Pbar = F P F^T + Q; False: Pbar = P_{k-1}
                 (synthetic_psmf.py:83-84)
    alpha, beta  rPSMF scaling factors (rpsmf.py:45-51); 1.0 unless use_scaling
    fixed_lambda rpsmf.py:36-40,170-171
    """

    robust: bool = False
    coef_update: bool = True
    eta_full: bool = True
    pbar_predict: bool = True
    alpha: float = 1.0
    beta: float = 1.0
    fixed_lambda: bool = False


@dataclasses.dataclass
class State:
    """Filter state after step k (all float64).

    rho is the diagonal of R: a scalar (uniform) or a (d,) vector.
    For rPSMF, Q / rho / lam are the *scaled* running values Q_k, R_k, lambda_k.
    """

    C: np.ndarray  # (d, r)
    V: np.ndarray  # (r, r)
    mu: np.ndarray  # (r,)
    P: np.ndarray  # (r, r)
    Q: np.ndarray  # (r, r)
    rho: object  # float or (d,)
    lam: float = 0.0
    theta: Optional[np.ndarray] = None  # (p,)
    gradsum: Optional[np.ndarray] = None  # (p,)

    def copy(self) -> "State":
        return State(
            C=self.C.copy(),
            V=self.V.copy(),
            mu=self.mu.copy(),
            P=self.P.copy(),
            Q=self.Q.copy(),
            rho=(self.rho.copy() if isinstance(self.rho, np.ndarray) else float(self.rho)),
            lam=float(self.lam),
            theta=None if self.theta is None else self.theta.copy(),
            gradsum=None if self.gradsum is None else self.gradsum.copy(),
        )


@dataclasses.dataclass
class StepInfo:
    """Per-step scalars / vectors that the golden fixtures pin."""

    y_pred: np.ndarray
    eta: float
    N: float
    s: float
    phi: float = 1.0
    omega: float = 1.0
    h: Optional[np.ndarray] = None
    ee: float = 0.0


# --------------------------------------------------------------------------
# dynamics f(theta, x, t) with derivatives
# --------------------------------------------------------------------------
class Dynamics:
    """State transition f(theta, x, t) plus dF/dx (r x r) and dF/dtheta (r x p).

    t is the 1-based integer step, as in the reference (psmf.py:104-105).
    """

    n_theta = 0

    def f(self, theta, x, t):
        raise NotImplementedError

    def jac_x(self, theta, x, t):
        raise NotImplementedError

    def jac_theta(self, theta, x, t):
        raise NotImplementedError


class RandomWalkDyn(Dynamics):
    """f(x) = x   (pypsmf/psmf/nonlinearities.py:42-56)."""

    n_theta = 0

    def f(self, theta, x, t):
        return x

    def jac_x(self, theta, x, t):
        return np.eye(x.shape[0])

    def jac_theta(self, theta, x, t):
        return np.zeros((x.shape[0], 0))


class CosPhaseDyn(Dynamics):
    """f = cos(2 pi theta t + x), elementwise, theta in R^r.

    ExperimentSynthetic/synthetic_psmf.py:105-106, data.py:16.
    """

    def __init__(self, r):
        self.n_theta = r

    def f(self, theta, x, t):
        return np.cos(2.0 * np.pi * theta * t + x)

    def jac_x(self, theta, x, t):
        return np.diag(-np.sin(2.0 * np.pi * theta * t + x))

    def jac_theta(self, theta, x, t):
        return np.diag(-np.sin(2.0 * np.pi * theta * t + x) * (2.0 * np.pi * t))


class CallableDyn(Dynamics):
    """Wraps a reference-style callable ``nonlinearity(theta(p,1), x(r,1), t) -> (r,1)``.

    Derivatives by complex step (exact to round-off for analytic numpy code such as
    pypsmf/psmf/nonlinearities.py:59-150); replaces autograd.jacobian of psmf.py:44.
    """

    def __init__(self, fn: Callable, n_theta: int, h: float = 1e-30):
        self.fn = fn
        self.n_theta = n_theta
        self.h = h

    def _call(self, theta, x, t):
        th = np.asarray(theta).reshape(-1, 1)
        return np.asarray(self.fn(th, np.asarray(x).reshape(-1, 1), t)).reshape(-1)

    def f(self, theta, x, t):
        return self._call(theta, x, t).real.astype(float)

    def jac_x(self, theta, x, t):
        r = x.shape[0]
        J = np.empty((r, r))
        for j in range(r):
            xp = x.astype(complex)
            xp[j] += 1j * self.h
            J[:, j] = self._call(np.asarray(theta, dtype=complex), xp, t).imag / self.h
        return J

    def jac_theta(self, theta, x, t):
        r = x.shape[0]
        p = self.n_theta
        J = np.empty((r, p))
        for j in range(p):
            tp = np.asarray(theta, dtype=complex).copy()
            tp[j] += 1j * self.h
            J[:, j] = self._call(tp, x.astype(complex), t).imag / self.h
        return J


# --------------------------------------------------------------------------
# one step, O(d r^2)
# --------------------------------------------------------------------------
def _rho_vec(rho, d):
    if np.isscalar(rho) or np.ndim(rho) == 0:
        return np.full(d, float(rho))
    return np.asarray(rho, dtype=float)


def lowrank_step(st: State, y, k, mode: Mode, dyn: Dynamics, Qk=None, rhok=None, mask=None,
                 want_grad=True):
    """Advance ``st`` (state after step k-1) by one observation y (d,).  Returns (State, StepInfo).

    Qk / rhok: the Q and diag(R) to use at this step.  PSMFIter reads Q[k], R[k]
    (psmf.py:115,123,141); rPSMFIter reads its own running Q_{k-1}, R_{k-1}
    (rpsmf.py:123,128,141) -> pass None to use st.Q / st.rho.
    mask: optional 0/1 observation mask (d,), ExperimentImpute semantics.
    """
    C, V, P, mu = st.C, st.V, st.P, st.mu
    d, r = C.shape
    Q = st.Q if Qk is None else Qk
    rho = _rho_vec(st.rho if rhok is None else rhok, d)
    theta = st.theta if st.theta is not None else np.zeros(0)
    m = np.ones(d) if mask is None else np.asarray(mask, dtype=float)

    mu_bar = dyn.f(theta, mu, k)
    if mode.pbar_predict:
        F = dyn.jac_x(theta, mu, k)
        P_bar = F @ P @ F.T + Q
    else:
        P_bar = P

    y_pred = C @ mu_bar  # stored unmasked (psmf.py:119; ExperimentImpute/PSMF.py:68)
    e = m * (y - y_pred)
    w = V @ mu_bar
    s = float(mu_bar @ w)

    if mode.eta_full:
        G = (C * m[:, None]).T @ C
        eta = (float(m @ rho) + float(np.sum(G * P_bar))) / d  # divide by d, not #observed
    else:
        eta = float(np.sum(rho)) / d
    N = s + eta

    kap = m / (rho + s)
    h = C.T @ e  # unweighted, for the theta gradient
    ee = float(e @ e)
    q = float(kap @ (e * e))

    if mode.coef_update:
        rho_in = st.rho if rhok is None else rhok
        if mask is None and np.ndim(rho_in) == 0 and mode.eta_full:
            # R = rho I and nothing masked: the weights are one number, C^T diag(kap) C = kap C^T C (the Gram formed above)
            G_R = kap[0] * G
            b = kap[0] * h
        else:
            G_R = (C * kap[:, None]).T @ C
            b = C.T @ (kap * e)
        P_plus = np.linalg.solve(np.eye(r) + P_bar @ G_R, P_bar)
        P_plus = 0.5 * (P_plus + P_plus.T)
        mu_new = mu_bar + P_plus @ b
        quad = q - float(b @ P_plus @ b)  # e^T S^-1 e
    else:
        P_plus = P_bar
        mu_new = mu_bar
        quad = q  # synthetic_rpsmf.py:91-107: S^-1 = inv(R + s I)

    C_new = C + np.outer(e, w) / N
    V_new = V - np.outer(w, w) / N

    phi = omega = 1.0
    Q_new, rho_new, lam_new = st.Q, st.rho, st.lam
    if mode.robust:
        lam = st.lam
        phi = (lam + ee / N) / (lam + d)
        omega = (lam + quad) / (lam + d)
        V_new = mode.alpha * phi * V_new
        if mode.coef_update:
            P_plus = mode.beta * omega * P_plus
            Q_new = omega * st.Q
        # simplified mode keeps Q and P (synthetic_rpsmf.py:108-111)
        rho_new = omega * (st.rho if rhok is None else rhok)
        lam_new = lam if mode.fixed_lambda else lam + d

    gradsum = st.gradsum
    if want_grad and dyn.n_theta > 0:
        Jt = dyn.jac_theta(theta, mu, k)
        if mode.robust:
            lam = st.lam
            D = lam * N
            gf = d * w / N + 0.5 * (d + lam) * (-2.0 * h / D - 2.0 * lam * ee * w / D**2) / (1.0 + ee / D)
        else:
            gf = d * w / N - h / N - ee * w / N**2
        g = Jt.T @ gf
        gradsum = g if gradsum is None else gradsum + g

    new = State(C=C_new, V=V_new, mu=mu_new, P=P_plus, Q=Q_new, rho=rho_new, lam=lam_new,
                theta=st.theta, gradsum=gradsum)
    info = StepInfo(y_pred=y_pred, eta=eta, N=N, s=s, phi=phi, omega=omega, h=h, ee=ee)
    return new, info


# --------------------------------------------------------------------------
# one step, reference-shaped dense algebra (O(d^2 r))
# --------------------------------------------------------------------------
def literal_step(st: State, y, k, mode: Mode, dyn: Dynamics, Qk=None, rhok=None):
    """Same step with the d x d matrices formed explicitly, unmasked only.  st.rho (or rhok) may be a d x d matrix: a
    non-diagonal R, which only this function takes (dense inverse of the innovation covariance, psmf.py:150-152).

    This is what the reference costs: dense R, kron(s, I_d), d x d inverse-innovation
    matrix, d x d trace argument (psmf.py:121-125,140-165).  Used for the
    "reference algorithm" CPU timing and as an independent check of lowrank_step.
    """
    C, V, P, mu = st.C, st.V, st.P, st.mu
    d, r = C.shape
    Q = st.Q if Qk is None else Qk
    rho_in = st.rho if rhok is None else rhok
    dense_R = np.ndim(rho_in) == 2        # a d x d matrix R: the reference's non-diagonal branch (psmf.py:150-152, rpsmf.py:150-152)
    Rm = np.asarray(rho_in, dtype=float) if dense_R else np.diag(_rho_vec(rho_in, d))
    theta = st.theta if st.theta is not None else np.zeros(0)
    col = lambda v: v.reshape(-1, 1)

    xb = col(dyn.f(theta, mu, k))
    if mode.pbar_predict:
        F = dyn.jac_x(theta, mu, k)
        Pb = F @ P @ F.T + Q
    else:
        Pb = P
    yk = col(y)
    yp = C @ xb
    if mode.eta_full:
        eta = np.trace(Rm + C @ Pb @ C.T) / d
    else:
        eta = np.trace(Rm) / d
    sv = (xb.T @ V @ xb).item()
    N = sv + eta
    resid = yk - yp
    C_new = C + (resid @ xb.T @ V.T) / N
    V_new = V - (V @ xb @ xb.T @ V) / N

    Rbar = Rm + sv * np.eye(d)
    if mode.coef_update:
        if dense_R:
            Sinv = np.linalg.inv(C @ Pb @ C.T + Rbar)          # psmf.py:150-152: no Woodbury form for a non-diagonal R
        else:
            Ri = np.diag(1.0 / np.diag(Rbar))
            RiC = Ri @ C
            inner = np.linalg.inv(Pb) + C.T @ RiC
            Sinv = Ri - RiC @ np.linalg.inv(inner) @ RiC.T
        mu_new = xb + Pb @ C.T @ Sinv @ resid
        P_new = Pb - Pb @ C.T @ Sinv @ C @ Pb
    else:
        Sinv = np.linalg.inv(Rbar)
        mu_new = xb
        P_new = Pb

    phi = omega = 1.0
    Q_new, rho_new, lam_new = st.Q, st.rho, st.lam
    if mode.robust:
        lam = st.lam
        ee = (resid.T @ resid).item()
        phi = lam / (lam + d) + ee / ((lam + d) * N)
        omega = (lam + (resid.T @ Sinv @ resid).item()) / (lam + d)
        V_new = mode.alpha * phi * V_new
        if mode.coef_update:
            P_new = mode.beta * omega * P_new
            Q_new = omega * st.Q
        rho_new = omega * (st.rho if rhok is None else rhok)
        lam_new = lam if mode.fixed_lambda else lam + d

    new = State(C=C_new, V=V_new, mu=mu_new.reshape(-1), P=P_new, Q=Q_new, rho=rho_new,
                lam=lam_new, theta=st.theta, gradsum=st.gradsum)
    info = StepInfo(y_pred=yp.reshape(-1), eta=float(eta), N=float(N), s=float(sv), phi=phi, omega=omega)
    return new, info


# --------------------------------------------------------------------------
# epoch / roll-out / optimiser
# --------------------------------------------------------------------------
def run_epoch(st: State, Y, mode: Mode, dyn: Dynamics, Qs=None, rhos=None, k0=0, step=lowrank_step,
              keep=(), want_grad=True):
    """Run steps k0+1 .. k0+T over Y (T, d) (time-major).  Returns (State, Y_pred (T,d), trace).

    Qs / rhos: None (constant st.Q / st.rho), or callables k -> value (PSMFIter's
    Q[k], R[k] dictionaries).  For rPSMF they must be None (running values are used).
    keep: iterable of step indices whose (State, StepInfo) should be recorded.
    """
    T, d = Y.shape
    Y_pred = np.empty((T, d))
    trace = {}
    keep = set(keep)
    for j in range(T):
        k = k0 + j + 1
        Qk = None if Qs is None else Qs(k)
        rk = None if rhos is None else rhos(k)
        if step is lowrank_step:
            st, info = step(st, Y[j], k, mode, dyn, Qk=Qk, rhok=rk, want_grad=want_grad)
        else:
            st, info = step(st, Y[j], k, mode, dyn, Qk=Qk, rhok=rk)
        Y_pred[j] = info.y_pred
        if k in keep:
            trace[k] = (st.copy(), info)
    return st, Y_pred, trace


def predict_rollout(C_T, mu_T, theta, dyn: Dynamics, T, n_pred):
    """psmf.py:182-188: mu rolled forward with f, y_hat = C_T mu_pred.  Returns (n_pred, d)."""
    out = np.empty((n_pred, C_T.shape[0]))
    mu = mu_T
    th = theta if theta is not None else np.zeros(0)
    for j, k in enumerate(range(T + 1, T + n_pred + 1)):
        mu = dyn.f(th, mu, k)
        out[j] = C_T @ mu
    return out


def adam_update(theta, gradsum, m, v, i, lr=1e-3, b1=0.9, b2=0.999, project=True):
    """psmf.py:224-242 (bias correction with the epoch index i, eps 1e-8 outside the sqrt)."""
    m = b1 * m + (1.0 - b1) * gradsum
    v = b2 * v + (1.0 - b2) * gradsum * gradsum
    m_hat = m / (1.0 - b1**i)
    v_hat = v / (1.0 - b2**i)
    theta = theta - lr * m_hat / (np.sqrt(v_hat) + 1e-8)
    if project:
        theta = np.maximum(theta, 0.0)
    return theta, m, v


def sgd_update(theta, gradsum, lr=1e-3, project=True):
    """psmf.py:244-248."""
    theta = theta - lr * gradsum
    if project:
        theta = np.maximum(theta, 0.0)
    return theta


# --------------------------------------------------------------------------
# synthetic series (semantics of ExperimentSynthetic/data.py:6-60)
# --------------------------------------------------------------------------
def synthetic_series(d, r, T, seed, var=0.1, noise="normal", dof=3.0, dtype=np.float32, chunk=512):
    """y_t = C_true x_t + sqrt(var) eps_t,  x_t = cos(2 pi theta_true t + x_{t-1}).

    Same generative model as data.py:6-31 (normal) / :34-60 (Student-t, dof 3), but
    vectorised over time and drawn from a seeded Generator (the reference's global-RNG
    draw order is irrelevant at benchmark sizes).  Returns Y (T, d) time-major.
    """
    rng = np.random.default_rng(seed)
    C_true = rng.standard_normal((d, r))
    theta_true = 1e-3 * np.arange(1, r + 1)
    x = rng.standard_normal(r)
    X = np.empty((T, r))
    for t in range(1, T + 1):
        x = np.cos(2.0 * np.pi * theta_true * t + x)
        X[t - 1] = x
    Y = np.empty((T, d), dtype=dtype)
    sd = np.sqrt(var)
    for a in range(0, T, chunk):
        b = min(T, a + chunk)
        if noise == "normal":
            eps = rng.standard_normal((b - a, d))
        else:
            eps = rng.standard_t(dof, (b - a, d))
        Y[a:b] = (X[a:b] @ C_true.T + sd * eps).astype(dtype)
    return Y
