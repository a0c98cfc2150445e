"""The ExperimentSynthetic classes written the way the reference writes them -- `from psmf import PSMFIter`,
`from psmf.tracking import TrackingMixin`, six overridden hooks, a `_prune` that keeps `_mu`, a plain function as the
nonlinearity, the reference's eight / eleven positional constructor arguments and nothing else -- run on the DEVICE
(default backend), because the constructor recognises what the hooks compute (rpsmf_amd/modes.py).

Own re-typing of the call pattern of ExperimentSynthetic/synthetic_psmf.py:46-102,105-106,135 and
synthetic_rpsmf.py:50-120,123-124,156-158; expected values: the reference's own run (tests/golden/*_simplified_cos.npz).
"""

import time

import numpy as np
import pytest

from conftest import load_golden, relerr

# the reference's import lines
from psmf import PSMFIter, rPSMFIter
from psmf.tracking import TrackingMixin


def _experiment_run(self, y, y_obs, theta_true, C_true, x_true, T, n_iter, n_pred, adam_gam=1e-3, live_plot=False, verbose=True):
    """The driver loop both experiment classes define as their `run` (synthetic_psmf.py:47-74)."""
    self.adam_init(gam=adam_gam)
    self.errors_init(y_obs, T, n_iter, n_pred, theta_true=theta_true)
    self.figures_init(live_plot=live_plot)
    self.log(0, n_iter, 0, verbose=verbose)
    for i in range(1, n_iter + 1):
        t_start = time.time()
        self.step(y, i, T)
        self.predict(i, T, n_pred)
        self.adam_update(i)
        self.errors_update(i, y_obs, T, n_pred, theta_true=theta_true)
        self.log(i, n_iter, time.time() - t_start, verbose=verbose)
        self.figures_update(y_obs, T, n_pred, live_plot=live_plot, x_true=x_true)


class PSMFIterSynthetic(TrackingMixin, PSMFIter):
    run = _experiment_run

    def step_reset(self):
        super().step_reset()
        self._V = {0: self.V0}

    def _predictive_covariance(self, i, k):
        return self._P[k - 1]

    def _compute_eta_k(self, k, P_bar):
        return np.trace(self._R[k - 1]) / self._d

    def _compute_inverse_coefficient_innovation(self, k, mu_bar, P_bar):
        pass

    def _update_coefficient_mean(self, k, yk, Skinv, mu_bar, P_bar):
        self._mu[k] = mu_bar

    def _update_coefficient_covariance(self, k, Skinv, P_bar, yk):
        self._P[k] = P_bar

    def _prune(self, k):
        del self._C[k - 1], self._V[k - 1], self._P[k - 1]


class rPSMFIterSynthetic(TrackingMixin, rPSMFIter):
    run = _experiment_run

    def step_reset(self):
        super().step_reset()
        self._V = {0: self.V0}

    def _predictive_covariance(self, i, k):
        return self._P[k - 1]

    def _compute_eta_k(self, k, P_bar):
        return np.trace(self._R[k - 1]) / self._d

    def _compute_inverse_coefficient_innovation(self, k, mu_bar, P_bar):
        Rbar = self._R[k - 1] + np.kron(mu_bar.T @ self._V[k - 1] @ mu_bar, np.eye(self._d))
        return np.linalg.inv(Rbar)

    def _update_coefficient_mean(self, k, yk, Skinv, mu_bar, P_bar):
        self._mu[k] = mu_bar

    def _update_coefficient_covariance(self, k, Skinv, P_bar, yk):
        omega_k = (self._lambda[k - 1] + (yk - self._y_pred[k]).T @ Skinv @ (yk - self._y_pred[k])) / (self._lambda[k - 1] + self._d)
        self._P[k] = P_bar
        self._Q[k] = self._Q[k - 1]
        self._R[k] = omega_k * self._R[k - 1]
        if not self.fixed_lambda:
            self._lambda[k] = self._lambda[k - 1] + self._d

    def _prune(self, k):
        del self._C[k - 1], self._V[k - 1], self._P[k - 1]


def nonlinearity(theta, x, t):
    return np.cos(2 * np.pi * theta * t + x)


def _construct(name):
    g = load_golden(name)
    T = int(g["T"])
    d, r = g["C0"].shape
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    if name.startswith("rpsmf"):
        f = rPSMFIterSynthetic(theta0, g["C0"], g["V0"], mu0, g["P0"], 0 * np.identity(r), np.identity(d), float(g["lambda0"]), nonlinearity)
    else:
        Qs = {k: 0 * np.identity(r) for k in range(T + 1)}
        Rs = {k: np.identity(d) for k in range(T + 1)}
        f = PSMFIterSynthetic(theta0, g["C0"], g["V0"], mu0, g["P0"], Qs, Rs, nonlinearity)
    return f, g


# ------------------------------------------------------------------------------------------------ host (no GPU)
@pytest.mark.parametrize("name", ["psmf_simplified_cos", "rpsmf_simplified_cos"])
def test_reference_style_subclass_is_recognised(name):
    """Construction with the reference's positional arguments only: default backend "hip", no hip_mode attribute, no
    backend keyword -- the hooks are recognised as the device's "simplified" mode and the plain function as cos-phase dynamics
    (evaluated inside the device loop).  No GPU is touched by construction."""
    from rpsmf_amd import _capi, nonlinearities as NL

    f, g = _construct(name)
    assert f.backend == "hip" and f.hip_mode == "simplified" and f.hip_mode_recognised
    assert isinstance(f._nl, NL.CosPhase) and f._nl.recognised_from is nonlinearity
    assert not f._host_stepped() and f._device_kwargs()["dyn_kind"] == _capi.DYN_COS_PHASE
    kw = f._device_kwargs()
    assert (kw["coef_update"], kw["eta_full"], kw["pbar_predict"]) == (False, False, False)


def test_hooks_that_match_no_device_mode_are_refused():
    """Negative control of the probe: a subclass whose eta is NOT one of the device modes' still raises (never a silent
    wrong answer), one that re-states the full mode in its own words is accepted as "full"."""
    r, d = 4, 9

    class Wrong(PSMFIter):
        def _compute_eta_k(self, k, P_bar):
            return 2.0 * np.trace(self._R[k]) / self._d

    class SameAsFull(PSMFIter):
        def _compute_eta_k(self, k, P_bar):
            C = self._C[k - 1]
            return np.trace(self._R[k] + C @ P_bar @ C.T) / self._d          # psmf.py:121-125 literally (d x d)

        def _compute_dictionary_innovation(self, k, eta_k, mu_bar, P_bar):
            return eta_k + mu_bar.T @ self._V[k - 1] @ mu_bar

    class StepIndexOff(PSMFIter):
        def _compute_eta_k(self, k, P_bar):                                  # full-mode formula with R[k-1] instead of R[k]
            C = self._C[k - 1]
            return np.trace(self._R[k - 1] + C @ P_bar @ C.T) / self._d

    args = (np.zeros((0, 1)), np.zeros((d, r)), np.eye(r), np.zeros((r, 1)), np.eye(r), {0: np.eye(r)}, {0: np.eye(d)}, lambda th, x, t: x)
    with pytest.raises(TypeError, match="match none"):
        Wrong(*args)
    with pytest.raises(TypeError, match="match none"):
        StepIndexOff(*args)
    f = SameAsFull(*args)
    assert f.hip_mode == "full" and f.hip_mode_recognised
    from rpsmf_amd import nonlinearities as NL

    assert isinstance(f._nl, NL.RandomWalk)


def test_unknown_callable_stays_host_stepped():
    from rpsmf_amd import modes

    assert modes.recognise_nonlinearity(lambda th, x, t: np.tanh(th[:3] * x), 3, 3) is None
    assert modes.recognise_nonlinearity(lambda th, x, t: np.cos(2 * np.pi * th * t + 1.0001 * x), 3, 3) is None
    fb = modes.recognise_nonlinearity(lambda th, x, t: th[0:1] * np.sin(2 * np.pi * th[2:3] * t + th[3:4] * x) + th[1:2] * np.cos(2 * np.pi * th[4:5] * t + th[5:6] * x), 6, 1)
    from rpsmf_amd import nonlinearities as NL

    assert isinstance(fb, NL.FourierBasis) and fb.N == 1      # ExperimentBeijing's family at r = 1


@pytest.mark.parametrize("name", ["psmf_simplified_cos", "rpsmf_simplified_cos"])
def test_mode_classes_reproduce_the_reference_run_on_the_host(name):
    """modes.SimplifiedPSMF / SimplifiedRPSMF (the library's statement of the simplified mode, what the probe compares
    with) against the reference's own run, numpy back end."""
    from rpsmf_amd import modes

    g = load_golden(name)
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])
    d, r = g["C0"].shape
    robust = name.startswith("rpsmf")
    y = {k + 1: g["Y_obs"][k][:, None] for k in range(T)}
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    cls = modes.mode_class("simplified", robust)

    class WithReset(cls):
        def step_reset(self):
            super().step_reset()
            self._V = {0: self.V0}

        def _prune(self, k):
            del self._C[k - 1], self._V[k - 1], self._P[k - 1]

    if robust:
        f = WithReset(theta0, g["C0"], g["V0"], mu0, g["P0"], 0 * np.eye(r), np.eye(d), float(g["lambda0"]), nonlinearity, backend="numpy")
    else:
        f = WithReset(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: 0 * np.eye(r) for k in range(T + 1)}, {k: np.eye(d) for k in range(T + 1)},
                      nonlinearity, backend="numpy")
    f.adam_init(gam=1e-3)
    for i in range(1, n_iter + 1):
        f.step(y, i, T)
        f.predict(i, T, n_pred)
        f.adam_update(i)
    theta = np.array([f._theta[i].reshape(-1) for i in range(n_iter + 1)])
    assert relerr(theta, g["theta"]) < 1e-5
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp, g["y_pred_last"]) < 1e-6


# ------------------------------------------------------------------------------------------------ device
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["psmf_simplified_cos", "rpsmf_simplified_cos"])
def test_reference_style_experiment_runs_unchanged_on_the_device(name):
    """The experiment's own `run` (adam_init, errors_init, log, step, predict, adam_update, errors_update per epoch) on the
    device: theta trajectory, predictions, the mean history `_mu[0..T]` and the three error norms of every epoch against
    the reference's run; the time loop ran on the GPU (blocked engine, cos-phase dynamics in the kernel), the error norms
    were reduced there."""
    f, g = _construct(name)
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])
    y_obs = {k + 1: g["Y_obs"][k][:, None] for k in range(T + n_pred)}
    y_train = {k: y_obs[k] for k in range(1, T + 1)}
    f.run(y_train, y_obs, g["theta_true"].reshape(-1, 1), None, None, T, n_iter, n_pred, adam_gam=1e-3, verbose=False)
    tol = 1e-6          # storage "auto": f32 under the blocked engine; d = 20: the state is r-sized float64
    geo = f._dev.geometry()
    assert geo["engine"] == "block" and not f._host_stepped()
    theta = np.array([f._theta[i].reshape(-1) for i in range(n_iter + 1)])
    assert relerr(theta, g["theta"]) < 10 * tol
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp, g["y_pred_last"]) < 10 * tol
    mu = np.array([f._mu[k].reshape(-1) for k in range(0, T + 1)])
    assert relerr(mu, g["mu_last"]) < 10 * tol
    for key, attr in (("E_y", "_E_y"), ("E_train", "_E_train"), ("E_pred", "_E_pred"), ("E_theta", "_E_theta")):
        mine = np.array([getattr(f, attr)[i] for i in range(n_iter + 1)])
        assert relerr(mine, g[key]) < 10 * tol, key
    assert f._tracking_on_device == n_iter
    assert len(f._logs) == n_iter + 1 and f._logs[1].startswith("[001/")
