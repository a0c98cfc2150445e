"""Host classes, numpy back end (hooks executed one by one, O(d r^2)) against the golden fixtures
produced by the reference.  CPU only.  These tests read like a run of the reference's own
experiment scripts: same constructors, same run/step/predict/adam calls."""

import numpy as np
import pytest

from conftest import load_golden, relerr
import rpsmf_amd as psmf


def ydict(Y):
    return {k + 1: Y[k][:, None].copy() for k in range(Y.shape[0])}


@pytest.mark.parametrize("name,robust", [("psmf_full_rw", False), ("rpsmf_full_rw", True)])
def test_full_filter_classes(name, robust):
    g = load_golden(name)
    Y = g["Y"]
    T, d = Y.shape
    r = g["C0"].shape[1]
    theta0 = np.zeros((0, 1))
    mu0 = g["mu0"].reshape(-1, 1)
    if robust:
        f = psmf.rPSMFIter(theta0, g["C0"], g["V0"], mu0, g["P0"], g["Q"], np.eye(d), 1.8, psmf.RandomWalk(),
                           backend="numpy")
    else:
        f = psmf.PSMFIter(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: g["Q"] for k in range(T + 1)},
                          {k: np.eye(d) for k in range(T + 1)}, psmf.RandomWalk(), backend="numpy")
    f.optim_init()
    f.step(ydict(Y), 1, T)
    f.optim_update(1)
    f.step(ydict(Y), 2, T)
    assert relerr(f._C[T], g["s_e2_k200_C"]) < 1e-10
    assert relerr(f._V[T], g["s_e2_k200_V"]) < 1e-10
    assert relerr(f._mu[T], g["s_e2_k200_mu"]) < 1e-10
    assert relerr(f._P[T], g["s_e2_k200_P"]) < 1e-10
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + 1)])
    assert relerr(yp, g["y_pred_e2"]) < 1e-10
    if robust:
        assert relerr(f._lambda[T], g["s_e2_k200_lam"]) < 1e-12
        assert relerr(np.asarray(f._R[T])[0, 0], g["s_e2_k200_rho"]) < 1e-10


def cos_nl(theta, x, t):
    return np.cos(2 * np.pi * theta * t + x)


def _synthetic_subclass(base):
    """The hook overrides of ExperimentSynthetic (synthetic_psmf.py:78-100), written against OUR
    base classes exactly as a user of the reference would write them."""

    class Synth(base):
        def step_reset(self):
            super().step_reset()
            self._V = {0: self.V0}

        def _predictive_covariance(self, i, k):
            return self._P[k - 1]

        def _compute_eta_k(self, k, P_bar):
            return np.trace(self._R[k - 1]) / self._d

        def _compute_inverse_coefficient_innovation(self, k, mu_bar, P_bar):
            if not self.robust:
                return None
            Rbar = self._R[k - 1] + np.kron(mu_bar.T @ self._V[k - 1] @ mu_bar, np.eye(self._d))
            return np.linalg.inv(Rbar)

        def _update_coefficient_mean(self, k, yk, Skinv, mu_bar, P_bar):
            self._mu[k] = mu_bar

        def _update_coefficient_covariance(self, k, Skinv, P_bar, yk):
            self._P[k] = P_bar
            if self.robust:
                e = yk - self._y_pred[k]
                omega = (self._lambda[k - 1] + e.T @ Skinv @ e) / (self._lambda[k - 1] + self._d)
                self._Q[k] = self._Q[k - 1]
                self._R[k] = omega * self._R[k - 1]
                self._lambda[k] = self._lambda[k - 1] + self._d

        def _prune(self, k):
            del self._C[k - 1], self._V[k - 1], self._P[k - 1]

    return Synth


@pytest.mark.parametrize("name,robust", [("psmf_simplified_cos", False), ("rpsmf_simplified_cos", True)])
def test_experiment_synthetic_subclass_with_plain_callable(name, robust):
    """Config A plumbing: user subclass overriding hooks + a plain-callable nonlinearity, numpy back end;
    theta path (closed-form gradient + Adam) against the reference run with its own driver loop."""
    g = load_golden(name)
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])
    d, r = g["C0"].shape
    y_obs = ydict(g["Y_obs"])
    y_train = {k: y_obs[k] for k in range(1, T + 1)}
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    if robust:
        cls = _synthetic_subclass(psmf.rPSMFIter)
        f = cls(theta0, g["C0"], g["V0"], mu0, g["P0"], 0 * np.eye(r), np.eye(d), 1.8, cos_nl, backend="numpy")
    else:
        cls = _synthetic_subclass(psmf.PSMFIter)
        f = cls(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: 0 * np.eye(r) for k in range(T + 1)},
                {k: np.eye(d) for k in range(T + 1)}, cos_nl, backend="numpy")
    f.adam_init(gam=1e-3)
    for i in range(1, n_iter + 1):
        f.step(y_train, i, T)
        f.predict(i, T, n_pred)
        assert relerr(f._gradsum.reshape(-1), g["gradsum"][i - 1]) < 1e-5
        f.adam_update(i)
    theta = np.array([f._theta[i].reshape(-1) for i in range(n_iter + 1)])
    assert relerr(theta, g["theta"]) < 1e-5
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp, g["y_pred_last"]) < 1e-6
    mu = np.array([f._mu[k].reshape(-1) for k in range(0, T + 1)])   # kept by the _prune override
    assert relerr(mu, g["mu_last"]) < 1e-6


def test_fourier_basis_beijing_configuration():
    g = load_golden("psmf_full_fourier")
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])
    Y = g["Y"]
    d, r = g["C0"].shape
    nl = psmf.FourierBasis(rank=r, N=1)
    assert nl.n_params == g["theta0"].size
    f = psmf.PSMFIter(g["theta0"].reshape(-1, 1), g["C0"], g["V0"], g["mu0"].reshape(-1, 1), g["P0"],
                      {k: g["Q"] for k in range(T + 1)}, {k: np.eye(d) for k in range(T + 1)}, nl, backend="numpy")
    f.adam_init(gam=1e-3)
    for i in range(1, n_iter + 1):
        f.step(ydict(Y[:T]), i, T)
        f.predict(i, T, n_pred)
        f.adam_update(i, project=True)
    theta = np.array([f._theta[i].reshape(-1) for i in range(n_iter + 1)])
    assert relerr(theta, g["theta"]) < 1e-5
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp, g["y_pred_last"]) < 1e-5


@pytest.mark.parametrize("name,robust", [("psmf_recursive", False), ("rpsmf_recursive", True)])
def test_recursive_classes(name, robust):
    g = load_golden(name)
    T, n_pred, ue = int(g["T"]), int(g["n_pred"]), int(g["update_every"])
    d, r = g["C0"].shape
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    if robust:
        f = psmf.rPSMFRecursive(theta0, g["C0"], g["V0"], mu0, g["P0"], g["Q"], np.eye(d), 1.8, cos_nl, backend="numpy")
    else:
        f = psmf.PSMFRecursive(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: g["Q"] for k in range(T + 1)},
                               {k: np.eye(d) for k in range(T + 1)}, cos_nl, backend="numpy")
    f.run(ydict(g["Y"]), T, n_pred, update_every=ue)
    theta = np.array([f._theta[k].reshape(-1) for k in range(T + 1)])
    assert relerr(theta, g["theta"]) < 1e-6
    assert relerr(f._C[T], g["C_T"]) < 1e-7
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp, g["y_pred"]) < 1e-7


def test_scalar_and_vector_R_equal_dense_R():
    rng = np.random.default_rng(3)
    d, r, T = 15, 3, 12
    Y = rng.standard_normal((T, d))
    C0 = rng.standard_normal((d, r))
    rho = 0.5 + rng.random(d)
    outs = []
    for R in (np.diag(rho), rho):
        f = psmf.PSMFIter(np.zeros((0, 1)), C0, 0.2 * np.eye(r), np.zeros((r, 1)), np.eye(r),
                          {k: 0.1 * np.eye(r) for k in range(T + 1)}, {k: R for k in range(T + 1)},
                          psmf.RandomWalk(), backend="numpy")
        f.optim_init()
        f.step(ydict(Y), 1, T)
        outs.append((f._C[T], f._P[T], f._mu[T]))
    for a, b in zip(*outs):
        assert relerr(a, b) < 1e-12


def test_compute_scaling_factor_matches_reference():
    """rPSMFIter.compute_scaling_factor / use_scaling (rpsmf.py:45-51,75-104) against the reference's own values, and a run
    with the factors applied (numpy back end) against the reference's run."""
    g = load_golden("rpsmf_scaling")
    Y = g["Y"]
    T, d = Y.shape
    r = g["C0"].shape[1]
    mk = lambda lam0, **kw: psmf.rPSMFIter(np.zeros((0, 1)), g["C0"], g["V0"], g["mu0"].reshape(-1, 1), g["P0"], g["Q"],
                                           np.eye(d), lam0, psmf.RandomWalk(), backend="numpy", **kw)
    for (lam0, dim, off), ref in zip(g["scaling_triples"], g["scaling_values"]):
        assert abs(mk(float(lam0)).compute_scaling_factor(int(dim), int(off)) - ref) < 1e-12 * abs(ref)
    f = mk(1.8, use_scaling=True)
    assert abs(f._alpha - float(g["alpha"])) < 1e-12 and abs(f._beta - float(g["beta"])) < 1e-12
    f.optim_init()
    f.step(ydict(Y), 1, T)
    assert relerr(f._C[T], g["C_T"]) < 1e-10 and relerr(f._V[T], g["V_T"]) < 1e-10
    assert relerr(f._mu[T], g["mu_T"].reshape(-1, 1)) < 1e-10 and relerr(f._P[T], g["P_T"]) < 1e-10


def test_rpsmf_vector_R0_equals_dense_R0():
    """rPSMFIter scales R by omega every step: a (d,) diagonal must stay a (d,) diagonal (it used to turn into a (1, d)
    array and take the dense branch with a broadcast R)."""
    rng = np.random.default_rng(5)
    d, r, T = 12, 3, 6
    Y = rng.standard_normal((T, d))
    C0 = rng.standard_normal((d, r))
    outs = []
    for R0 in (2.0 * np.eye(d), np.full(d, 2.0), np.full((d, 1), 2.0), 2.0):
        f = psmf.rPSMFIter(np.zeros((0, 1)), C0, 0.2 * np.eye(r), np.zeros((r, 1)), np.eye(r), 0.1 * np.eye(r), R0, 1.8,
                           psmf.RandomWalk(), backend="numpy")
        f.optim_init()
        f.step(ydict(Y), 1, T)
        outs.append((f._C[T], f._P[T], f._V[T], f._mu[T]))
        assert np.shape(f._R[T]) == np.shape(R0)
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert relerr(a, b) < 1e-12
    with pytest.raises(ValueError):
        f = psmf.rPSMFIter(np.zeros((0, 1)), C0, 0.2 * np.eye(r), np.zeros((r, 1)), np.eye(r), 0.1 * np.eye(r), np.ones(d + 1), 1.8,
                           psmf.RandomWalk(), backend="numpy")
        f.optim_init()
        f.step(ydict(Y), 1, T)


def test_array_series_is_indexed_like_the_dict():
    """y may be an array-like: y[k] for k = 1..T, row 0 unused -- the same rule in both back ends."""
    rng = np.random.default_rng(6)
    d, r, T = 10, 2, 8
    Y = rng.standard_normal((T, d))
    C0 = rng.standard_normal((d, r))
    outs = []
    for y in (ydict(Y), np.vstack([np.full((1, d), np.nan), Y])):
        f = psmf.PSMFIter(np.zeros((0, 1)), C0, 0.2 * np.eye(r), np.zeros((r, 1)), np.eye(r),
                          {k: 0.1 * np.eye(r) for k in range(T + 1)}, {k: 1.0 for k in range(T + 1)}, psmf.RandomWalk(), backend="numpy")
        f.optim_init()
        f.step(y, 1, T)
        outs.append(f._C[T])
    assert np.array_equal(outs[0], outs[1])


def test_hip_backend_refuses_unrecognised_overrides_and_callables():
    class Custom(psmf.PSMFIter):
        def _compute_eta_k(self, k, P_bar):
            return 1.0

    args = (np.zeros((0, 1)), np.zeros((4, 2)), np.eye(2), np.zeros((2, 1)), np.eye(2), {0: np.eye(2)}, {0: 1.0})
    with pytest.raises(TypeError):
        Custom(*args, psmf.RandomWalk())                 # overrides without a declared hip_mode
    f = psmf.PSMFIter(*args, lambda th, x, t: np.tanh(x))    # arbitrary callable: accepted, f is evaluated on the host, one device step
    assert f._host_stepped()                             # at a time (psmf_step_host) -- the d-sized work stays on the device
    f = psmf.PSMFIter(*args, lambda th, x, t: x)         # ... unless it IS one of the closed-form families (probed, modes.py)
    assert not f._host_stepped() and isinstance(f._nl, psmf.RandomWalk)
    Custom(*args, psmf.RandomWalk(), backend="numpy")    # fine on the host

    class Declared(Custom):
        hip_mode = "full"                                # declares a mode that does not cover the override

    with pytest.raises(TypeError):
        Declared(*args, psmf.RandomWalk())

    class OwnLoop(psmf.PSMFRecursive):                   # synthetic_recursive_psmf.py:77-140 style: its own inner()
        def inner(self, k, yk):
            super().inner(k, yk)

    with pytest.raises(TypeError):
        OwnLoop(*args, psmf.RandomWalk())
    OwnLoop(*args, psmf.RandomWalk(), backend="numpy")
    with pytest.raises(NotImplementedError):
        psmf.PSMFIterMissing()
    with pytest.raises(NotImplementedError):
        psmf.rPSMFIterMissing()
    with pytest.raises(AssertionError):
        psmf.PSMFIter(*args, psmf.RandomWalk(), optim="lbfgs", backend="numpy")


def test_builtin_nonlinearities_numpy_backend_gradients():
    """ScaledWalk / Sinusoid / FourierBasis with their analytic Jacobians (numpy back end) against the same functions passed
    as plain callables (complex-step derivatives): identical epochs, gradients included."""
    rng = np.random.default_rng(4)
    d, r, T = 14, 3, 10
    Y = rng.standard_normal((T, d))
    C0 = rng.standard_normal((d, r))
    for nl in (psmf.ScaledWalk(r), psmf.Sinusoid(r), psmf.Sinusoid(r, scaled=False, phased=False), psmf.FourierBasis(r, N=2)):
        theta0 = (0.3 * rng.random(nl.n_params)).reshape(-1, 1)
        outs = []
        for fn in (nl, lambda th, x, t, nl=nl: nl(th, x, t)):
            f = psmf.PSMFIter(theta0, C0, 0.2 * np.eye(r), np.zeros((r, 1)), np.eye(r), {k: 0.1 * np.eye(r) for k in range(T + 1)},
                              {k: 1.0 for k in range(T + 1)}, fn, backend="numpy")
            f.optim_init()
            f.step(ydict(Y), 1, T)
            outs.append((f._C[T], f._P[T], f._gradsum))
        for a, b in zip(*outs):
            assert relerr(a, b) < 1e-10


def test_tracking_mixin_host_path():
    """errors_init / errors_update / log with the reference's attribute names and line format (tracking.py:20-76)."""
    class Tracked(psmf.TrackingMixin, psmf.PSMFIter):
        pass

    rng = np.random.default_rng(8)
    d, r, T, n_pred = 9, 2, 12, 4
    Y = rng.standard_normal((T + n_pred, d))
    y = ydict(Y)
    f = Tracked(np.zeros((0, 1)), rng.standard_normal((d, r)), 0.2 * np.eye(r), np.zeros((r, 1)), np.eye(r),
                {k: 0.1 * np.eye(r) for k in range(T + 1)}, {k: 1.0 for k in range(T + 1)}, psmf.RandomWalk(), backend="numpy")
    f.optim_init()
    f.errors_init(y, T, 1, n_pred)
    assert relerr(f._E_y[0], np.linalg.norm(Y)) < 1e-14 and relerr(f._E_pred[0], np.linalg.norm(Y[T:])) < 1e-14
    f.step({k: y[k] for k in range(1, T + 1)}, 1, T)
    f.predict(1, T, n_pred)
    f.errors_update(1, y, T, n_pred)
    Yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(f._E_y[1], np.linalg.norm(Yp - Y)) < 1e-13 and relerr(f._E_train[1], np.linalg.norm(Yp[:T] - Y[:T])) < 1e-13
    f.log(1, 1, 0.5, verbose=False)
    assert f._logs[0].startswith("[001/1] ||y - Cx||^2 = ") and f._logs[0].endswith("Δt = 0.500") and f._tracking_on_device == 0


def test_inverse_innovation_operator_matches_dense():
    from rpsmf_amd.linop import InverseInnovation

    rng = np.random.default_rng(0)
    d, r = 9, 3
    C = rng.standard_normal((d, r))
    w = 1.0 / (0.5 + rng.random(d))
    P = np.eye(r) * 0.7
    U = w[:, None] * C
    K = np.linalg.inv(np.linalg.inv(P) + C.T @ U)
    S = InverseInnovation(w, U, K)
    dense = np.linalg.inv(C @ P @ C.T + np.diag(1.0 / w))
    e = rng.standard_normal((d, 1))
    assert relerr(S @ e, dense @ e) < 1e-12
    assert relerr(e.T @ S @ e, e.T @ dense @ e) < 1e-12
    assert relerr(C.T @ (S @ C), C.T @ dense @ C) < 1e-12
    assert relerr(S.toarray(), dense) < 1e-12
