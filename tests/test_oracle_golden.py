"""The CPU oracle against the golden fixtures produced by running the reference
(tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import psmf_oracle as O

TOL = 1e-11  # float64 round-off of two different but algebraically equal evaluations
TOL_GRAD = 1e-6  # the fixture's theta-gradient comes from a finite-difference autograd stub


def _check_snap(g, ep, k, st, info, robust, tol=TOL):
    p = f"s_e{ep}_k{k}_"
    assert relerr(st.C, g[p + "C"]) < tol
    assert relerr(st.V, g[p + "V"]) < tol
    assert relerr(st.mu, g[p + "mu"].reshape(-1)) < tol
    assert relerr(st.P, g[p + "P"]) < tol or np.max(np.abs(g[p + "P"])) == 0.0
    assert relerr(info.eta, g[p + "eta"]) < tol
    assert relerr(info.N, g[p + "N"]) < tol
    if robust:
        assert relerr(st.lam, g[p + "lam"]) < tol
        assert relerr(st.rho, g[p + "rho"]) < tol
        assert relerr(st.Q, g[p + "Q"]) < tol or np.max(np.abs(g[p + "Q"])) == 0.0


@pytest.mark.parametrize("step", [O.lowrank_step, O.literal_step], ids=["lowrank", "literal"])
@pytest.mark.parametrize("name,robust", [("psmf_full_rw", False), ("rpsmf_full_rw", True)])
def test_full_filter_random_walk(name, robust, step):
    """PSMFIter / rPSMFIter, all hooks as shipped, RandomWalk, two epochs (d=20, r=5, T=200)."""
    g = load_golden(name)
    Y = g["Y"]
    st = O.State(C=g["C0"], V=g["V0"], mu=g["mu0"], P=g["P0"], Q=g["Q"], rho=float(g["rho"]),
                 lam=float(g["lambda0"]))
    mode = O.Mode(robust=robust)
    keep = (1, 2, 10, 200)
    for ep in (1, 2):
        if robust:  # rpsmf.py:106-114: lambda, R, Q are reset at every epoch start
            st.Q, st.rho, st.lam = g["Q"], float(g["rho"]), float(g["lambda0"])
        st, Yp, tr = O.run_epoch(st, Y, mode, O.RandomWalkDyn(), keep=keep, step=step)
        for k in keep:
            _check_snap(g, ep, k, tr[k][0], tr[k][1], robust)
    assert relerr(Yp, g["y_pred_e2"]) < TOL


def _run_experiment(g, robust, dyn, mode, n_iter, T, n_pred, Q, rho, reset_V, keep, Y, project=True,
                    check=None):
    theta = g["theta0"].copy()
    st = O.State(C=g["C0"], V=g["V0"], mu=g["mu0"], P=g["P0"], Q=Q, rho=rho,
                 lam=float(g["lambda0"]) if robust else 0.0, theta=theta)
    m = np.zeros_like(theta)
    v = np.zeros_like(theta)
    thetas = [theta.copy()]
    grads = []
    for i in range(1, n_iter + 1):
        st.gradsum = np.zeros_like(theta)
        if reset_V:
            st.V = g["V0"]
        if robust:
            st.Q, st.rho, st.lam = Q, rho, float(g["lambda0"])
        st, Yp, tr = O.run_epoch(st, Y[:T], mode, dyn, keep=keep)
        if check is not None:
            for k in keep:
                check(i, k, tr[k])
        Ypred = O.predict_rollout(st.C, st.mu, st.theta, dyn, T, n_pred)
        grads.append(st.gradsum.copy())
        theta, m, v = O.adam_update(st.theta, st.gradsum, m, v, i, lr=1e-3, project=project)
        st.theta = theta
        thetas.append(theta.copy())
    return st, np.vstack([Yp, Ypred]), np.array(thetas), np.array(grads)


@pytest.mark.parametrize("name,robust", [("psmf_simplified_cos", False), ("rpsmf_simplified_cos", True)])
def test_synthetic_simplified(name, robust):
    """ExperimentSynthetic subclasses: P = 0, eta = tr(R)/d, no coefficient update, V reset per epoch,
    cos dynamics, Adam on theta (synthetic_psmf.py:47-100, synthetic_rpsmf.py:51-118)."""
    g = load_golden(name)
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])
    r = g["C0"].shape[1]
    mode = O.Mode(robust=robust, coef_update=False, eta_full=False, pbar_predict=False)
    keep = (1, 2, 10, T)

    def check(i, k, rec):
        # epoch 1 is independent of the fixture's finite-difference theta gradient
        _check_snap(g, i, k, rec[0], rec[1], robust, tol=1e-10 if i == 1 else 1e-6)

    st, Yall, thetas, grads = _run_experiment(
        g, robust, O.CosPhaseDyn(r), mode, n_iter, T, n_pred, np.zeros((r, r)), 1.0, True, keep,
        g["Y_obs"], check=check)
    assert relerr(grads, g["gradsum"]) < TOL_GRAD
    assert relerr(thetas, g["theta"]) < 1e-5
    assert relerr(Yall, g["y_pred_last"]) < 1e-6
    Yobs = g["Y_obs"]
    assert relerr(np.linalg.norm(Yall - Yobs), g["E_y"][-1]) < 1e-6
    assert relerr(np.linalg.norm(Yall[:T] - Yobs[:T]), g["E_train"][-1]) < 1e-6
    assert relerr(np.linalg.norm(Yall[T:] - Yobs[T:]), g["E_pred"][-1]) < 1e-6


def test_full_filter_fourier_basis():
    """Un-simplified PSMFIter with FourierBasis(N=1), r=1 (Beijing configuration), analytic
    derivatives by complex step, Adam with projection."""
    g = load_golden("psmf_full_fourier")
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])

    def fourier(theta, x, t):  # own restatement of nonlinearities.py:117-150 at N=1, r=1
        A, D, b, c, e, f = [theta[i] for i in range(6)]
        return A * np.sin(2 * np.pi * b * t + c * x) + D * np.cos(2 * np.pi * e * t + f * x)

    dyn = O.CallableDyn(fourier, 6)
    g = dict(g)
    g["lambda0"] = 0.0
    keep = (1, 2, 10, T)

    def check(i, k, rec):
        _check_snap(g, i, k, rec[0], rec[1], False, tol=1e-7 if i > 1 else 1e-10)

    st, Yall, thetas, grads = _run_experiment(
        g, False, dyn, O.Mode(), n_iter, T, n_pred, g["Q"], float(g["rho"]), False, keep, g["Y"],
        check=check)
    assert relerr(grads[0], g["gradsum"][0]) < TOL_GRAD
    assert relerr(thetas, g["theta"]) < 1e-5
    assert relerr(Yall, g["y_pred_last"]) < 1e-5


@pytest.mark.parametrize("name,robust", [("psmf_recursive", False), ("rpsmf_recursive", True)])
def test_recursive(name, robust):
    """PSMFRecursive / rPSMFRecursive: gradient accumulated per step, Adam every `update_every`
    steps inside the time loop with bias correction index k (psmf.py:287-310)."""
    g = load_golden(name)
    T, n_pred, ue = int(g["T"]), int(g["n_pred"]), int(g["update_every"])
    r = g["C0"].shape[1]
    dyn = O.CosPhaseDyn(r)
    theta = g["theta0"].copy()
    st = O.State(C=g["C0"], V=g["V0"], mu=g["mu0"], P=g["P0"], Q=g["Q"], rho=float(g["rho"]),
                 lam=float(g["lambda0"]), theta=theta, gradsum=np.zeros_like(theta))
    mode = O.Mode(robust=robust)
    m = np.zeros_like(theta)
    v = np.zeros_like(theta)
    thetas = [theta.copy()]
    Yp = []
    for k in range(1, T + 1):
        st, info = O.lowrank_step(st, g["Y"][k - 1], k, mode, dyn)
        Yp.append(info.y_pred)
        if k % ue == 0:
            th, m, v = O.adam_update(st.theta, st.gradsum, m, v, k, lr=1e-3)
            st.theta = th
            st.gradsum = np.zeros_like(th)
        thetas.append(st.theta.copy())
    Ypred = O.predict_rollout(st.C, st.mu, st.theta, dyn, T, n_pred)
    assert relerr(np.array(thetas), g["theta"]) < 1e-6
    assert relerr(st.C, g["C_T"]) < 1e-7
    assert relerr(st.V, g["V_T"]) < 1e-7
    assert relerr(st.mu, g["mu_T"]) < 1e-7
    assert relerr(st.P, g["P_T"]) < 1e-7
    assert relerr(np.vstack([np.array(Yp), Ypred]), g["y_pred"]) < 1e-7


def test_lowrank_equals_literal_nonuniform_R():
    """Diagonal, non-uniform R: the O(d r^2) form equals the dense d x d form."""
    rng = np.random.default_rng(5)
    d, r = 37, 6
    st = O.State(C=rng.standard_normal((d, r)), V=0.3 * np.eye(r), mu=rng.standard_normal(r),
                 P=0.5 * np.eye(r), Q=0.1 * np.eye(r), rho=0.5 + rng.random(d), lam=2.5)
    for robust in (False, True):
        a, b = st.copy(), st.copy()
        for k in range(1, 30):
            y = rng.standard_normal(d)
            a, ia = O.lowrank_step(a, y, k, O.Mode(robust=robust), O.RandomWalkDyn())
            b, ib = O.literal_step(b, y, k, O.Mode(robust=robust), O.RandomWalkDyn())
        assert relerr(a.C, b.C) < 1e-10 and relerr(a.P, b.P) < 1e-10 and relerr(a.mu, b.mu) < 1e-10
        assert relerr(a.rho, b.rho) < 1e-10
