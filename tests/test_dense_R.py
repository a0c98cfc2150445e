"""A non-diagonal observation covariance R (psmf.py:150-152, rpsmf.py:150-152: the dense d x d branch of
`_compute_inverse_coefficient_innovation`) -- CPU part: the oracle's dense step and the numpy back end against what the reference
itself computed (tests/golden/{psmf,rpsmf}_dense_R.npz, made by make_golden.py:case_dense_R), and the identity the device route
rests on: with R = U diag(lam) U^T the whole recursion is the diagonal-R recursion on U^T y, U^T C (every update is equivariant
under an orthogonal change of the observation coordinates, eta and the likelihood's residual norm are invariant).
The device route itself: tests/test_hip_dense_R.py."""

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import psmf_oracle as O
import rpsmf_amd as psmf

KEEP = (1, 2, 7, 60)


def ydict(Y):
    return {k + 1: Y[k][:, None].copy() for k in range(Y.shape[0])}


def _check(g, ep, k, st, info, robust, tol, R0=None):
    p = f"s_e{ep}_k{k}_"
    assert relerr(st.C, g[p + "C"]) < tol
    assert relerr(st.V, g[p + "V"]) < tol
    assert relerr(st.mu, g[p + "mu"].reshape(-1)) < tol
    assert relerr(st.P, g[p + "P"]) < tol
    assert relerr(info.eta, g[p + "eta"]) < tol and relerr(info.N, g[p + "N"]) < tol
    if robust:
        assert relerr(st.lam, g[p + "lam"]) < tol and relerr(st.Q, g[p + "Q"]) < tol
        if R0 is not None:
            assert relerr(np.asarray(st.rho)[0, 0], g[p + "rho"]) < tol


@pytest.mark.parametrize("name,robust", [("psmf_dense_R", False), ("rpsmf_dense_R", True)])
def test_oracle_dense_step_vs_reference(name, robust):
    g = load_golden(name)
    Y, R0 = g["Y"], g["R0"]
    assert np.count_nonzero(R0 - np.diag(np.diag(R0))) > 0          # the reference did take its dense branch
    st = O.State(C=g["C0"], V=g["V0"], mu=g["mu0"], P=g["P0"], Q=g["Q"], rho=R0.copy(), lam=float(g["lambda0"]))
    for ep in (1, 2):
        if robust:
            st.Q, st.rho, st.lam = g["Q"], R0.copy(), float(g["lambda0"])
        st, Yp, tr = O.run_epoch(st, Y, O.Mode(robust=robust), O.RandomWalkDyn(), keep=KEEP, step=O.literal_step)
        for k in KEEP:
            _check(g, ep, k, tr[k][0], tr[k][1], robust, 1e-10, R0)
    assert relerr(Yp, g["y_pred_e2"]) < 1e-10


@pytest.mark.parametrize("name,robust", [("psmf_dense_R", False), ("rpsmf_dense_R", True)])
def test_rotated_diagonal_recursion_equals_dense(name, robust):
    """What the device does: eigen-decompose R once, run the O(d r^2) diagonal-R step on the rotated series, rotate C and y_hat back."""
    g = load_golden(name)
    Y, R0 = g["Y"], g["R0"]
    lam, U = np.linalg.eigh(R0)
    st = O.State(C=U.T @ g["C0"], V=g["V0"], mu=g["mu0"], P=g["P0"], Q=g["Q"], rho=lam.copy(), lam=float(g["lambda0"]))
    for ep in (1, 2):
        if robust:
            st.Q, st.rho, st.lam = g["Q"], lam.copy(), float(g["lambda0"])
        st, Yp, tr = O.run_epoch(st, Y @ U, O.Mode(robust=robust), O.RandomWalkDyn(), keep=KEEP, want_grad=False)
        for k in KEEP:
            s, info = tr[k]
            s = s.copy()
            s.C = U @ s.C
            _check(g, ep, k, s, info, robust, 1e-10)
            if robust:
                assert relerr(np.asarray(s.rho)[0] / lam[0] * R0[0, 0], g[f"s_e{ep}_k{k}_rho"]) < 1e-10
    assert relerr(Yp @ U.T, g["y_pred_e2"]) < 1e-10


@pytest.mark.parametrize("name,robust", [("psmf_dense_R", False), ("rpsmf_dense_R", True)])
def test_numpy_backend_dense_R_vs_reference(name, robust):
    g = load_golden(name)
    Y, R0 = g["Y"], g["R0"]
    T, d = Y.shape
    r = g["C0"].shape[1]
    a = (np.zeros((0, 1)), g["C0"], g["V0"], g["mu0"].reshape(-1, 1), g["P0"])
    if robust:
        f = psmf.rPSMFIter(*a, g["Q"], R0.copy(), float(g["lambda0"]), psmf.RandomWalk(), backend="numpy")
    else:
        f = psmf.PSMFIter(*a, {k: g["Q"] for k in range(T + 1)}, {k: R0 for k in range(T + 1)}, psmf.RandomWalk(), backend="numpy")
    f.optim_init()
    f.step(ydict(Y), 1, T)
    f.optim_update(1)
    f.step(ydict(Y), 2, T)
    p = f"s_e2_k{T}_"
    assert relerr(f._C[T], g[p + "C"]) < 1e-10 and relerr(f._V[T], g[p + "V"]) < 1e-10
    assert relerr(f._P[T], g[p + "P"]) < 1e-10 and relerr(f._mu[T], g[p + "mu"].reshape(-1, 1)) < 1e-10
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + 1)])
    assert relerr(yp, g["y_pred_e2"]) < 1e-10
