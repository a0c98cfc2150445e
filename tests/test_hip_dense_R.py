"""A non-diagonal observation covariance R on the device (psmf.py:150-152, rpsmf.py:150-152: the reference's dense d x d branch).
The handle works in the eigenbasis of R (psmf_set_noise_rotation: series and C rotated at the boundary, the non-uniform-diagonal
step in between; tests/test_dense_R.py shows the identity on the CPU) -- against what the reference itself computed
(tests/golden/{psmf,rpsmf}_dense_R.npz) and, at a size where the d x d inverse still runs in seconds, against the oracle's dense
step.  GPU only: `pytest -m gpu`."""

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import psmf_oracle as O
import rpsmf_amd as psmf

pytestmark = pytest.mark.gpu


def _capi():
    from rpsmf_amd import _capi

    return _capi


def ydict(Y):
    return {k + 1: Y[k][:, None].copy() for k in range(Y.shape[0])}


def _make(robust, C0, V0, mu0, P0, Q, R0, lam0, T, **kw):
    a = (np.zeros((0, 1)), C0, V0, np.asarray(mu0).reshape(-1, 1), P0)
    if robust:
        return psmf.rPSMFIter(*a, Q, R0.copy(), lam0, psmf.RandomWalk(), **kw)
    return psmf.PSMFIter(*a, {k: Q for k in range(T + 1)}, {k: R0 for k in range(T + 1)}, psmf.RandomWalk(), **kw)


@pytest.mark.parametrize("name,robust", [("psmf_dense_R", False), ("rpsmf_dense_R", True)])
def test_dense_R_vs_reference_answers(name, robust):
    """The class surface with the fixture's dense R0, two epochs, float64 storage: state and y_hat against the reference's own run."""
    g = load_golden(name)
    Y, R0 = g["Y"], g["R0"]
    T, d = Y.shape
    f = _make(robust, g["C0"], g["V0"], g["mu0"], g["P0"], g["Q"], R0, float(g["lambda0"]), T, storage="f64")
    f.optim_init()
    for i in (1, 2):
        f.step(ydict(Y), i, T)
        assert f._dev.geometry()["engine"] == "step"
        p = f"s_e{i}_k{T}_"
        assert relerr(f._C[T], g[p + "C"]) < 1e-9 and relerr(f._V[T], g[p + "V"]) < 1e-9
        assert relerr(f._P[T], g[p + "P"]) < 1e-9 and relerr(f._mu[T], g[p + "mu"].reshape(-1, 1)) < 1e-9
        if robust:
            assert relerr(np.asarray(f._R[T])[0, 0], g[p + "rho"]) < 1e-9 and relerr(f._Q[T], g[p + "Q"]) < 1e-9
        f.optim_update(i)
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + 1)])
    assert relerr(yp, g["y_pred_e2"]) < 1e-9


def _dense_problem(d, r, T, seed, heavy):
    rng = np.random.default_rng(seed)
    Ct = rng.standard_normal((d, r))
    B = rng.standard_normal((d, 12)) / np.sqrt(12.0)
    R0 = np.diag(0.2 + rng.random(d)) + 0.7 * (B @ B.T)            # low-rank-plus-diagonal: strongly correlated rows
    L = np.linalg.cholesky(R0)
    x = rng.standard_normal(r)
    Y = np.empty((T, d))
    for t in range(T):
        x = x + 0.1 * rng.standard_normal(r)
        Y[t] = Ct @ x + L @ (rng.standard_t(3.0, d) if heavy else rng.standard_normal(d))
    return Y, 0.1 * rng.standard_normal((d, r)), R0


@pytest.mark.parametrize("storage,tol", [("f64", 1e-8), ("f32", 1e-5)])
@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_dense_R_vs_oracle_dense_step(robust, storage, tol):
    """d = 700 (not a multiple of the rotation's 64-wide tiles), r = 9: the oracle inverts the 700 x 700 innovation covariance at
    every step; the device never forms it.  run() with a roll-out: the predictions beyond T come back in original coordinates."""
    d, r, T, n_pred = 700, 9, 40, 6
    Y, C0, R0 = _dense_problem(d, r, T + n_pred, 5 + robust, robust)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    f = _make(robust, C0, V0, np.zeros(r), P0, Q, R0, 1.8, T, storage=storage)
    f.run(ydict(Y), T, 2, n_pred)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=R0.copy(), lam=1.8)
    for i in (1, 2):
        if robust:
            st.Q, st.rho, st.lam = Q, R0.copy(), 1.8
        st, Yp, _ = O.run_epoch(st, Y[:T], O.Mode(robust=robust), O.RandomWalkDyn(), step=O.literal_step)
    assert relerr(f._C[T], st.C) < tol and relerr(f._V[T], st.V) < tol and relerr(f._P[T], st.P) < tol
    assert relerr(f._mu[T], st.mu.reshape(-1, 1)) < tol
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp[:T], Yp) < tol
    assert relerr(yp[T:], O.predict_rollout(st.C, st.mu, None, O.RandomWalkDyn(), T, n_pred)) < tol


def test_rotation_boundary_of_the_handle():
    """psmf_set_noise_rotation at the C-ABI: C goes in as U^T C and comes back as U (U^T C) = C (odd sizes: the GEMM's ragged tiles),
    projections come back in original coordinates, and what the rotation cannot do is refused."""
    c = _capi()
    d, r = 131, 7
    rng = np.random.default_rng(3)
    A = rng.standard_normal((d, d))
    R0 = A @ A.T / d + 0.1 * np.eye(d)
    lam, U = np.linalg.eigh(R0)
    C0 = rng.standard_normal((d, r))
    for storage, tol in (("f64", 1e-13), ("f32", 1e-6)):
        h = c.DeviceFilter(d, r, storage=storage, nonuniform_R=True, engine="step")
        h.set_noise_rotation(U, lam)
        h.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0)
        assert relerr(h.get_state(want_C=True)["C"], C0) < tol
        mu = rng.standard_normal((5, r))
        assert relerr(h.project(mu), mu @ C0.T) < tol
        with pytest.raises(c.PsmfError):
            h.set_noise_rotation(U, lam)               # after set_state: the resident C is already rotated
        with pytest.raises(c.PsmfError):
            h.comm_init_host(2, 0, lambda buf: None)   # one shard by construction
        h.close()
    h = c.DeviceFilter(d, r, storage="f64", nonuniform_R=True, engine="step")
    with pytest.raises(ValueError):
        h.set_noise_rotation(U + 0.01, lam)            # not orthonormal
    with pytest.raises(ValueError):
        h.set_noise_rotation(U, lam - lam[-1])         # negative eigenvalues
    h.close()
    h = c.DeviceFilter(d, r, storage="f64", engine="step")
    with pytest.raises(c.PsmfError):
        h.set_noise_rotation(U, lam)                   # nonuniform_R = 0
    h.close()


def test_dense_R_that_the_device_refuses():
    """A time-varying dense R and an indefinite 'covariance' have no rotation that serves every step: refused by name, never run wrong."""
    d, r, T = 30, 3, 5
    rng = np.random.default_rng(8)
    A = rng.standard_normal((d, d))
    R0 = A @ A.T / d + 0.1 * np.eye(d)
    Y = rng.standard_normal((T, d))
    C0 = rng.standard_normal((d, r))
    a = (np.zeros((0, 1)), C0, 0.1 * np.eye(r), np.zeros((r, 1)), np.eye(r), {k: 0.1 * np.eye(r) for k in range(T + 1)})
    f = psmf.PSMFIter(*a, {k: R0 * (1.0 + 0.1 * k) + 0.01 * np.diag(np.arange(d) * k) for k in range(T + 1)}, psmf.RandomWalk())
    f.optim_init()
    with pytest.raises(NotImplementedError):
        f.step(ydict(Y), 1, T)
    f = psmf.PSMFIter(*a, {k: R0 - 2.0 * np.eye(d) for k in range(T + 1)}, psmf.RandomWalk())
    f.optim_init()
    with pytest.raises(NotImplementedError):
        f.step(ydict(Y), 1, T)
