"""GPU: the host classes with backend="hip", and the masked batched engine, against the golden
fixtures / the reference's stored known answers / the CPU oracle.  `pytest -m gpu`."""

import json

import numpy as np
import pytest

from conftest import load_golden, relerr
import rpsmf_amd as psmf
from rpsmf_amd import impute, impute_harness as H
from oracle.impute_oracle import impute_filter

pytestmark = pytest.mark.gpu


def ydict(Y):
    return {k + 1: Y[k][:, None].copy() for k in range(Y.shape[0])}


@pytest.mark.parametrize("name,robust", [("psmf_full_rw", False), ("rpsmf_full_rw", True)])
def test_classes_full_filter_on_device(name, robust):
    """Same driver calls as the numpy-backend test, default backend (hip), float64 storage."""
    g = load_golden(name)
    Y = g["Y"]
    T, d = Y.shape
    r = g["C0"].shape[1]
    theta0, mu0 = np.zeros((0, 1)), g["mu0"].reshape(-1, 1)
    if robust:
        f = psmf.rPSMFIter(theta0, g["C0"], g["V0"], mu0, g["P0"], g["Q"], np.eye(d), 1.8, psmf.RandomWalk(),
                           storage="f64")
    else:
        f = psmf.PSMFIter(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: g["Q"] for k in range(T + 1)},
                          {k: np.eye(d) for k in range(T + 1)}, psmf.RandomWalk(), storage="f64")
    f.optim_init()
    y = ydict(Y)
    f.step(y, 1, T)
    f.optim_update(1)
    f.step(y, 2, T)
    assert relerr(f._C[T], g["s_e2_k200_C"]) < 1e-9
    assert relerr(f._V[T], g["s_e2_k200_V"]) < 1e-9
    assert relerr(f._mu[T], g["s_e2_k200_mu"]) < 1e-9
    for k in (1, 2, 10):     # mean history (kept by the experiments' _prune overrides), fetched lazily
        assert f._mu[k].shape == (r, 1) and relerr(f._mu[k], g[f"s_e2_k{k}_mu"]) < 1e-9
    assert 0 in f._mu and T + 1 not in f._mu
    assert relerr(f._P[T], g["s_e2_k200_P"]) < 1e-9
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + 1)])
    assert relerr(yp, g["y_pred_e2"]) < 1e-9
    assert relerr(f.sq_errors(T), np.sum((g["y_pred_e2"] - Y) ** 2)) < 1e-9


def test_use_scaling_on_device_vs_reference_run():
    """rPSMFIter(use_scaling=True): alpha, beta computed as the reference computes them and applied in the device loop
    (rpsmf.py:45-51,133-171), against the reference's own run."""
    g = load_golden("rpsmf_scaling")
    Y = g["Y"]
    T, d = Y.shape
    r = g["C0"].shape[1]
    f = psmf.rPSMFIter(np.zeros((0, 1)), g["C0"], g["V0"], g["mu0"].reshape(-1, 1), g["P0"], g["Q"], np.eye(d), 1.8,
                       psmf.RandomWalk(), use_scaling=True, storage="f64")
    assert abs(f._alpha - float(g["alpha"])) < 1e-12 and abs(f._beta - float(g["beta"])) < 1e-12
    f.optim_init()
    f.step(ydict(Y), 1, T)
    assert relerr(f._C[T], g["C_T"]) < 1e-9 and relerr(f._V[T], g["V_T"]) < 1e-9
    assert relerr(f._mu[T], g["mu_T"].reshape(-1, 1)) < 1e-9 and relerr(f._P[T], g["P_T"]) < 1e-9
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + 1)])
    assert relerr(yp, g["y_pred"]) < 1e-9


@pytest.mark.parametrize("storage,tol", [("f64", 1e-6), ("f32", 1e-5)])
@pytest.mark.parametrize("name,robust", [("psmf_simplified_cos", False), ("rpsmf_simplified_cos", True)])
def test_experiment_synthetic_on_device(name, robust, storage, tol):
    """ExperimentSynthetic: simplified hooks declared as hip_mode, cos dynamics on the device, Adam on the
    host between epochs; the reference's theta trajectory, predictions and error norms."""
    g = load_golden(name)
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])
    d, r = g["C0"].shape
    base = psmf.rPSMFIter if robust else psmf.PSMFIter

    class Synth(base):
        hip_mode = "simplified"

        def step_reset(self):
            super().step_reset()
            self._V = {0: self.V0}   # V is re-initialised every epoch (synthetic_psmf.py:78-81)

    y_obs = ydict(g["Y_obs"])
    y_train = {k: y_obs[k] for k in range(1, T + 1)}
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    nl = psmf.CosPhase(r)
    if robust:
        f = Synth(theta0, g["C0"], g["V0"], mu0, g["P0"], 0 * np.eye(r), np.eye(d), 1.8, nl, storage=storage)
    else:
        f = Synth(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: 0 * np.eye(r) for k in range(T + 1)},
                  {k: np.eye(d) for k in range(T + 1)}, nl, storage=storage)
    f.adam_init(gam=1e-3)
    for i in range(1, n_iter + 1):
        f.step(y_train, i, T)
        f.predict(i, T, n_pred)
        assert relerr(f._gradsum.reshape(-1), g["gradsum"][i - 1]) < 20 * tol
        f.adam_update(i)
    theta = np.array([f._theta[i].reshape(-1) for i in range(n_iter + 1)])
    assert relerr(theta, g["theta"]) < 10 * tol
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp, g["y_pred_last"]) < 10 * tol
    assert relerr(np.linalg.norm(yp - g["Y_obs"]), g["E_y"][-1]) < 10 * tol


@pytest.mark.parametrize("name,robust", [("psmf_recursive", False), ("rpsmf_recursive", True)])
def test_recursive_classes_on_device(name, robust):
    g = load_golden(name)
    T, n_pred, ue = int(g["T"]), int(g["n_pred"]), int(g["update_every"])
    d, r = g["C0"].shape
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    nl = psmf.CosPhase(r)
    if robust:
        f = psmf.rPSMFRecursive(theta0, g["C0"], g["V0"], mu0, g["P0"], g["Q"], np.eye(d), 1.8, nl, storage="f64")
    else:
        f = psmf.PSMFRecursive(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: g["Q"] for k in range(T + 1)},
                               {k: np.eye(d) for k in range(T + 1)}, nl, storage="f64")
    f.run(ydict(g["Y"]), T, n_pred, update_every=ue)
    assert relerr(f._theta[T].reshape(-1), g["theta"][-1]) < 1e-6
    assert relerr(f._C[T], g["C_T"]) < 1e-6
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
    assert relerr(yp, g["y_pred"]) < 1e-6


# ----------------------------------------------------------------------------- masked engine
def test_impute_reference_functions_synthetic():
    """Drop-in functions vs the reference functions' own outputs (d=19, n=400, r=10, 40 % missing)."""
    g = load_golden("impute_synth")
    Yint = np.nan_to_num(g["Yorig"], nan=0.0)
    d, n = Yint.shape
    r = g["C0"].shape[1]
    V, Q, P, R = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r), 10 * np.eye(d)
    X = g["X0"].copy()
    ep, ef, rt, ib = impute.ProbabilisticSequentialMatrixFactorizer(
        g["Y"], g["C0"].copy(), X, d, n, r, g["M"], g["Mmiss"], 10, V, Q, R, P, 2, 2, Yint, float(g["Einit"]))
    assert relerr(ep, g["psmf_Epred"]) < 1e-9 and relerr(ef, g["psmf_Efull"]) < 1e-9
    assert abs(ib - float(g["psmf_inside"])) < 1e-12
    assert relerr(X, g["psmf_X"]) < 1e-8
    X = g["X0"].copy()
    ep, ef, rt, ib = impute.robust_PSMF(
        g["Y"], g["C0"].copy(), X, d, n, r, g["M"], g["Mmiss"], V, Q, R, P, 1.8, 2, 2, Yint, float(g["Einit"]))
    assert relerr(ep, g["rpsmf_Epred"]) < 1e-9 and relerr(ef, g["rpsmf_Efull"]) < 1e-9
    assert abs(ib - float(g["rpsmf_inside"])) < 1e-12
    assert relerr(X, g["rpsmf_X"]) < 1e-8


def test_impute_baseline_filters_vs_reference_outputs():
    """MLE-SMF and TMF (the baseline filters that share the masked contractions) through the drop-in functions,
    against the reference functions' own outputs on the impute_synth inputs; MLE-SMF bands and final C against the oracle."""
    from oracle.impute_oracle import mle_smf_filter

    g, b = load_golden("impute_synth"), load_golden("impute_baselines")
    Yint = np.nan_to_num(g["Yorig"], nan=0.0)
    d, n = Yint.shape
    r = g["C0"].shape[1]
    Q, P, R = 0.1 * np.eye(r), np.eye(r), 10 * np.eye(d)
    X = g["X0"].copy()
    ep, ef, rt, ib = impute.stochasticGradientStateSpaceMF(
        g["Y"], g["C0"].copy(), X, d, n, r, g["M"], g["Mmiss"], 10, Q, R, P, 2, 2, Yint, float(g["Einit"]))
    assert relerr(ep, b["mle_Epred"]) < 1e-9 and relerr(ef, b["mle_Efull"]) < 1e-9
    assert abs(ib - float(b["mle_inside"])) < 1e-12
    assert relerr(X, b["mle_X"]) < 1e-8
    X = g["X0"].copy()
    ep, ef, rt = impute.temporalRegularizedMF(
        g["Y"], g["C0"].copy(), X, d, n, r, g["M"], g["Mmiss"], 10, R, 2, Yint, float(g["Einit"]))
    assert relerr(ep, b["tmf_Epred"]) < 1e-9 and relerr(ef, b["tmf_Efull"]) < 1e-9
    assert relerr(X, b["tmf_X"]) < 1e-8
    # batch of 2 identical replicas with bands: final C, bands vs the oracle
    Xo = g["X0"].copy()
    _, _, _, st = mle_smf_filter(g["Y"], g["C0"], Xo, g["M"], g["Mmiss"], Q, 10.0, P, 2, 2, Yint, float(g["Einit"]), return_state=True)
    res = impute.impute_batch(Yint, np.stack([g["M"]] * 2), np.stack([g["Mmiss"]] * 2), np.stack([g["C0"]] * 2), np.stack([g["X0"]] * 2),
                              np.eye(r), Q, 10.0, P, 2, 2, method="mle_smf", want_bands=True)
    for rep in range(2):
        assert relerr(res["C"][rep], st["C"]) < 1e-10
        assert relerr(res["YrecL"][rep], st["YrecL"]) < 1e-9 and relerr(res["YrecH"][rep], st["YrecH"]) < 1e-9


@pytest.mark.parametrize("method", ["PSMF", "rPSMF"])
def test_impute_stored_known_answers_batched(method):
    """The reference's stored results for LondonAir_PM25 (40 %, seed 123), first two repeats as ONE batch."""
    g = load_golden("impute_kat_pm25")
    Yorig = g["Yorig"]
    Yint = np.nan_to_num(Yorig, nan=0.0)
    prm = json.loads(str(g[f"{method}_40_params"]))
    d, n = Yorig.shape
    r = prm["r"]
    np.random.seed(123)
    pbs = [H.draw_problem(Yorig, 40, r) for _ in range(2)]
    res = impute.impute_batch(
        Yint, np.stack([p["M"] for p in pbs]), np.stack([p["Mmiss"] for p in pbs]), np.stack([p["C"] for p in pbs]),
        np.stack([p["X"] for p in pbs]), prm["v"] * np.eye(r), prm["q"] * np.eye(r), float(prm["rho"]),
        prm["p"] * np.eye(r), prm["sig"], prm["Iter"], robust=(method == "rPSMF"), lambda0=prm.get("lambda0", 1.8))
    for rep in range(2):
        assert relerr(res["Epred"][rep, -1], g[f"{method}_40_error_predict"][rep]) < 1e-8
        assert relerr(res["Efull"][rep, -1], g[f"{method}_40_error_full"][rep]) < 1e-8
        assert abs(res["inside"][rep] - g[f"{method}_40_inside_sig"][rep]) < 1e-12


@pytest.mark.parametrize("robust", [False, True])
def test_impute_bands_and_state_vs_oracle_ragged(robust):
    """d = 300 (two row rounds per thread), r = 7 (odd: identity-padded 2x2 pivots), bands returned."""
    rng = np.random.default_rng(4)
    d, n, r = 300, 60, 7
    Yorig = np.cumsum(0.3 * rng.standard_normal((d, n)), axis=1)
    M = (rng.random((d, n)) > 0.4).astype(int)
    M[5] = 0            # a row that is never observed
    M[:, 17] = 0        # a column with no observation at all
    Mmiss = ((1 - M) * (rng.random((d, n)) > 0.2)).astype(float)
    C0, X0 = rng.random((d, r)), rng.random((r, n))
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    Xo = X0.copy()
    ep, ef, ib, st = impute_filter(Yorig * M, C0, Xo, M, Mmiss, V, Q, 10.0, P, 2, 2, Yorig, 0.0, robust=robust,
                                   lambda0=1.8, return_state=True)
    res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8,
                              want_bands=True)
    assert relerr(res["Epred"][0], ep[0, 1:]) < 1e-9 and relerr(res["Efull"][0], ef[0, 1:]) < 1e-9
    assert abs(res["inside"][0] - ib) < 1e-12
    assert relerr(res["C"][0], st["C"]) < 1e-8 and relerr(res["X"][0], st["X"]) < 1e-8
    for k in ("Yrec", "YrecL", "YrecH"):
        assert relerr(res[k][0], st[k]) < 1e-8


def test_impute_argument_errors():
    with pytest.raises(ValueError):
        impute.impute_batch(np.zeros((3, 10)), np.ones((3, 10)), np.ones((3, 10)), np.ones((3, 65)), np.ones((65, 10)),
                            np.eye(65), np.eye(65), 1.0, np.eye(65), 2, 1)   # r > PSMF_RMAX (r = 17 .. 64: the masked per-step engine)
