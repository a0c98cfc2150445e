"""Known-answer tests the reference itself stores for the masked (ExperimentImpute) path,
replayed through the host harness + the CPU oracle.  CPU only.

Fixtures: tests/golden/impute_kat_{pm25,pm10,sp500}.npz = the reference's LondonAir_PM25.csv, LondonAir_PM10.csv,
sp500_closing_prices.csv plus, for 20/30/40 % missing x {PSMF, rPSMF, MLESMF, TMF}, ALL 100 repeats of `hashes` and
`results` from ExperimentImpute/output/<dataset>_<pct>_<method>.json (seed 123, Makefile:176).  The oracle replays repeat 0 of
every (dataset, method, percentage) and one deeper repeat per data set here (about a second each); the GPU replays all 100
repeats of all 36 files in tests/test_hip_kat_all.py."""

import json

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle.impute_oracle import impute_filter
from rpsmf_amd import impute_harness as H


def _replay(g, pct, method, n_rep):
    Yorig = g["Yorig"]
    YorigInt = np.nan_to_num(Yorig, nan=0.0)
    prm = json.loads(str(g[f"{method}_{pct}_params"]))
    d, n = Yorig.shape
    r = prm["r"]
    np.random.seed(123)
    for rep in range(n_rep):
        pb = H.draw_problem(Yorig, pct, r)
        yield rep, pb, prm, YorigInt, d, n, r


@pytest.mark.parametrize("pct", [20, 30, 40])
def test_rng_plumbing_hashes(pct):
    """seed 123 -> prepare_missing -> rand(d,r) -> rand(r,n) reproduces the stored input hashes."""
    g = load_golden("impute_kat_pm25")
    for rep, pb, prm, *_ in _replay(g, pct, "PSMF", 2):
        assert H.matrix_hash(pb["Y"]) == str(g[f"PSMF_{pct}_hash_Y"][rep])
        assert H.matrix_hash(pb["C"]) == str(g[f"PSMF_{pct}_hash_C"][rep])
        assert H.matrix_hash(pb["X"]) == str(g[f"PSMF_{pct}_hash_X"][rep])
    # the JSON keeps the ratio of the LAST of its 100 repeats; repeats differ by < 1e-3
    assert abs(pb["ratio"] - float(g[f"PSMF_{pct}_missing_ratio"])) < 2e-3


@pytest.mark.parametrize("method", ["PSMF", "rPSMF"])
@pytest.mark.parametrize("pct", [20, 40])
def test_stored_results(pct, method):
    """error_predict / error_full / inside_sig of the stored JSON, repeat 0 (2 passes, 8786 steps)."""
    g = load_golden("impute_kat_pm25")
    for rep, pb, prm, YorigInt, d, n, r in _replay(g, pct, method, 1):
        V, Q, P = prm["v"] * np.eye(r), prm["q"] * np.eye(r), prm["p"] * np.eye(r)
        Einit = H.RMSEM(pb["C"] @ pb["X"], YorigInt, pb["Mmiss"])
        ep, ef, ib = impute_filter(pb["Y"], pb["C"], pb["X"], pb["M"], pb["Mmiss"], V, Q, float(prm["rho"]),
                                   P, prm["sig"], prm["Iter"], YorigInt, Einit,
                                   robust=(method == "rPSMF"), lambda0=prm.get("lambda0", 1.8))
        key = f"{method}_{pct}_"
        assert relerr(ep[0, -1], g[key + "error_predict"][rep]) < 1e-9
        assert relerr(ef[0, -1], g[key + "error_full"][rep]) < 1e-9
        assert abs(ib - g[key + "inside_sig"][rep]) < 1e-12


def test_reference_functions_on_synthetic():
    """Outputs of the reference's two functions run in the build container (d=19, n=400, r=10, 40 %)."""
    g = load_golden("impute_synth")
    Yint = np.nan_to_num(g["Yorig"], nan=0.0)
    r = g["C0"].shape[1]
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    for robust, key in ((False, "psmf"), (True, "rpsmf")):
        X = g["X0"].copy()
        ep, ef, ib = impute_filter(g["Y"], g["C0"], X, g["M"], g["Mmiss"], V, Q, 10.0, P, 2, 2, Yint,
                                   float(g["Einit"]), robust=robust, lambda0=1.8)
        assert relerr(ep, g[key + "_Epred"]) < 1e-10
        assert relerr(ef, g[key + "_Efull"]) < 1e-10
        assert abs(ib - float(g[key + "_inside"])) < 1e-12
        assert relerr(X, g[key + "_X"]) < 1e-9


def test_baseline_filters_on_synthetic():
    """MLE-SMF (MLESMF.py:40-92) and TMF (TMF.py:30-73) restatements vs the reference functions run in the build
    container on the impute_synth inputs (fixture tests/golden/impute_baselines.npz, make_golden.py)."""
    from oracle.impute_oracle import mle_smf_filter, tmf_filter

    g, b = load_golden("impute_synth"), load_golden("impute_baselines")
    Yint = np.nan_to_num(g["Yorig"], nan=0.0)
    r = g["C0"].shape[1]
    X = g["X0"].copy()
    ep, ef, ib = mle_smf_filter(g["Y"], g["C0"], X, g["M"], g["Mmiss"], 0.1 * np.eye(r), 10.0, np.eye(r), 2, 2, Yint, float(g["Einit"]))
    assert relerr(ep, b["mle_Epred"]) < 1e-12 and relerr(ef, b["mle_Efull"]) < 1e-12
    assert abs(ib - float(b["mle_inside"])) < 1e-12
    assert relerr(X, b["mle_X"]) < 1e-12
    X = g["X0"].copy()
    ep, ef = tmf_filter(g["Y"], g["C0"], X, g["M"], g["Mmiss"], 2, Yint, float(g["Einit"]))
    assert relerr(ep, b["tmf_Epred"]) < 1e-12 and relerr(ef, b["tmf_Efull"]) < 1e-12
    assert relerr(X, b["tmf_X"]) < 1e-12


# ------------------------------------------------------------------------------------------------ all three data sets, four methods
import kat_replay as K  # noqa: E402


@pytest.mark.parametrize("ds,n_rep", [("pm25", 100), ("pm10", 100), ("sp500", 12)])
@pytest.mark.parametrize("pct", [20, 30, 40])
def test_rng_plumbing_hashes_all_datasets(ds, pct, n_rep):
    """Input hashes of the stored repeats (all 100 for the LondonAir sets; the first 12 of the S&P 500 set, whose draw costs
    0.15 s per repeat on the build container): the masks, C0 and X0 every method's known answers were computed from."""
    g = K.fixture(ds)
    for pb in K.draws(ds, pct, n_rep):
        rep = pb["rep"]
        assert pb["hY"] == str(g[f"PSMF_{pct}_hash_Y"][rep]), (ds, pct, rep)
        assert pb["hC"] == str(g[f"PSMF_{pct}_hash_C"][rep]) and pb["hX"] == str(g[f"PSMF_{pct}_hash_X"][rep])


def _oracle_run(g, method, pct, pb):
    from oracle.impute_oracle import mle_smf_filter, tmf_filter

    Yint = np.nan_to_num(g["Yorig"], nan=0.0)
    prm = K.params(g, method, pct)
    r = prm["r"]
    M, Mmiss = pb["M"].astype(float), pb["Mmiss"].astype(float)
    Y = Yint * M
    X = pb["X"].copy()
    if method in ("PSMF", "rPSMF"):
        ep, ef, ib = impute_filter(Y, pb["C"], X, M, Mmiss, prm["v"] * np.eye(r), prm["q"] * np.eye(r), float(prm["rho"]), prm["p"] * np.eye(r),
                                   prm["sig"], prm["Iter"], Yint, 0.0, robust=(method == "rPSMF"), lambda0=prm.get("lambda0", 1.8))
    elif method == "MLESMF":
        ep, ef, ib = mle_smf_filter(Y, pb["C"], X, M, Mmiss, prm["q"] * np.eye(r), float(prm["rho"]), prm["p"] * np.eye(r), prm["sig"], prm["Iter"], Yint, 0.0)
    else:
        ep, ef = tmf_filter(Y, pb["C"], X, M, Mmiss, prm["Iter"], Yint, 0.0)
        ib = None
    return ep[0, -1], ef[0, -1], ib


@pytest.mark.parametrize("method", K.METHODS)
@pytest.mark.parametrize("ds", ["pm25", "pm10", "sp500"])
def test_stored_results_all_datasets_methods(ds, method):
    """error_predict / error_full / inside_sig of repeat 0 at 20, 30, 40 % for every data set and method the reference stores
    answers for (d = 27, 75, 505); repeat 7 at 40 % as well for PSMF / rPSMF (the RNG carried over seven repeats)."""
    g = K.fixture(ds)
    for pct in (20, 30, 40):
        want = {0, 7} if (pct == 40 and method in ("PSMF", "rPSMF")) else {0}
        for pb in K.draws(ds, pct, max(want) + 1, want=want):
            ep, ef, ib = _oracle_run(g, method, pct, pb)
            key, rep = f"{method}_{pct}_", pb["rep"]
            if method == "MLESMF":
                # The reference's own MLESMF.py, run on these very inputs, does not reproduce its stored JSONs: 1e-5 relative on the
                # LondonAir sets, 1 % on S&P 500 at 20 / 30 % -- and exactly at 40 % (other versions of the script wrote most of
                # them; make_golden.py:case_impute_kat_mlesmf_refrun).  Pinned instead to the reference FUNCTION's outputs on the
                # same inputs; the JSONs are a sanity bound only.
                ref = load_golden("impute_kat_mlesmf_refrun")[f"{ds}_{pct}"][rep]
                assert relerr(ep, ref[0]) < K.TOL[ds] and relerr(ef, ref[1]) < K.TOL[ds] and abs(ib - ref[2]) < 1e-9, (ds, pct, rep)
                assert relerr(ep, g[key + "error_predict"][rep]) < 2e-2 and relerr(ef, g[key + "error_full"][rep]) < 2e-2
                assert abs(ib - g[key + "inside_sig"][rep]) < 1e-2
                continue
            # (tolerances per data set: tests/kat_replay.py -- the S&P 500 recursion is badly conditioned on some draws)
            tol = K.TOL[ds]
            assert relerr(ep, g[key + "error_predict"][rep]) < tol, (ds, method, pct, rep)
            assert relerr(ef, g[key + "error_full"][rep]) < tol, (ds, method, pct, rep)
            if ib is not None:
                assert abs(ib - g[key + "inside_sig"][rep]) < K.TOL_INSIDE[ds]
