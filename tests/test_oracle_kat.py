"""Known-answer tests the reference itself stores for the masked (ExperimentImpute) path,
replayed through the host harness + the CPU oracle.  CPU only.

Fixture: tests/golden/impute_kat_pm25.npz = the reference's LondonAir_PM25.csv plus, for
20/30/40 % missing x {PSMF, rPSMF}, the first repeats of `hashes` and `results` from
ExperimentImpute/output/LondonAir_PM25_<pct>_<method>.json (seed 123, Makefile:176)."""

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
