"""The small-shape masked column loop (psmf_impute_kernel3: d <= 80, r <= 14 -- the shapes of ExperimentImpute) against the
CPU oracle and against version 2 of the loop (PSMF_IMPUTE_V3=0) on the same inputs: every method (PSMF, rPSMF, MLE-SMF, TMF),
even and odd ranks, every LDS row-group instantiation (d <= 12, <= 20, <= 32, <= 48, <= 80), Q = q I (two inversions side by side) and a
general Q (two sweeps in turn), rows / columns without observations, bands.  GPU only: `pytest -m gpu`."""

import os

import numpy as np
import pytest

from conftest import relerr
from oracle.impute_oracle import impute_filter, mle_smf_filter, tmf_filter
from rpsmf_amd import impute

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _problem(d, n, r, seed, empty_column=True):
    rng = np.random.default_rng(seed)
    Yorig = np.cumsum(0.3 * rng.standard_normal((d, n)), axis=1)
    M = (rng.random((d, n)) > 0.4).astype(int)
    if d > 3:
        M[2] = 0                     # a row that is never observed
    for t in np.flatnonzero(M.sum(axis=0) == 0):
        M[(0 if d <= 3 else 3), t] = 1
    if empty_column and d >= 12:
        M[:, min(17, n - 1)] = 0     # a column with no observation at all: eta = 0 there and V loses a direction (with d of
                                     # a few rows the reference's own N = s + eta then rounds to either side of zero; its
                                     # MLE-SMF divides by that eta, MLESMF.py:79)
    Mmiss = ((1 - M) * (rng.random((d, n)) > 0.2)).astype(float)
    return Yorig, M, Mmiss, rng.random((d, r)), rng.random((r, n))


def _err(a, b):
    """relerr that accepts the reference's own NaN bands (sqrt of an N = s + eta that a column with no observation left at
    zero and rounding took below it, PSMF.py:83-84): NaN in the same places, the rest compared."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    return relerr(np.nan_to_num(a), np.nan_to_num(b))


def _with_env(name, value, fn):
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def _v2(fn):
    return _with_env("PSMF_IMPUTE_V3", "0", fn)


def _sequential(fn):
    """PSMF_IMPUTE_PAR=0: the two r x r inversions of a column one after the other on one wave (the Q = q I case takes them side by
    side on two by default)"""
    return _with_env("PSMF_IMPUTE_PAR", "0", fn)


SHAPES = [(19, 10), (19, 9), (7, 3), (12, 5), (16, 14), (20, 13), (21, 2), (32, 14), (25, 7), (2, 1),
          (33, 5), (40, 7), (48, 14), (64, 14), (65, 3), (75, 10), (80, 13)]       # rows on one lane each; beyond 64 on a second wave


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("d,r", SHAPES)
def test_small_shapes_vs_oracle_and_version_2(d, r, robust):
    n = 90
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 100 * d + r)
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    ep, ef, ib, st = impute_filter(Yorig * M, C0, X0.copy(), M, Mmiss, V, Q, 10.0, P, 2, 2, Yorig, 0.0, robust=robust,
                                   lambda0=1.8, return_state=True)
    run = lambda: impute.impute_batch(Yorig, np.stack([M] * 3), np.stack([Mmiss] * 3), np.stack([C0] * 3), np.stack([X0] * 3),
                                      V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8, want_bands=True)
    res, res2 = run(), _v2(run)
    res3 = _sequential(run) if (d, r) in ((19, 10), (7, 3), (33, 5), (75, 10)) else None
    tol = 1e-8 if robust else 1e-10
    for rep in (0, 2):
        assert relerr(res["Epred"][rep], ep[0, 1:]) < 1e-9 and relerr(res["Efull"][rep], ef[0, 1:]) < 1e-9
        assert abs(res["inside"][rep] - ib) < 1e-12
        assert relerr(res["C"][rep], st["C"]) < tol and relerr(res["X"][rep], st["X"]) < tol
        for k in ("Yrec", "YrecL", "YrecH"):
            assert _err(res[k][rep], st[k]) < tol, k
    assert np.array_equal(res["X"][0], res["X"][2])            # replicas of one problem: the same bits
    for k in ("C", "X", "Yrec", "YrecL", "YrecH"):
        assert _err(res[k], res2[k]) < tol, k
        if res3 is not None:
            assert _err(res[k], res3[k]) < tol, ("PSMF_IMPUTE_PAR=0", k)


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_small_shape_general_Q(robust):
    """Q not a multiple of the identity: wave 0 inverts P + Q and (P + Q)^-1 + kappa G in turn (PSMF.py:30-36 as written)."""
    d, n, r = 19, 120, 10
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 5)
    rng = np.random.default_rng(6)
    B = rng.standard_normal((r, r))
    V, Q, P = 2 * np.eye(r), 0.05 * np.eye(r) + 0.01 * B @ B.T, np.eye(r)
    ep, ef, ib, st = impute_filter(Yorig * M, C0, X0.copy(), M, Mmiss, V, Q, 10.0, P, 2, 2, Yorig, 0.0, robust=robust,
                                   lambda0=1.8, return_state=True)
    res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8, want_bands=True)
    tol = 1e-8 if robust else 1e-10
    assert relerr(res["Epred"][0], ep[0, 1:]) < 1e-9 and abs(res["inside"][0] - ib) < 1e-12
    assert relerr(res["C"][0], st["C"]) < tol and relerr(res["X"][0], st["X"]) < tol
    assert relerr(res["YrecL"][0], st["YrecL"]) < tol


@pytest.mark.parametrize("d,r", [(19, 10), (9, 3), (30, 13), (75, 10)])
def test_small_shape_baseline_filters(d, r):
    """MLE-SMF and TMF (gradient step on C, no V) in the small-shape loop."""
    n = 150
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 40 + d, empty_column=False)
    Q, P = 0.1 * np.eye(r), np.eye(r)
    _, _, ib, st = mle_smf_filter(Yorig * M, C0, X0.copy(), M, Mmiss, Q, 10.0, P, 2, 2, Yorig, 0.0, return_state=True)
    res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, np.eye(r), Q, 10.0, P, 2, 2, method="mle_smf", want_bands=True)
    assert relerr(res["C"][0], st["C"]) < 1e-10 and relerr(res["X"][0], st["X"]) < 1e-9
    assert relerr(res["YrecL"][0], st["YrecL"]) < 1e-9 and abs(res["inside"][0] - ib) < 1e-12
    ep, ef, st = tmf_filter(Yorig * M, C0, X0.copy(), M, Mmiss, 2, Yorig, 0.0, return_state=True)
    res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, np.eye(r), Q, 10.0, P, 2, 2, method="tmf")
    assert relerr(res["C"][0], st["C"]) < 1e-10 and relerr(res["X"][0], st["X"]) < 1e-9
    assert relerr(res["Epred"][0], ep[0, 1:]) < 1e-9 and relerr(res["Efull"][0], ef[0, 1:]) < 1e-9


@pytest.mark.parametrize("seed", range(24))
def test_small_shapes_drawn_at_random(seed):
    """Shapes, ranks, methods and horizons drawn at random (fixed seeds) over the whole range of the small-shape loop -- d = 1 ... 80,
    r = 1 ... 14, r > d included -- against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    d, r, n = int(rng.integers(1, 81)), int(rng.integers(1, 15)), int(rng.integers(20, 120))
    method = ["psmf", "rpsmf", "mle_smf", "tmf"][int(rng.integers(0, 4))]
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 7000 + seed, empty_column=False)
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    if method in ("psmf", "rpsmf"):
        robust = method == "rpsmf"
        ep, ef, ib, st = impute_filter(Yorig * M, C0, X0.copy(), M, Mmiss, V, Q, 10.0, P, 2, 2, Yorig, 0.0, robust=robust, lambda0=1.8,
                                       return_state=True)
        res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8, want_bands=True)
        tol = 1e-7 if (robust or r >= d) else 1e-9
        assert _err(res["YrecL"][0], st["YrecL"]) < tol and abs(res["inside"][0] - ib) < 1e-12
    elif method == "mle_smf":
        ep, ef, ib, st = mle_smf_filter(Yorig * M, C0, X0.copy(), M, Mmiss, Q, 10.0, P, 2, 2, Yorig, 0.0, return_state=True)
        res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, np.eye(r), Q, 10.0, P, 2, 2, method="mle_smf", want_bands=True)
        tol = 1e-8
    else:
        ep, ef, st = tmf_filter(Yorig * M, C0, X0.copy(), M, Mmiss, 2, Yorig, 0.0, return_state=True)
        res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, np.eye(r), Q, 10.0, P, 2, 2, method="tmf")
        tol = 1e-8
    assert _err(res["C"][0], st["C"]) < tol and _err(res["X"][0], st["X"]) < tol, (d, r, n, method)
    assert _err(res["Epred"][0], ep[0, 1:]) < 1e-8 and _err(res["Efull"][0], ef[0, 1:]) < 1e-8


def test_replicas_dealt_over_a_device_list_are_the_same_filters():
    """impute_batch(device=[...]): contiguous slices of the replicas on the listed devices, one host thread each (config
    D's seeds over the GPUs of a node).  On a one-GPU box the list names device 0 three times: same bits as the single call."""
    d, n, r, B = 19, 120, 10, 7
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 321)
    rng = np.random.default_rng(5)
    Ms = np.stack([(rng.random((d, n)) > 0.3).astype(float) for _ in range(B)])
    Mm = 1.0 - Ms
    Cs = np.stack([rng.random((d, r)) for _ in range(B)])
    Xs = np.stack([rng.random((r, n)) for _ in range(B)])
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    one = impute.impute_batch(Yorig, Ms, Mm, Cs, Xs, V, Q, 10.0, P, 2, 2, robust=True, lambda0=1.8, want_bands=True)
    three = impute.impute_batch(Yorig, Ms, Mm, Cs, Xs, V, Q, 10.0, P, 2, 2, robust=True, lambda0=1.8, want_bands=True, device=[0, 0, 0])
    assert three["devices"] == [(0, 0, 3), (0, 3, 5), (0, 5, 7)]
    for k in ("Epred", "Efull", "inside", "C", "X", "Yrec", "YrecL", "YrecH", "status"):
        assert np.array_equal(one[k], three[k], equal_nan=True), k
