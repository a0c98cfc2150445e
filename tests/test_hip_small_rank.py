"""psmf_blk_filter6 (psmf_blk16.hip), the block filter of ranks r <= 16: ranks 1 ... 16 (even / odd: the identity-padded pivot,
the augmented column at r2; 15, 16: no column left for it), every dynamics kind and flag set incl. > 4 terms (the generic term loops), the hook configurations,
rPSMF, a general Q, R_k / Q_k schedules with a dense Jacobian, two blocks and a ragged last block -- against the oracle on the
same callables, and against the general one-group kernel (PSMF_FILTER6=0) on the same inputs.  GPU only: `pytest -m gpu`."""

import os

import numpy as np
import pytest

from conftest import relerr
from oracle import psmf_oracle as O
from rpsmf_amd import nonlinearities as NL

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _capi():
    from rpsmf_amd import _capi

    return _capi


def _theta_for(nl, rng, r):
    th = 0.1 * rng.random(nl.n_params)
    if isinstance(nl, NL.ScaledWalk) or getattr(nl, "scaled", False):
        th[:r * r] = (0.8 * np.eye(r) + 0.05 * rng.standard_normal((r, r))).reshape(-1)
    if isinstance(nl, NL.FourierBasis):
        for t in range(2 * nl.N):
            th[t * r * r:(t + 1) * r * r] = (0.5 * np.eye(r) + 0.05 * rng.standard_normal((r, r))).reshape(-1) / nl.N
    return th


def _device(c, nl, d, r, Y, C0, V0, P0, Q, mu0, theta, T, robust, mode_kw, sched=None, want="psmf_blk_filter6"):
    f = c.DeviceFilter(d, r, robust=robust, storage="f64", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags,
                       dyn_terms=nl.device_terms, **mode_kw)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, mu0, rho=1.0, lambda0=1.8, theta=theta if nl.n_params else None)
    if sched:
        f.set_schedules(*sched)
    kern = f.geometry()["filter_kernel"]
    # (psmf_blk_filter6d = the same programs with the two inversions side by side: random walk with Q = q I -- at r = 1 every Q is)
    assert kern == want or (want == "psmf_blk_filter6" and kern == "psmf_blk_filter6d" and nl.n_params == 0 and r == 1), f.geometry()
    if nl.n_params:
        f.zero_gradsum()
    f.run(0, T // 2)
    f.run(T // 2, T)
    s, yp = f.get_state(), f.y_pred(0, T)
    f.close()
    return s, yp


KINDS = [
    ("random_walk", lambda r: NL.RandomWalk()),
    ("scaled_walk_bias", lambda r: NL.ScaledWalk(r, bias=True)),
    ("sinusoid", lambda r: NL.Sinusoid(r)),
    ("sinusoid_unphased", lambda r: NL.Sinusoid(r, phased=False)),
    ("sinusoid_unscaled", lambda r: NL.Sinusoid(r, scaled=False)),
    ("cos_phase", lambda r: NL.CosPhase(r)),
    ("fourier1", lambda r: NL.FourierBasis(r, N=1)),
    ("fourier2", lambda r: NL.FourierBasis(r, N=2)),
    ("fourier3", lambda r: NL.FourierBasis(r, N=3)),
]
FULL = dict(coef_update=True, eta_full=True, pbar_predict=True)
MODES = {"full": FULL, "no_update": dict(coef_update=False, eta_full=True, pbar_predict=True),
         "eta_R": dict(coef_update=True, eta_full=False, pbar_predict=True), "pbar_P": dict(coef_update=True, eta_full=True, pbar_predict=False)}


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("r", [1, 2, 5, 10, 13, 14, 15, 16])
@pytest.mark.parametrize("name,make", KINDS, ids=[k[0] for k in KINDS])
def test_small_rank_kinds_vs_oracle(name, make, r, robust):
    c = _capi()
    d, T = 260, 70            # two blocks of 48: the second one ragged; and two runs (the state carried between launches)
    nl = make(r)
    rng = np.random.default_rng(100 * r + len(name))
    Y = O.synthetic_series(d, r, T, 7 + r, noise="t" if robust else "normal", dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    theta = _theta_for(nl, rng, r) if nl.n_params else np.zeros(0)
    V0, P0 = 0.1 * np.eye(r), np.eye(r)
    B = rng.standard_normal((r, r))
    general_Q = name in ("sinusoid", "random_walk")          # Q not a multiple of I
    Q = 0.1 * np.eye(r) + (0.01 * B @ B.T if general_Q else 0.0)
    mu0 = 0.2 * rng.standard_normal(r)
    dyn = O.CallableDyn(nl, nl.n_params) if nl.n_params else O.RandomWalkDyn()
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=robust), dyn, want_grad=bool(nl.n_params))
    want = "psmf_blk_filter6"
    s, yp = _device(c, nl, d, r, Y, C0, V0, P0, Q, mu0, theta, T, robust, FULL, want=want)
    tol = 1e-7 if robust else 1e-8
    for k in ("C", "V", "mu", "P"):
        assert relerr(s[k], getattr(st, k)) < tol, k
    assert relerr(yp, Yp) < tol
    if nl.n_params:
        assert relerr(s["gradsum"], st.gradsum) < 10 * tol


@pytest.mark.parametrize("mode", ["no_update", "eta_R", "pbar_P"])
@pytest.mark.parametrize("name,make", [KINDS[1], KINDS[7], KINDS[5]], ids=["scaled_walk_bias", "fourier2", "cos_phase"])
def test_small_rank_hook_configurations(name, make, mode):
    """The reference's hook overrides one at a time (no coefficient update / eta = tr R / d / P_bar = P), r = 9 (odd)."""
    c = _capi()
    d, r, T = 260, 9, 60
    nl = make(r)
    rng = np.random.default_rng(3)
    Y = O.synthetic_series(d, r, T, 17, dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    theta = _theta_for(nl, rng, r)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mu0 = 0.2 * rng.standard_normal(r)
    kw = MODES[mode]
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(**kw), O.CallableDyn(nl, nl.n_params))
    s, yp = _device(c, nl, d, r, Y, C0, V0, P0, Q, mu0, theta, T, False, kw)
    for k in ("C", "V", "mu", "P"):
        assert relerr(s[k], getattr(st, k)) < 1e-8, k
    assert relerr(yp, Yp) < 1e-8 and relerr(s["gradsum"], st.gradsum) < 1e-7


def test_small_rank_schedules_with_a_dense_jacobian_and_vs_general_kernel():
    """R_k, Q_k schedules (psmf.py:115,123,141) with FourierBasis dynamics, against the oracle; and the same run on the general
    one-group kernel (PSMF_FILTER6=0): the two kernels agree to round-off."""
    c = _capi()
    d, r, T = 300, 10, 96
    nl = NL.FourierBasis(r, N=2)
    rng = np.random.default_rng(5)
    Y = O.synthetic_series(d, r, T, 19, dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    theta = _theta_for(nl, rng, r)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mu0 = 0.2 * rng.standard_normal(r)
    rho_k, q_k = 0.5 + rng.random(T + 1), 0.5 + rng.random(T + 1)
    q_k[1] = 1.0
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=0.0, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(), O.CallableDyn(nl, nl.n_params), Qs=lambda k: q_k[k] * Q, rhos=lambda k: rho_k[k])
    s, yp = _device(c, nl, d, r, Y, C0, V0, P0, Q, mu0, theta, T, False, FULL, sched=(rho_k, q_k))
    for k in ("C", "V", "mu", "P"):
        assert relerr(s[k], getattr(st, k)) < 1e-8, k
    assert relerr(yp, Yp) < 1e-8 and relerr(s["gradsum"], st.gradsum) < 1e-7
    os.environ["PSMF_FILTER6"] = "0"
    try:
        s1, yp1 = _device(c, nl, d, r, Y, C0, V0, P0, Q, mu0, theta, T, False, FULL, sched=(rho_k, q_k), want="psmf_blk_filter")
    finally:
        os.environ.pop("PSMF_FILTER6", None)
    for k in ("C", "V", "mu", "P", "gradsum"):
        assert relerr(s[k], s1[k]) < 1e-9, k
    assert relerr(yp, yp1) < 1e-9


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("r", [1, 2, 7, 12, 14, 15, 16])
def test_small_rank_random_walk_inversions_side_by_side(r, robust):
    """The default model (random walk, Q = q I) at r <= 14 on psmf_blk_filter6d with W_k = (M_k / beta + I / q_k)^-1 formed beside
    P+_k = M_k^-1 (one sweep on the path of a step instead of two); rPSMF: q, rho, lambda run with omega.  And the same run with
    PSMF_FILTER6_DUAL=0 (filter3s): the two kernels agree."""
    c = _capi()
    d, T = 260, 130           # three blocks, the last one ragged; two runs
    nl = NL.RandomWalk()
    rng = np.random.default_rng(40 + r)
    Y = O.synthetic_series(d, r, T, 9 + r, noise="t" if robust else "normal", dtype=np.float64)
    C0 = 0.1 * rng.standard_normal((d, r))
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mu0 = 0.2 * rng.standard_normal(r)
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8)
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=robust), O.RandomWalkDyn(), want_grad=False)
    s, yp = _device(c, nl, d, r, Y, C0, V0, P0, Q, mu0, np.zeros(0), T, robust, FULL, want="psmf_blk_filter6d")
    tol = 1e-7 if robust else 1e-9
    for k in ("C", "V", "mu", "P"):
        assert relerr(s[k], getattr(st, k)) < tol, k
    assert relerr(yp, Yp) < tol and relerr(s["Q"], st.Q) < tol
    os.environ["PSMF_FILTER6_DUAL"] = "0"
    try:
        s3, yp3 = _device(c, nl, d, r, Y, C0, V0, P0, Q, mu0, np.zeros(0), T, robust, FULL, want="psmf_blk_filter3s")
    finally:
        os.environ.pop("PSMF_FILTER6_DUAL", None)
    for k in ("C", "V", "mu", "P"):
        assert relerr(s[k], s3[k]) < 10 * tol, k
