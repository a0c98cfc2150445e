"""Ranks 33 ... 64 (the reference has no rank limit, pypsmf/psmf/psmf.py:32; no experiment of it goes beyond r = 10): the per-step
engine with the r x r inversions as wave-local tile sweeps on the matrix cores -- 3 x 3 tiles of 16 x 16 up to r = 48, 4 x 4 beyond
(psmf_wave16.hip: solve_block_wave_big_t), side by side for the random walk with Q = q I, one after the other otherwise -- and the
512-worker serial stage (psmf_serial_wide).  Against the float64 oracle, float64 storage: tile boundaries (33, 48, 49, 64), odd ranks
(identity padding to an even size), PSMF and rPSMF (beta, omega enter the second inversion), a general Q (no dual form), a
non-uniform diagonal R (weighted Gram, no dual form), a second run on a carried state (the dual form's carried Lbar), and the masked
filter.  GPU only: `pytest -m gpu`."""

import numpy as np
import pytest

from conftest import relerr
from oracle import psmf_oracle as O
from oracle.impute_oracle import impute_filter
from rpsmf_amd import impute

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _capi():
    from rpsmf_amd import _capi

    return _capi


def _problem(d, r, T, seed, heavy=False):
    rng = np.random.default_rng(seed)
    Ct = rng.standard_normal((d, r))
    x = rng.standard_normal(r)
    Y = np.empty((T, d))
    for t in range(T):
        x = x + 0.1 * rng.standard_normal(r)
        Y[t] = Ct @ x + 0.3 * (rng.standard_t(3.0, d) if heavy else rng.standard_normal(d))
    return Y, 0.1 * rng.standard_normal((d, r))


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("r", [33, 40, 48, 49, 63, 64])
def test_big_rank_random_walk_vs_oracle(r, robust):
    c = _capi()
    d, T = 900 + r, 50
    Y, C0 = _problem(d, r, T, 300 + r, robust)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    f = c.DeviceFilter(d, r, storage="f64", robust=robust)
    assert f.geometry()["engine"] == "step"
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    # two runs: the second starts on the carried state (a run's first step derives Lbar itself: three sweeps on one wave)
    for a, b in ((0, 20), (20, T)):
        st, Yp, _ = O.run_epoch(st, Y[a:b], O.Mode(robust=robust), O.RandomWalkDyn(), k0=a, want_grad=False)
        f.run(a, b)
        s = f.get_state()
        for n in ("C", "V", "mu", "P"):
            assert relerr(s[n], getattr(st, n)) < 1e-9, (r, n, b)
        assert relerr(f.y_pred(a, b - a), Yp) < 1e-9
        if robust:
            assert relerr(s["rho"], st.rho) < 1e-9 and relerr(s["Q"], st.Q) < 1e-9
    f.close()


@pytest.mark.parametrize("case", ["general_Q", "nonuniform_R"])
def test_big_rank_without_the_dual_form(case):
    """A Q that is not a multiple of the identity, a non-uniform diagonal R: the two inversions run one after the other on one wave."""
    c = _capi()
    d, r, T = 700, 40, 30
    Y, C0 = _problem(d, r, T, 77)
    rng = np.random.default_rng(5)
    V0, P0 = 0.1 * np.eye(r), np.eye(r)
    A = rng.standard_normal((r, r)) / np.sqrt(r)
    Q = 0.1 * np.eye(r) + (0.05 * (A @ A.T) if case == "general_Q" else 0.0)
    rho = 0.3 + 2.0 * rng.random(d) if case == "nonuniform_R" else 1.0
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=rho, lam=1.8)
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(), O.RandomWalkDyn(), want_grad=False)
    f = c.DeviceFilter(d, r, storage="f64", nonuniform_R=(case == "nonuniform_R"))
    if case == "nonuniform_R":
        f.set_row_noise(rho)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    f.run(0, T)
    s = f.get_state()
    for n in ("C", "V", "mu", "P"):
        assert relerr(s[n], getattr(st, n)) < 1e-9, (case, n)
    assert relerr(f.y_pred(0, T), Yp) < 1e-9
    f.close()


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_big_rank_masked_filter(robust):
    """ExperimentImpute's masked filter at r = 40 (PSMF.py:40-95, rPSMF.py:40-148 take any r): the masked per-step engine with the
    3 x 3 tile solve, the step's masked Gram read at its source."""
    d, n, r = 300, 70, 40
    rng = np.random.default_rng(11)
    Yorig = np.cumsum(0.3 * rng.standard_normal((d, n)), axis=1) + 3.0 * rng.random((d, 1))
    M = (rng.random((d, n)) > 0.4).astype(int)
    M[3] = 0
    Mmiss = ((1 - M) * (rng.random((d, n)) > 0.1)).astype(float)
    C0, X0 = rng.random((d, r)), rng.random((r, n))
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    Xo = X0.copy()
    ep, ef, ib, st = impute_filter(Yorig * M, C0, Xo, M, Mmiss, V, Q, 10.0, P, 2, 2, Yorig, 0.0, robust=robust, lambda0=1.8, return_state=True)
    res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8, want_bands=True)
    assert res["kernel"] == "masked per-step engine" and res["status"][0] == 0
    errs = dict(Epred=relerr(res["Epred"][0], ep[0, 1:]), Efull=relerr(res["Efull"][0], ef[0, 1:]), inside=abs(res["inside"][0] - ib),
                C=relerr(res["C"][0], st["C"]), X=relerr(res["X"][0], st["X"]), Yrec=relerr(res["Yrec"][0], st["Yrec"]))
    assert max(errs.values()) < 5e-9, errs


PERSISTENT_BIG = [
    ("cos_phase_r36", 36, dict(dyn="cos"), False),
    ("cos_phase_recursive_r41_rPSMF", 41, dict(dyn="cos", recursive=True, update_every=5), True),
    ("simplified_hooks_r45", 45, dict(coef_update=False, eta_full=False, pbar_predict=False), False),
    ("general_Q_r37", 37, dict(general_Q=True), True),
    ("schedules_r48", 48, dict(sched=True), False),
    ("d_1e5_r48_sixteen_row_passes", 48, dict(d=100_000, T=12), True),      # 512 rows per workgroup, 196 partial rows of 20 per fan-in thread
    ("d_1e5_r40_f32", 40, dict(d=100_000, T=12, storage="f32"), False),
]


@pytest.mark.parametrize("name,r,opts,robust", PERSISTENT_BIG, ids=[c[0] for c in PERSISTENT_BIG])
def test_persistent_kernel_at_33_to_48_matches_the_launched_form(name, r, opts, robust):
    """33 <= r <= 48 on the persistent per-step kernel (round 5: the hub with LDS-resident matrices, psmf_pstep.hip) against the two
    launches per timestep, which the tests above and test_hip_step_engine_random.py pin to the oracle: every mode the kernel takes,
    a run cut in three."""
    import os

    c = _capi()
    d, T = opts.get("d", 1100 + r), opts.get("T", 36)
    storage = opts.get("storage", "f64")
    Y, C0 = _problem(d, r, T, 900 + r, robust)
    if storage == "f32":
        Y, C0 = Y.astype(np.float32).astype(np.float64), C0.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(r)
    A = rng.standard_normal((r, r)) / np.sqrt(r)
    Q = 0.1 * np.eye(r) + (0.05 * (A @ A.T) if opts.get("general_Q") else 0.0)
    kw = {k: v for k, v in opts.items() if k in ("coef_update", "eta_full", "pbar_predict", "recursive", "update_every")}
    if opts.get("dyn") == "cos":
        kw["dyn_kind"] = c.DYN_COS_PHASE
    theta = 0.01 * (1 + np.arange(r)) / r if opts.get("dyn") == "cos" else None
    out = {}
    old = os.environ.get("PSMF_STEP_PERSISTENT")
    try:
        for persistent in (True, False):
            os.environ["PSMF_STEP_PERSISTENT"] = "1" if persistent else "0"
            f = c.DeviceFilter(d, r, storage=storage, robust=robust, engine="step", **kw)
            if opts.get("sched"):
                f.set_schedules(1.0 + 0.1 * np.sin(np.arange(T + 1)), 1.0 + 0.2 * np.cos(np.arange(T + 1)))
            f.upload_series(Y)
            f.set_state(C0, 0.1 * np.eye(r), np.eye(r), Q, np.zeros(r), rho=1.0, lambda0=1.8, theta=theta)
            if kw.get("recursive"):
                f.set_adam(np.zeros(r), np.zeros(r))
            assert f.geometry()["filter_kernel"] == ("psmf_pstep_k" if persistent else "psmf_sweep_solve")
            for a, b in ((0, 7), (7, 8), (8, T)):
                f.run(a, b)
            if persistent and d >= 100_000:
                assert f.geometry().get("filter_kernel") == "psmf_pstep_k"
            s = f.get_state()
            s["yp"] = f.y_pred(0, T)
            out[persistent] = s
            f.close()
    finally:
        if old is None:
            os.environ.pop("PSMF_STEP_PERSISTENT", None)
        else:
            os.environ["PSMF_STEP_PERSISTENT"] = old
    # float32 storage: the launched form rounds C every timestep, the persistent one keeps it in float64 on chip
    tol = 1e-10 if storage == "f64" else 1e-5
    for n in ("C", "V", "mu", "P", "Q", "yp") + (("theta", "gradsum") if theta is not None else ()):
        assert relerr(out[True][n], out[False][n]) < tol, (name, n, relerr(out[True][n], out[False][n]))
