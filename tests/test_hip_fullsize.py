"""Parity at BASELINE.json's full sizes and horizons (configs B, C, D and the single-GPU shard of E), through the C ABI,
against the CPU oracle.  GPU only: `pytest -m gpu`.  Sized so that the oracle legs take about five minutes in all on the
GPU box's host cores (the oracle is numpy float64, O(d r^2) per timestep).

Why these exist: the f32-storage error of the recursion peaks around k = 300 and plateaus only after k = 500-2000
(SURVEY section 0), and the Newton-Schulz starts of the blocked engine are carried over hundreds of blocks -- short
series cannot show either.  Tolerance: BASELINE.json north_star, 1e-5 relative (max-abs over max-abs), f32 storage.
"""

import numpy as np
import pytest

from conftest import load_golden as load_golden_fixture, relerr, relerr_elementwise
from oracle import psmf_oracle as O
from oracle.impute_oracle import impute_filter

pytestmark = pytest.mark.gpu

TOL = 1e-5
# Entry-by-entry bound (entries >= 1e-3 of the largest), as a multiple of the tolerance of the normalised norm.  With float32 storage
# of C, y, y_hat the worst entry-by-entry errors measured over configs B, C, E (two epochs of 10 000 timesteps, PSMF and rPSMF) are
# 6.9e-4 (y_hat of the tracked series), 3.5e-4 (C), 1.5e-4 (V), 5.4e-5 (P) -- profiles/r4_parity_fullsize.txt; with float64 storage
# 3.4e-9 on C (BENCH r5, other_configs.E_f64_storage).  150 x tol = 1.5e-3 leaves a factor 2 over the worst measured value, so a
# regression by a small factor fails here (the bound was 1000 x tol until round 5).
ELEMENTWISE_FACTOR = 150.0


def _capi():
    from rpsmf_amd import _capi

    return _capi


def _bench_problem(d, r, T, robust):
    """The synthetic workload of bench.py / SURVEY 8(d): data.py semantics, C0 = 0.1 randn, V0 = 0.1 I, P0 = I, Q = 0.1 I,
    R = I, mu0 = 0, seeds of the reference Makefile (35853 PSMF, 35833 rPSMF with lambda0 = 1.8, Student-t noise dof 3)."""
    seed = 35833 if robust else 35853
    Y = O.synthetic_series(d, r, T, seed, noise="t" if robust else "normal", dtype=np.float32)
    rng = np.random.default_rng(seed + 7)
    C0 = (0.1 * rng.standard_normal((d, r))).astype(np.float32).astype(np.float64)
    return Y, C0


def _checkpointed_parity(d, r, T, robust, checkpoints, engine="auto", storage="f32", tol=TOL):
    c = _capi()
    Y, C0 = _bench_problem(d, r, T, robust)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mode = O.Mode(robust=robust)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    st, Yp, trace = O.run_epoch(st, Y.astype(np.float64), mode, O.RandomWalkDyn(), keep=checkpoints, want_grad=False)
    f = c.DeviceFilter(d, r, robust=robust, storage=storage, engine=engine)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    worst = {}
    k_prev = 0
    for k in checkpoints:
        f.run(k_prev, k)
        k_prev = k
        s = f.get_state()
        ref = trace[k][0]
        for name in ("C", "V", "mu", "P"):
            e = relerr(s[name], getattr(ref, name))
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < tol, (name, k, e)
            # the same comparison entry by entry (entries >= 1e-3 of the largest): reported, and bounded at ELEMENTWISE_FACTOR x the
            # tolerance of the normalised norm -- an entry a thousand times smaller than the largest may carry the same absolute error
            ee, share = relerr_elementwise(s[name], getattr(ref, name))
            worst[name + "_elementwise"] = max(worst.get(name + "_elementwise", 0.0), ee)
            worst[name + "_elementwise_share"] = share
            assert ee < ELEMENTWISE_FACTOR * tol, (name, k, ee)
        if robust:
            assert relerr(s["rho"], ref.rho) < tol and relerr(s["lam"], ref.lam) < 1e-12
    e = relerr(f.y_pred(0, T), Yp)
    assert e < tol, ("y_pred", e)
    geo = f.geometry()
    f.close()
    return worst, geo


@pytest.mark.parametrize("robust", [False, True], ids=["configB_PSMF", "configC_rPSMF"])
def test_config_B_C_full_size(robust):
    """BASELINE configs B / C: d = 10 000, r = 20, T = 5 000, f32 storage, the engine the library selects by itself,
    checked at k = 300 (the transient's peak), 1 000 and 5 000 (psmf.py:85-102, rpsmf.py:116-184)."""
    worst, geo = _checkpointed_parity(10_000, 20, 5_000, robust, (300, 1000, 5000))
    assert geo["engine"] == "block"
    print("config", "C" if robust else "B", "worst rel-err:", worst)


def test_config_E_single_gpu_shard_size():
    """BASELINE config E at the size one GPU holds when N = 1, against the oracle run HERE: d = 100 000, r = 32, the first 300
    timesteps (the peak of the f32 transient; the oracle needs ~0.1 s per timestep at this size), f32 storage, chained blocked
    engine.  The whole horizon -- 10 000 timesteps and a second carried-state epoch, PSMF and rPSMF -- is covered by the
    stored oracle answers of test_config_E_full_horizon_two_epochs_vs_oracle_fixture below."""
    worst, geo = _checkpointed_parity(100_000, 32, 300, False, (100, 300))
    assert geo["engine"] == "block" and geo["block_steps"] == 32
    print("config E (1 GPU) worst rel-err:", worst)


@pytest.mark.parametrize("which", ["psmf", "rpsmf"])
def test_config_E_full_horizon_two_epochs_vs_oracle_fixture(which):
    """BASELINE config E over its WHOLE horizon and into a second, carried-state epoch -- what bench.py's headline number is
    quoted on: d = 100 000, r = 32, T = 10 000, bench.py's own series and initial state, PSMF and rPSMF.  The oracle's answers
    were computed once in the build container (tests/golden/make_golden_fullsize.py, ~15 min of CPU per variant) and are stored
    as r-sized state + fixed sketches of the d-sized quantities at k = 300 ... 20 000; y_hat of four series is compared at
    EVERY timestep of both epochs.  Tolerance 1e-5 (north_star), f32 storage, the engine the library selects."""
    import bench
    from golden.make_golden_fullsize import sketch_matrix

    g = load_golden_fixture(f"fullsize_E_{which}")
    d, r, T, robust = int(g["d"]), int(g["r"]), int(g["T"]), bool(g["robust"])
    c = _capi()
    series = bench.Series(d, r, T, int(g["seed"]), 0, d, robust)
    st0 = bench.init_state(d, r, int(g["seed"]))
    f = c.DeviceFilter(d, r, robust=robust, storage="f32")
    for a, Yc in series.chunks(chunk=1000):
        f.upload_series(Yc, t0=a, T_total=T)
    f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
    S, rows, track = sketch_matrix(d), g["rows"], g["track"]
    worst = {}

    def check(name, got, ref, k, tol=TOL):
        e = relerr(got, ref)
        worst[name] = max(worst.get(name, 0.0), e)
        assert e < tol, (which, name, k, e)
        if np.ndim(ref) >= 1 and np.size(ref) > 1:          # entry by entry as well (entries >= 1e-3 of the largest): reported
            ee, _ = relerr_elementwise(got, ref)
            worst[name + "_elementwise"] = max(worst.get(name + "_elementwise", 0.0), ee)
            assert ee < ELEMENTWISE_FACTOR * tol, (which, name, k, ee)

    epochs = int(g["epochs"])
    cps = [int(k) for k in g["checkpoints"]]
    for ep in range(epochs):
        if ep > 0 and robust:          # epoch start (rpsmf.py:106-114): Q, R, lambda back to their initial values; C, V, mu, P carried
            f.set_state(Q=st0["Q"], rho=st0["rho"], lambda0=st0["lam"])
        k_prev = 0
        for kg in [c_ for c_ in cps if ep * T < c_ <= (ep + 1) * T]:
            k = kg - ep * T
            f.run(k_prev, k)
            k_prev = k
            s = f.get_state()
            p = f"k{kg}_"
            for name in ("V", "P", "mu"):
                check(name, s[name], g[p + name], kg)
            check("StC", S.T @ s["C"], g[p + "StC"], kg)
            check("Crows", s["C"][rows], g[p + "Crows"], kg)
            check("yhat_rows", f.y_pred(k - 1, 1)[0][rows], g[p + "yhat_rows"], kg)
            check("eta", s["eta"], g[p + "eta"], kg)
            check("N", s["N"], g[p + "N"], kg)
            if robust:
                check("rho", s["rho"], g[p + "rho"], kg)
                check("q", s["Q"][0, 0], g[p + "q"], kg)
                assert relerr(s["lam"], g[p + "lam"]) < 1e-12
        if k_prev < T:
            f.run(k_prev, T)
        yt = np.concatenate([f.y_pred(a, 1000, dtype=np.float32)[:, track] for a in range(0, T, 1000)])
        check("yhat_track", yt, g["yhat_track"][ep * T:(ep + 1) * T], (ep + 1) * T)
    geo = f.geometry()
    cnt = f.counters()
    f.close()
    assert geo["engine"] == "block" and geo["block_steps"] == 32
    print(f"config E {which}, two epochs of {T}: worst rel-err", {k: f"{v:.2e}" for k, v in worst.items()}, "counters", cnt)


def test_per_step_engine_tolerance_r40():
    """r > 32 runs on the per-step engine (one rounding of C per timestep with f32 storage would breach 1e-5 around
    k = 300: DESIGN section 5), so the library stores C in float64 there.  d = 20 000, r = 40, T = 1 000."""
    c = _capi()
    d, r, T = 20_000, 40, 1_000
    worst, geo = _checkpointed_parity(d, r, T, False, (300, 1000), storage="auto")
    assert geo["engine"] in ("step", "block")
    print("r = 40 worst rel-err:", worst)


# ----------------------------------------------------------------------------- config D
def _gas_sensor_standin(d, n, seed=20160930):
    """Synthetic stand-in of the gas-sensor array (the CSV is not in the reference checkout, .MISSING_LARGE_BLOBS):
    smooth random-walk channels + 1 % native NaN, as tools/bench_impute.py and SURVEY 8(d)."""
    rng = np.random.default_rng(seed)
    Yorig = np.cumsum(0.05 * rng.standard_normal((d, n)), axis=1) + 10.0 * rng.random((d, 1))
    Yorig[rng.random((d, n)) < 0.01] = np.nan
    return Yorig


def _draw(Yorig, r, seeds):
    from rpsmf_amd import impute_harness as H

    np.random.seed(123)
    M, Mm, C0, X0 = [], [], [], []
    for _ in range(seeds):
        p = H.draw_problem(Yorig, 40, r)
        M.append(p["M"].astype(np.uint8))
        Mm.append(p["Mmiss"].astype(np.uint8))
        C0.append(p["C"])
        X0.append(p["X"])
    return np.stack(M), np.stack(Mm), np.stack(C0), np.stack(X0)


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_config_D_full_length_seed_vs_oracle(robust):
    """BASELINE config D, one seed at full length: 19 x 295 719, r = 10, Iter = 2, 40 % missing
    (ExperimentImpute/PSMF.py:59-93, rPSMF.py:75-135): 591 438 sequential masked steps against the oracle."""
    from rpsmf_amd import impute

    d, n, r = 19, 295_719, 10
    Yorig = _gas_sensor_standin(d, n)
    Yint = np.nan_to_num(Yorig, nan=0.0)
    M, Mm, C0, X0 = _draw(Yorig, r, 1)
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    Xo = X0[0].copy()
    ep, ef, ib, st = impute_filter(Yint * M[0], C0[0], Xo, M[0], Mm[0].astype(float), V, Q, 10.0, P, 2, 2, Yint, 0.0,
                                   robust=robust, lambda0=1.8, return_state=True)
    res = impute.impute_batch(Yint, M, Mm, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8)
    # float64 on both sides; 6e5 sequential steps of a contracting filter: agreement stays at round-off level
    assert relerr(res["Epred"][0], ep[0, 1:]) < 1e-8 and relerr(res["Efull"][0], ef[0, 1:]) < 1e-8
    # coverage = a COUNT of held-out entries inside the band over their number (2.2 million here): an entry within round-off of
    # a band edge may fall on either side (the two sides sum the masked Gram in different orders) -- at most two such entries
    assert abs(res["inside"][0] - ib) * float(Mm[0].sum()) < 2.5
    assert relerr(res["C"][0], st["C"]) < 1e-7 and relerr(res["X"][0], st["X"]) < 1e-7


def test_config_D_fifty_seed_batch_equals_single_runs():
    """The 50-seed batch (one launch, one workgroup per replica) against the same replicas run one by one: every replica of
    a 20 000-column prefix, and three replicas (first, middle, last) of the full-length batch.  Replicas are independent,
    the arithmetic per replica is the same code: bit-identical."""
    from rpsmf_amd import impute

    d, r, seeds = 19, 10, 50
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    for n, singles in ((20_000, range(seeds)), (295_719, (0, 24, 49))):
        Yorig = _gas_sensor_standin(d, n)
        Yint = np.nan_to_num(Yorig, nan=0.0)
        M, Mm, C0, X0 = _draw(Yorig, r, seeds)
        C0_before = C0.copy()
        res = impute.impute_batch(Yint, M, Mm, C0, X0, V, Q, 10.0, P, 2, 2, robust=False)
        assert np.array_equal(C0, C0_before)          # inputs are not mutated (the reference rebinds C, PSMF.py:80)
        assert np.all(np.isfinite(res["Epred"])) and np.all(np.isfinite(res["Efull"]))
        assert len(set(np.round(res["Epred"][:, -1], 12))) == seeds      # 50 different problems, 50 different answers
        for i in singles:
            one = impute.impute_batch(Yint, M[i], Mm[i], C0[i], X0[i], V, Q, 10.0, P, 2, 2, robust=False)
            for k in ("Epred", "Efull", "inside", "C", "X"):
                assert np.array_equal(one[k][0], res[k][i]), (n, i, k)


# ----------------------------------------------------------------------------- filter4 at size
@pytest.mark.parametrize("robust,q,recursive", [(False, 0.1, False), (True, 1.0, False), (False, 1.0, True)],
                         ids=["PSMF_q0.1", "rPSMF_q1", "PSMFRecursive_q1"])
def test_filter4_cos_phase_at_size(robust, q, recursive):
    """psmf_blk_filter4 (diagonal-Jacobian dynamics: sequential Newton-Schulz inversions with sweep fallback, theta gradient,
    in-loop Adam) at d = 20 000, r = 20 against the oracle (pypsmf/psmf/psmf.py:90-115,167-177,287-304; rpsmf.py:116-184).

    The horizon is short on purpose.  The FULL filter with f = cos(2 pi theta t + x) is a chaotic recursion on this data: any
    two float64 implementations separate by a factor ~1e5 every 50 timesteps (the general kernel, this kernel and the oracle agree
    to 1e-11 at k = 50, 1e-6 at k = 100 and not at all at k = 200: tools/probe_f4g.py, float64 storage) -- the reference itself
    would not reproduce its own run under a different BLAS.  So: float64 storage over 48 timesteps (two blocks) at 1e-6, float32
    storage over the first 30 at 1e-5 (north_star)."""
    c = _capi()
    d, r, T = 20_000, 20, 48
    Y, C0 = _bench_problem(d, r, T, robust)
    rng = np.random.default_rng(77)
    theta0 = 0.05 + 0.1 * rng.random(r)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), q * np.eye(r)
    dyn = O.CosPhaseDyn(r)
    Y64 = Y.astype(np.float64)
    mode = O.Mode(robust=robust)
    for storage, Tn, tol in (("f64", 48, 1e-6), ("f32", 30, TOL)):
        st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta0.copy(), gradsum=np.zeros(r))
        if recursive:
            m = v = np.zeros(r)
            Yp = np.empty((Tn, d))
            for k in range(1, Tn + 1):
                st, info = O.lowrank_step(st, Y64[k - 1], k, mode, dyn)
                Yp[k - 1] = info.y_pred
                st.theta, m, v = O.adam_update(st.theta, st.gradsum, m, v, k)      # update_every = 1 (psmf.py:299-304)
                st.gradsum = np.zeros(r)
        else:
            st, Yp, _ = O.run_epoch(st, Y64[:Tn], mode, dyn)
        kw = dict(recursive=True, update_every=1, adam_lr=1e-3) if recursive else {}
        f = c.DeviceFilter(d, r, robust=robust, storage=storage, dyn_kind=c.DYN_COS_PHASE, **kw)
        f.upload_series(Y)
        f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8, theta=theta0)
        assert f.geometry()["filter_kernel"] == "psmf_blk_filter4"      # (the choice depends on the Q just uploaded: Q = q I)
        f.zero_gradsum()
        if recursive:
            f.set_adam(np.zeros(r), np.zeros(r))
        f.counters(reset=True)
        f.run(0, Tn)
        s = f.get_state()
        cnt = f.counters()
        for name in ("C", "V", "mu", "P"):
            assert relerr(s[name], getattr(st, name)) < tol, (storage, name, relerr(s[name], getattr(st, name)))
        assert relerr(f.y_pred(0, Tn), Yp) < tol
        if recursive:
            assert relerr(s["theta"], st.theta) < tol
        else:
            assert relerr(s["gradsum"], st.gradsum) < 10 * tol
        assert cnt["ns_steps"] + cnt["sweep_steps"] == Tn
        f.close()


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_filter5_simplified_cos_phase_at_size(robust):
    """The ExperimentSynthetic configuration (simplified hooks, f = cos(2 pi theta t + x): synthetic_psmf.py:78-106,
    synthetic_rpsmf.py:82-124) at d = 20 000, r = 20, T = 1 000, f32 storage, on psmf_blk_filter5 -- state, predictions and the
    theta gradient against the oracle.  (Unlike the full filter this recursion is stable: mu_k = mu_bar_k does not see the data.)"""
    c = _capi()
    d, r, T = 20_000, 20, 1_000
    Y, C0 = _bench_problem(d, r, T, robust)
    theta0 = 1e-3 * np.arange(1, r + 1) + 0.01 * np.random.default_rng(5).random(r)
    V0, P0, Q = 0.1 * np.eye(r), np.zeros((r, r)), np.zeros((r, r))
    mode = O.Mode(robust=robust, coef_update=False, eta_full=False, pbar_predict=False)
    mu0 = np.random.default_rng(6).standard_normal(r)
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta0.copy(), gradsum=np.zeros(r))
    st, Yp, _ = O.run_epoch(st, Y.astype(np.float64), mode, O.CosPhaseDyn(r))
    f = c.DeviceFilter(d, r, robust=robust, storage="f32", dyn_kind=c.DYN_COS_PHASE, coef_update=False, eta_full=False, pbar_predict=False)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, mu0, rho=1.0, lambda0=1.8, theta=theta0)
    assert f.geometry()["filter_kernel"] == "psmf_blk_filter5"
    f.zero_gradsum()
    f.run(0, T)
    s = f.get_state()
    for name in ("C", "V", "mu"):
        assert relerr(s[name], getattr(st, name)) < TOL, (name, relerr(s[name], getattr(st, name)))
    assert relerr(f.y_pred(0, T), Yp) < TOL
    assert relerr(s["gradsum"], st.gradsum) < 10 * TOL
    if robust:
        assert relerr(s["rho"], st.rho) < TOL and relerr(s["lam"], st.lam) < 1e-12
    f.close()


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_filter6_fourier_and_random_walk_at_size(robust):
    """The small-rank block filter at d = 10 000, f32 storage, T = 300: ExperimentBeijing's dynamics family (FourierBasis N = 2,
    r = 10: beijing_psmf.py:97-140) -- state, predictions and the theta gradient against the oracle on the same callable -- and the
    default model (random walk, r = 12) on the side-by-side form of the kernel."""
    from rpsmf_amd import nonlinearities as NL

    c = _capi()
    d, T = 10_000, 300          # (the oracle's complex-step Jacobians over 480 parameters set the price of this test)
    for r, nl in ((10, NL.FourierBasis(10, N=2)), (12, NL.RandomWalk())):
        Y, C0 = _bench_problem(d, r, T, robust)
        rng = np.random.default_rng(7 + r)
        theta0 = 0.1 * rng.random(nl.n_params)                                   # beijing_psmf.py:117
        V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
        mu0 = 0.2 * rng.standard_normal(r)
        dyn = O.CallableDyn(nl, nl.n_params) if nl.n_params else O.RandomWalkDyn()
        st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta0.copy(), gradsum=np.zeros(nl.n_params))
        st, Yp, _ = O.run_epoch(st, Y.astype(np.float64), O.Mode(robust=robust), dyn, want_grad=bool(nl.n_params))
        f = c.DeviceFilter(d, r, robust=robust, storage="f32", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags, dyn_terms=nl.device_terms)
        f.upload_series(Y)
        f.set_state(C0, V0, P0, Q, mu0, rho=1.0, lambda0=1.8, theta=theta0 if nl.n_params else None)
        assert f.geometry()["filter_kernel"] == ("psmf_blk_filter6" if nl.n_params else "psmf_blk_filter6d")
        if nl.n_params:
            f.zero_gradsum()
        f.run(0, T)
        s = f.get_state()
        for name in ("C", "V", "mu", "P"):
            assert relerr(s[name], getattr(st, name)) < TOL, (r, name, relerr(s[name], getattr(st, name)))
        assert relerr(f.y_pred(0, T), Yp) < TOL
        if nl.n_params:
            assert relerr(s["gradsum"], st.gradsum) < 10 * TOL
        f.close()
