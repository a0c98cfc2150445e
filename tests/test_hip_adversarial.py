"""Adversarial series against the carried Newton-Schulz starts of the blocked engine (psmf_blk_filter3): an abrupt level
shift, blocks of 1000-sigma outliers, a process noise of 1e-8, a nearly singular P0 (tests/adversarial_cases.py) -- device
(f32 storage, the engine the library selects) against the CPU oracle at checkpoints around the event and at the end, 1e-5
(north_star), plus what the inversion counters (psmf_counters) say happened: where the previous step's inverse is useless as
a start the direct symmetric sweep must take over, and the result must not care.  GPU only: `pytest -m gpu`.
Reference: pypsmf/psmf/psmf.py:85-102,140-165 (np.linalg.inv every step, no carried state to go stale)."""

import numpy as np
import pytest

from adversarial_cases import CASES, make_case
from conftest import relerr
from oracle import psmf_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5
D, T = 10_000, 1000          # (d = 20 000 when profiles/r3_parity_adversarial.txt was written; the oracle sets the price of these tests)


# every case on the headline kernel (PSMF, r = 32: psmf_blk_filter3), the two with the hardest starts also for rPSMF at r = 20
# (masked iterates, omega-scaled Q and R); all ten combinations passed when this file was written (profiles/r3_parity_adversarial.txt)
COMBOS = [(n, 32, False) for n in CASES] + [("outlier_block", 20, True), ("tiny_Q", 20, True)]


@pytest.mark.parametrize("name,r,robust", COMBOS, ids=[f"{n}-{'rPSMF' if rb else 'PSMF'}_r{r}" for n, r, rb in COMBOS])
def test_adversarial_series_vs_oracle(name, r, robust):
    from rpsmf_amd import _capi

    cs = make_case(name, D, r, T, robust)
    st = O.State(C=cs["C0"].copy(), V=cs["V0"], mu=np.zeros(r), P=cs["P0"], Q=cs["Q"], rho=1.0, lam=1.8)
    st, Yp, trace = O.run_epoch(st, cs["Y"].astype(np.float64), O.Mode(robust=robust), O.RandomWalkDyn(), keep=cs["checkpoints"],
                                want_grad=False)
    f = _capi.DeviceFilter(D, r, robust=robust, storage="f32")
    assert f.geometry()["engine"] == "block"
    f.upload_series(cs["Y"])
    f.set_state(cs["C0"], cs["V0"], cs["P0"], cs["Q"], np.zeros(r), rho=1.0, lambda0=1.8)
    k_prev, counters, worst = 0, {}, 0.0
    for k in cs["checkpoints"]:
        f.counters(reset=True)
        f.run(k_prev, k)
        s = f.get_state()
        counters[k] = f.counters()
        ref = trace[k][0]
        for n in ("C", "V", "mu", "P"):
            e = relerr(s[n], getattr(ref, n))
            worst = max(worst, e)
            assert e < TOL, (name, n, k, e)
        if robust:
            assert relerr(s["rho"], ref.rho) < TOL and relerr(s["Q"], ref.Q) < TOL
        k_prev = k
    e = relerr(f.y_pred(0, T), Yp)
    f.close()
    assert e < TOL, (name, "y_pred", e)
    # every timestep was inverted one way or the other, and the counters add up
    for a, k in zip((0,) + cs["checkpoints"][:-1], cs["checkpoints"]):
        c = counters[k]
        assert c["ns_steps"] + c["sweep_steps"] == k - a, (k, c)
    ev = cs["event"]
    after = counters[ev + 40] if (ev + 40) in counters else None
    if name in ("outlier_block", "outliers_then_quiet"):
        # 1000-sigma innovations move kappa and G by orders of magnitude from one step to the next: the carried start is useless,
        # the iteration is given up (ns_failed) and the direct sweep inverts those steps
        assert after["ns_failed"] >= 1 and after["sweep_steps"] >= 1, after
    if name == "tiny_Q":
        # W = (M / beta + I / q)^-1 with 1 / q = 1e8: the early steps cannot be started from anything
        assert counters[cs["checkpoints"][0]]["sweep_steps"] >= 1
    tail = counters[T]
    assert tail["sweep_steps"] <= 0.2 * (T - cs["checkpoints"][-2]), tail        # ... and the filter goes back to iterating
    print(name, "rPSMF" if robust else "PSMF", f"worst rel-err {worst:.1e}, y_pred {e:.1e}",
          {k: (c["ns_steps"], c["sweep_steps"], c["ns_iterations"], c["ns_failed"]) for k, c in counters.items()})


# ---- the same cases on psmf_blk_filter4 (round 5): its Newton-Schulz starts are given up only beyond a residual of 0.6
# (PSMF_NS_FAR4; filter3: 0.3) because its fallback is the cheap wave-local sweep -- a threshold no adversarial series had run under.
# A random walk with R_k / Q_k schedules is kept off filter3 (blk_dual_ok) and runs the full filter, Q = q I, at 17 <= r <= 32 on
# filter4; constant schedules (all ones) leave the oracle's arithmetic exactly the unscheduled one.  Not chaotic (unlike the full
# cos-phase filter, tools/probe_f4g.py), so the horizon can be long.
F4_COMBOS = [("tiny_Q", 24, False), ("tiny_P0", 24, False), ("outlier_block", 32, False), ("tiny_Q", 32, False), ("level_shift", 20, False)]


@pytest.mark.parametrize("name,r,robust", F4_COMBOS, ids=[f"f4-{n}-{'rPSMF' if rb else 'PSMF'}_r{r}" for n, r, rb in F4_COMBOS])
def test_adversarial_series_on_filter4(name, r, robust):
    from rpsmf_amd import _capi

    d, T = 6_000, 600
    cs = make_case(name, d, r, T, robust)
    st = O.State(C=cs["C0"].copy(), V=cs["V0"], mu=np.zeros(r), P=cs["P0"], Q=cs["Q"], rho=1.0, lam=1.8)
    cps = tuple(k for k in cs["checkpoints"] if k <= T)
    st, Yp, trace = O.run_epoch(st, cs["Y"].astype(np.float64), O.Mode(robust=robust), O.RandomWalkDyn(), keep=cps, want_grad=False)
    f = _capi.DeviceFilter(d, r, robust=robust, storage="f32")
    f.set_schedules(np.ones(T + 1), np.ones(T + 1))
    f.upload_series(cs["Y"])
    f.set_state(cs["C0"], cs["V0"], cs["P0"], cs["Q"], np.zeros(r), rho=1.0, lambda0=1.8)
    assert f.geometry()["filter_kernel"] == "psmf_blk_filter4", f.geometry()
    k_prev, worst, tot = 0, 0.0, dict(ns_steps=0, sweep_steps=0, ns_failed=0)
    for k in cps:
        f.counters(reset=True)
        f.run(k_prev, k)
        s = f.get_state()
        c = f.counters()
        for key in tot:
            tot[key] += c[key]
        ref = trace[k][0]
        for n in ("C", "V", "mu", "P"):
            e = relerr(s[n], getattr(ref, n))
            worst = max(worst, e)
            assert e < TOL, (name, n, k, e, c)
        k_prev = k
    e = relerr(f.y_pred(0, T), Yp)
    f.close()
    assert e < TOL, (name, "y_pred", e)
    assert tot["ns_steps"] + tot["sweep_steps"] == T, tot
    print("filter4", name, f"r={r} worst rel-err {worst:.1e}, y_pred {e:.1e}", tot)


STEP_COMBOS = [(n, 32, False, "f64") for n in CASES] + [("tiny_Q", 20, True, "f64"), ("tiny_P0", 40, False, "f64"), ("tiny_Q", 44, True, "f64"),
                                                     ("outlier_block", 32, False, "f32")]


@pytest.mark.parametrize("name,r,robust,storage", STEP_COMBOS, ids=[f"step-{n}-{'rPSMF' if rb else 'PSMF'}_r{r}_{s}" for n, r, rb, s in STEP_COMBOS])
def test_adversarial_series_on_the_per_step_engine(name, r, robust, storage):
    """The same series on the persistent per-step kernel, whose r x r inversions are direct -- by blocks since round 5 (psmf_ns.hip:
    leading block, Schur complement on one tile; r = 20, 32: 2 x 2 tiles, r = 40, 44: 3 x 3): q = 1e-8, a nearly singular P0 and
    1000-sigma innovations against the float64 oracle at float64 tolerance.  (q = 1e-8 with P0 = I: psmf_set_state takes the
    inversions one after the other there -- the side-by-side form's Lbar' = (I / q - W / q^2) / omega cancels eight digits.)"""
    from rpsmf_amd import _capi

    cs = make_case(name, D, r, T, robust)
    st = O.State(C=cs["C0"].copy(), V=cs["V0"], mu=np.zeros(r), P=cs["P0"], Q=cs["Q"], rho=1.0, lam=1.8)
    st, Yp, trace = O.run_epoch(st, cs["Y"].astype(np.float64), O.Mode(robust=robust), O.RandomWalkDyn(), keep=cs["checkpoints"],
                                want_grad=False)
    f = _capi.DeviceFilter(D, r, robust=robust, storage=storage, engine="step")
    f.upload_series(cs["Y"])
    f.set_state(cs["C0"], cs["V0"], cs["P0"], cs["Q"], np.zeros(r), rho=1.0, lambda0=1.8)
    assert f.geometry()["filter_kernel"] == "psmf_pstep_k"
    tol = 1e-8 if storage == "f64" else TOL
    k_prev = 0
    for k in cs["checkpoints"]:
        f.run(k_prev, k)
        s = f.get_state()
        ref = trace[k][0]
        for n in ("C", "V", "mu", "P"):
            e = relerr(s[n], getattr(ref, n))
            assert e < tol, (name, n, k, e)
        k_prev = k
    e = relerr(f.y_pred(0, T), Yp)
    f.close()
    assert e < tol, (name, "y_pred", e)
