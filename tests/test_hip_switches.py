"""Every environment switch that selects an alternative kernel or schedule, one small run each against the float64
oracle: a fallback nobody runs rots.  Switches are read per handle at psmf_create, so each case sets its variables, creates a
handle, and restores the environment.  (Switches with richer tests of their own: PSMF_FILTER6 / PSMF_FILTER6_DUAL in
test_hip_small_rank.py, PSMF_IMPUTE_V3 in test_hip_impute_small.py, PSMF_NS_PREDICT / PSMF_BULK_WGS / PSMF_FORCE_COLLECTIVE in
test_hip_filter.py, PSMF_BLOCK_PIPE / PSMF_FILTER3 with several shards in test_hip_multishard.py.)  GPU only: `pytest -m gpu`.
Reference: pypsmf/psmf/psmf.py:85-165, rpsmf.py:116-171 (one code path; every variant here must reproduce it)."""

import os
from contextlib import contextmanager

import numpy as np
import pytest

from conftest import relerr
from oracle import psmf_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@contextmanager
def _env(vars_):
    old = {k: os.environ.get(k) for k in vars_}
    os.environ.update(vars_)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _problem(d, r, T, robust, seed, f32):
    rng = np.random.default_rng(seed)
    Ct = rng.standard_normal((d, r))
    x = rng.standard_normal(r)
    Y = np.empty((T, d))
    for t in range(T):
        x = x + 0.1 * rng.standard_normal(r)
        Y[t] = Ct @ x + 0.3 * (rng.standard_t(3.0, d) if robust else rng.standard_normal(d))
    C0 = 0.1 * rng.standard_normal((d, r))
    if f32:
        Y = Y.astype(np.float32).astype(np.float64)
        C0 = C0.astype(np.float32).astype(np.float64)
    return Y, C0


# (id, environment, engine, storage, d, r, T, robust, schedules, kernel the handle must report -- None: any)
CASES = [
    # ---- blocked engine
    ("block_dual_off", {"PSMF_BLOCK_DUAL": "0"}, "block", "f32", 3000, 20, 150, False, False, "psmf_blk_filter7"),
    ("bulk2_off", {"PSMF_BULK2": "0"}, "block", "f32", 3000, 20, 150, True, False, "psmf_blk_filter3"),
    ("filter3_off", {"PSMF_FILTER3": "0"}, "block", "f32", 3000, 20, 150, False, False, "psmf_blk_filter2"),
    ("filter4_default", {}, "block", "f32", 3000, 24, 150, False, True, "psmf_blk_filter4"),
    ("filter4_off", {"PSMF_FILTER4": "0"}, "block", "f32", 3000, 24, 150, False, True, "psmf_blk_filter7"),
    ("filter4_filter7_off", {"PSMF_FILTER4": "0", "PSMF_FILTER7": "0"}, "block", "f32", 3000, 24, 150, False, True, "psmf_blk_filter"),
    ("ns_off", {"PSMF_NS": "0"}, "block", "f32", 3000, 32, 150, True, False, "psmf_blk_filter3"),
    ("chain_off", {"PSMF_BLOCK_CHAIN": "0"}, "block", "f32", 3000, 32, 200, False, False, "psmf_blk_filter3"),
    ("flags_off", {"PSMF_BLOCK_FLAGS": "0"}, "block", "f32", 3000, 32, 200, False, False, "psmf_blk_filter3"),
    ("pipe_off", {"PSMF_BLOCK_PIPE": "0"}, "block", "f32", 3000, 32, 200, True, False, "psmf_blk_filter3"),
    ("reserved_cus_off", {"PSMF_RESERVED_CUS": "0"}, "block", "f32", 3000, 32, 200, False, False, "psmf_blk_filter3"),
    ("engine_env_step", {"PSMF_ENGINE": "1"}, "auto", "f64", 2500, 20, 40, False, False, "psmf_pstep_k"),
    ("engine_env_block", {"PSMF_ENGINE": "2"}, "auto", "f64", 2500, 20, 80, True, False, "psmf_blk_filter3"),
    # ---- per-step engine: the persistent launch (default) and the two launches per timestep behind it, with their own switches
    ("persistent_default", {}, "step", "f64", 2500, 20, 40, True, False, "psmf_pstep_k"),
    ("persistent_off", {"PSMF_STEP_PERSISTENT": "0"}, "step", "f64", 2500, 20, 40, True, False, "psmf_sweep_solve"),
    ("persistent_off_f32", {"PSMF_STEP_PERSISTENT": "0"}, "step", "f32", 2500, 32, 40, False, False, "psmf_sweep_solve"),
    ("step_dual_off", {"PSMF_STEP_PERSISTENT": "0", "PSMF_STEP_DUAL": "0"}, "step", "f64", 2500, 20, 40, False, False, "psmf_sweep_solve"),
    ("persistent_step_dual_off", {"PSMF_STEP_DUAL": "0"}, "step", "f64", 2500, 20, 40, True, False, "psmf_pstep_k"),
    ("wave_solve_off", {"PSMF_STEP_WAVE_SOLVE": "0"}, "step", "f64", 2500, 20, 40, False, False, "psmf_sweep_solve"),
    ("sweep_threads_256", {"PSMF_STEP_PERSISTENT": "0", "PSMF_SWEEP_THREADS": "256"}, "step", "f64", 2500, 20, 40, True, False, "psmf_sweep_solve"),
    ("tail_reduce_off", {"PSMF_STEP_PERSISTENT": "0", "PSMF_TAIL_REDUCE": "0"}, "step", "f64", 2500, 20, 40, True, False, "psmf_sweep_solve"),
    ("tail_reduce_off_r40", {"PSMF_TAIL_REDUCE": "0", "PSMF_PSTEP_BIG": "0"}, "step", "f64", 1500, 40, 30, False, False, "psmf_sweep_solve"),
    ("pstep_big_default", {}, "step", "f64", 1500, 40, 30, True, False, "psmf_pstep_k"),
    ("pstep_big_off", {"PSMF_PSTEP_BIG": "0"}, "step", "f64", 1500, 40, 30, True, False, "psmf_sweep_solve"),
    ("tail_reduce_forced", {"PSMF_STEP_PERSISTENT": "0", "PSMF_TAIL_REDUCE": "1"}, "step", "f32", 60000, 32, 30, False, False, "psmf_sweep_solve"),
    ("wave_big_off", {"PSMF_STEP_WAVE_BIG": "0"}, "step", "f64", 1500, 40, 30, False, False, "psmf_sweep_solve"),
    ("serial_wide_off", {"PSMF_SERIAL_WIDE": "0", "PSMF_PSTEP_BIG": "0"}, "step", "f64", 1500, 40, 30, True, False, "psmf_sweep_solve"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_switch_reproduces_the_oracle(case):
    from rpsmf_amd import _capi as c

    _, env, engine, storage, d, r, T, robust, sched, want = case
    Y, C0 = _problem(d, r, T, robust, 4200 + len(env) + r, storage == "f32")
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    cut = T // 3
    with _env(env):
        f = c.DeviceFilter(d, r, storage=storage, robust=robust, engine=engine)
        if sched:          # constant schedules: the oracle's arithmetic is the unscheduled one, the device takes the scheduled kernels
            f.set_schedules(np.ones(T + 1), np.ones(T + 1))
        f.upload_series(Y)
        f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
        kern = f.geometry()["filter_kernel"]
        assert want is None or kern == want, (case[0], kern)
        tol = 1e-9 if storage == "f64" else 1e-5
        for a, b in ((0, cut), (cut, T)):
            st, Yp, _ = O.run_epoch(st, Y[a:b], O.Mode(robust=robust), O.RandomWalkDyn(), k0=a, want_grad=False)
            f.run(a, b)
            s = f.get_state()
            for n in ("C", "V", "mu", "P"):
                assert relerr(s[n], getattr(st, n)) < tol, (case[0], n, b, relerr(s[n], getattr(st, n)))
            assert relerr(f.y_pred(a, b - a), Yp) < tol, (case[0], "y_pred", b)
        f.close()


def test_two_launch_engine_graph_and_eager_agree_bitwise():
    """The hipGraph replay of the two launches per timestep against eager launches (the persistent kernel is neither)."""
    from rpsmf_amd import _capi as c

    d, r, T = 3000, 16, 300
    Y, C0 = _problem(d, r, T, False, 77, False)
    outs = []
    with _env({"PSMF_STEP_PERSISTENT": "0"}):
        for graph in (True, False):
            f = c.DeviceFilter(d, r, storage="f64", engine="step", use_graph=graph)
            f.upload_series(Y)
            f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
            assert f.geometry()["filter_kernel"] == "psmf_sweep_solve"
            f.run(0, T)
            outs.append(f.get_state())
            f.close()
    for n in ("C", "V", "mu", "P"):
        assert np.array_equal(outs[0][n], outs[1][n]), n


def test_persistent_kernel_is_deterministic_and_splits_like_one_run():
    """Two identical runs of the persistent kernel give the same bits (fixed summation orders, no float atomics); a run cut into
    three launches lands on the one-launch result (the float64 copy of C is rounded to float32 storage at the cuts: 1e-6)."""
    from rpsmf_amd import _capi as c

    d, r, T = 20_000, 32, 120
    Y, C0 = _problem(d, r, T, True, 78, True)

    def run(cuts, storage):
        f = c.DeviceFilter(d, r, storage=storage, robust=True, engine="step")
        f.upload_series(Y)
        f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
        assert f.geometry()["filter_kernel"] == "psmf_pstep_k"
        for a, b in zip(cuts[:-1], cuts[1:]):
            f.run(a, b)
        s = f.get_state()
        s["yp"] = f.y_pred(0, T)
        f.close()
        return s

    a, b = run((0, T), "f32"), run((0, T), "f32")
    for n in ("C", "V", "mu", "P", "yp"):
        assert np.array_equal(a[n], b[n]), n
    one, three = run((0, T), "f64"), run((0, 17, 64, T), "f64")
    for n in ("C", "V", "mu", "P", "yp"):
        assert relerr(three[n], one[n]) < 1e-12, (n, relerr(three[n], one[n]))
    cut32 = run((0, 17, 64, T), "f32")
    for n in ("C", "V", "mu", "P", "yp"):
        assert relerr(cut32[n], a[n]) < 2e-6, (n, relerr(cut32[n], a[n]))


@pytest.mark.parametrize("r", [9, 24, 40, 64])
@pytest.mark.parametrize("wgram", ["1", "0"], ids=["wgram_mfma", "wgram_vector"])
def test_weighted_gram_of_a_nonuniform_R_both_kernels(wgram, r):
    """Non-uniform diagonal R (psmf.py:140-152 with a diagonal R): the step's weighted Gram sum_i c_i c_i^T / (rho_i + s) on the
    matrix cores (psmf_wgram_mfma, the default) and on the vector units (PSMF_WGRAM_MFMA=0), float32 and float64 storage."""
    from rpsmf_amd import _capi as c

    d, T = 3001, 40
    rng = np.random.default_rng(90 + r)
    rho = 0.3 + 2.0 * rng.random(d)
    for storage in ("f64", "f32"):
        Y, C0 = _problem(d, r, T, True, 500 + r, storage == "f32")
        V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
        st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=rho, lam=1.8)
        st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=True), O.RandomWalkDyn(), want_grad=False)
        with _env({"PSMF_WGRAM_MFMA": wgram}):
            f = c.DeviceFilter(d, r, storage=storage, robust=True, engine="step", nonuniform_R=True)
            f.set_row_noise(rho)
            f.upload_series(Y)
            f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
            f.run(0, T)
            s = f.get_state()
            tol = 1e-9 if storage == "f64" else 1e-5
            for n in ("C", "V", "mu", "P"):
                assert relerr(s[n], getattr(st, n)) < tol, (wgram, storage, n, relerr(s[n], getattr(st, n)))
            assert relerr(f.y_pred(0, T), Yp) < tol
            f.close()
