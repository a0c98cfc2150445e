"""The theta path on the device: state-transition functions beyond the random walk, evaluated inside the
blocked engine with analytic Jacobians (psmf_dyn.hip) or host-stepped (psmf_step_host), the in-loop optimiser of the
recursive classes, and the per-step R_k / Q_k schedules of PSMFIter -- against the CPU oracle (complex-step derivatives of the
same callables) and the reference-generated golden fixtures.  GPU only: `pytest -m gpu`.
"""

import os

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import psmf_oracle as O
import rpsmf_amd as psmf
from rpsmf_amd import nonlinearities as NL

# what runs the configurations no specialised kernel takes at 17 <= r <= 32 (PSMF_FILTER7=0: the one-group general kernel)
GENERAL_17_32 = "psmf_blk_filter" if os.environ.get("PSMF_FILTER7") == "0" else "psmf_blk_filter7"

pytestmark = pytest.mark.gpu


def _capi():
    from rpsmf_amd import _capi

    return _capi


def ydict(Y):
    return {k + 1: Y[k][:, None].copy() for k in range(Y.shape[0])}


def _problem(d, r, T, seed):
    rng = np.random.default_rng(seed)
    Ct = rng.standard_normal((d, r))
    x = rng.standard_normal(r)
    Y = np.empty((T, d))
    for t in range(T):
        x = 0.9 * np.sin(x + 0.3) + 0.1 * rng.standard_normal(r)
        Y[t] = Ct @ x + 0.3 * rng.standard_normal(d)
    return Y, 0.1 * rng.standard_normal((d, r))


def _theta_for(nl, rng, r):
    """A theta in the regime the experiments use (beijing_psmf.py:117: 0.1 * rand): matrices near a contraction."""
    th = 0.1 * rng.random(nl.n_params)
    if isinstance(nl, (NL.ScaledWalk,)) or getattr(nl, "scaled", False):
        th[:r * r] = (0.8 * np.eye(r) + 0.05 * rng.standard_normal((r, r))).reshape(-1)
    if isinstance(nl, NL.FourierBasis):
        for t in range(2 * nl.N):
            th[t * r * r:(t + 1) * r * r] = (0.5 * np.eye(r) + 0.05 * rng.standard_normal((r, r))).reshape(-1) / nl.N
    return th


KINDS = [
    ("scaled_walk_bias", lambda r: NL.ScaledWalk(r, bias=True), 6),
    ("scaled_walk", lambda r: NL.ScaledWalk(r, bias=False), 17),
    ("sinusoid", lambda r: NL.Sinusoid(r), 8),
    ("sinusoid_unscaled", lambda r: NL.Sinusoid(r, scaled=False), 20),
    ("sinusoid_unphased", lambda r: NL.Sinusoid(r, phased=False), 5),
    ("sinusoid_plain", lambda r: NL.Sinusoid(r, scaled=False, phased=False), 32),
    ("fourier1", lambda r: NL.FourierBasis(r, N=1), 7),
    ("fourier3", lambda r: NL.FourierBasis(r, N=3), 4),
    ("cos_phase", lambda r: NL.CosPhase(r), 9),
    ("fourier1_r20", lambda r: NL.FourierBasis(r, N=1), 20),      # dense Jacobians beyond the small-rank kernel: psmf_blk_filter<32>,
    ("sinusoid_r27", lambda r: NL.Sinusoid(r), 27),               #   F P F^T on the matrix cores (odd rank: the zero padding of the images)
    ("scaled_walk_bias_r32", lambda r: NL.ScaledWalk(r, bias=True), 32),
]


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("name,make,r", KINDS, ids=[k[0] for k in KINDS])
def test_device_dynamics_vs_oracle(name, make, r, robust):
    """mu_bar = f(theta, mu, k), P_bar = F P F^T + Q with the analytic F, and gradsum = sum_k J_theta^T g_f, all inside the
    blocked engine's time loop, against the oracle run on the same callable (complex-step F and J_theta)."""
    c = _capi()
    d, T = 300, 90
    nl = make(r)
    rng = np.random.default_rng(11 + r)
    Y, C0 = _problem(d, r, T, 50 + r)
    theta = _theta_for(nl, rng, r)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mu0 = 0.2 * rng.standard_normal(r)
    mode = O.Mode(robust=robust)
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    st, Yp, _ = O.run_epoch(st, Y, mode, O.CallableDyn(nl, nl.n_params))
    f = c.DeviceFilter(d, r, robust=robust, storage="f64", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags,
                       dyn_terms=nl.device_terms)
    assert f.geometry()["engine"] == "block" and f.n_theta == nl.n_params
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, mu0, rho=1.0, lambda0=1.8, theta=theta)
    # diagonal-Jacobian kinds (cos-phase, unscaled sinusoid) with Q = q I: the role-specialised filter4; dense Jacobians: the
    # general blocked kernel
    diag = name in ("cos_phase", "sinusoid_unscaled", "sinusoid_plain")
    assert f.geometry()["filter_kernel"].replace("filter6d", "filter6") == ("psmf_blk_filter6" if r <= 16 else
                                             (("psmf_blk_filter4" if r > 16 else "psmf_blk_filter4s") if diag else GENERAL_17_32))
    f.zero_gradsum()
    f.run(0, T)
    s = f.get_state()
    for k in ("C", "V", "mu", "P"):
        assert relerr(s[k], getattr(st, k)) < 1e-8, k
    assert relerr(f.y_pred(0, T), Yp) < 1e-8
    assert relerr(s["gradsum"], st.gradsum) < 1e-7
    # roll-out with the same f (psmf.py:182-188)
    assert relerr(f.predict(T, 5), O.predict_rollout(st.C, st.mu, theta, O.CallableDyn(nl, nl.n_params), T, 5)) < 1e-8
    f.close()


def test_fourier_beijing_configuration_on_device():
    """ExperimentBeijing's periodic configuration (FourierBasis(rank=1, N=1), full filter, Adam between epochs:
    beijing_psmf.py:97-140) with the default back end, against the reference's own run of PSMFIter; then the experiment's
    subclass (V re-initialised every epoch, `_mu` kept: beijing_psmf.py:82-88) against the numpy back end."""
    g = load_golden("psmf_full_fourier")
    T, n_pred, n_iter = int(g["T"]), int(g["n_pred"]), int(g["n_iter"])
    Y = g["Y"]
    d, r = g["C0"].shape

    class Beijing(psmf.PSMFIter):
        def step_reset(self):
            super().step_reset()
            self._V = {0: self.V0}

        def _prune(self, k):
            del self._C[k - 1], self._V[k - 1], self._P[k - 1]

    def run(cls, **kw):
        f = cls(g["theta0"].reshape(-1, 1), g["C0"], g["V0"], g["mu0"].reshape(-1, 1), g["P0"],
                {k: g["Q"] for k in range(T + 1)}, {k: np.eye(d) for k in range(T + 1)}, psmf.FourierBasis(rank=r, N=1), **kw)
        f.adam_init(gam=1e-3)
        for i in range(1, n_iter + 1):
            f.step(ydict(Y[:T]), i, T)
            f.predict(i, T, n_pred)
            f.adam_update(i, project=True)
        theta = np.array([f._theta[i].reshape(-1) for i in range(n_iter + 1)])
        yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
        return f, theta, yp

    f, theta, yp = run(psmf.PSMFIter, storage="f64")
    assert f._dev.geometry()["engine"] == "block" and f._dev.dyn_kind == _capi().DYN_FOURIER
    assert relerr(theta, g["theta"]) < 1e-7 and relerr(yp, g["y_pred_last"]) < 1e-7
    fb, theta_b, yp_b = run(Beijing, storage="f64")
    fn, theta_n, yp_n = run(Beijing, backend="numpy")
    assert relerr(theta_b, theta_n) < 1e-9 and relerr(yp_b, yp_n) < 1e-9
    mu_b = np.array([fb._mu[k].reshape(-1) for k in range(T + 1)])
    mu_n = np.array([fn._mu[k].reshape(-1) for k in range(T + 1)])
    assert relerr(mu_b, mu_n) < 1e-9


def tanh_mix(theta, x, t):
    """an arbitrary user callable: not one of the device kinds"""
    r = np.asarray(x).size
    A = np.asarray(theta)[:r * r].reshape(r, r)
    return np.tanh(A @ np.asarray(x).reshape(r, 1)) + 0.01 * np.asarray(theta)[r * r:].reshape(r, 1) * np.cos(0.1 * t)


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_host_stepped_arbitrary_callable(robust):
    """A plain Python callable with backend='hip': f, F, J_theta on the host (r-sized), everything d-sized on the device, one
    psmf_step_host per timestep -- against the oracle on the same callable; two epochs with Adam in between."""
    d, r, T, n_pred = 2000, 5, 60, 4
    rng = np.random.default_rng(3)
    Y, C0 = _problem(d, r, T + n_pred, 7)
    theta0 = np.concatenate([(0.7 * np.eye(r) + 0.05 * rng.standard_normal((r, r))).reshape(-1), rng.random(r)]).reshape(-1, 1)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mu0 = np.zeros((r, 1))
    if robust:
        f = psmf.rPSMFIter(theta0, C0, V0, mu0, P0, Q, 1.0, 1.8, tanh_mix, storage="f64")
    else:
        f = psmf.PSMFIter(theta0, C0, V0, mu0, P0, {k: Q for k in range(T + 1)}, {k: 1.0 for k in range(T + 1)}, tanh_mix, storage="f64")
    f.adam_init(gam=1e-3)
    dyn = O.CallableDyn(tanh_mix, theta0.size)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta0.reshape(-1).copy())
    m = v = np.zeros(theta0.size)
    for i in (1, 2):
        f.step(ydict(Y[:T]), i, T)
        f.predict(i, T, n_pred)
        st.gradsum = np.zeros(theta0.size)
        if robust:
            st.Q, st.rho, st.lam = Q, 1.0, 1.8
        st, Yp, _ = O.run_epoch(st, Y[:T], O.Mode(robust=robust), dyn)
        assert f._dev.dyn_kind == _capi().DYN_HOST and f._dev.geometry()["engine"] == "step"
        assert relerr(f._C[T], st.C) < 1e-8 and relerr(f._P[T], st.P) < 1e-8 and relerr(f._mu[T], st.mu.reshape(-1, 1)) < 1e-8
        assert relerr(f._gradsum.reshape(-1), st.gradsum) < 1e-7
        yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
        ref = np.vstack([Yp, O.predict_rollout(st.C, st.mu, st.theta, dyn, T, n_pred)])
        assert relerr(yp, ref) < 1e-8
        f.adam_update(i)
        st.theta, m, v = O.adam_update(st.theta, st.gradsum, m, v, i)
        assert relerr(f._theta[i].reshape(-1), st.theta) < 1e-7


def test_general_kind_beyond_r32_runs_in_the_device_loop():
    """Sinusoid at r = 40 through the drop-in class: the blocked engine (r <= 32) cannot take it, the per-step engine's serial stage
    evaluates it (round 5; it was host-stepped before)."""
    d, r, T = 500, 40, 25
    rng = np.random.default_rng(5)
    Y, C0 = _problem(d, r, T, 9)
    nl = NL.Sinusoid(r, scaled=False)
    theta0 = (0.1 * rng.random(nl.n_params)).reshape(-1, 1)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    f = psmf.PSMFIter(theta0, C0, V0, np.zeros((r, 1)), P0, {k: Q for k in range(T + 1)}, {k: 1.0 for k in range(T + 1)}, nl, storage="f64")
    f.adam_init()
    f.step(ydict(Y), 1, T)
    assert f._dev.dyn_kind == _capi().DYN_SINUSOID and f._dev.geometry()["filter_kernel"] == "psmf_sweep_solve"
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=0.0, theta=theta0.reshape(-1).copy(), gradsum=np.zeros(nl.n_params))
    st, _, _ = O.run_epoch(st, Y, O.Mode(), O.CallableDyn(nl, nl.n_params))
    assert relerr(f._C[T], st.C) < 1e-8 and relerr(f._gradsum.reshape(-1), st.gradsum) < 1e-7


# the kinds with a matrix in them on the PER-STEP engine (r > 32, engine = 1, a non-uniform R): psmf_dyn.hip inside psmf_serial /
# psmf_serial_wide -- mu_bar, F, P_bar = F P F^T + Q as two r x r x r products, gradsum += J_theta^T g_f
STEP_KINDS = [
    ("scaled_walk_bias", lambda r: NL.ScaledWalk(r, bias=True), 6, False),
    ("sinusoid", lambda r: NL.Sinusoid(r), 12, False),
    ("sinusoid_unscaled", lambda r: NL.Sinusoid(r, scaled=False), 20, False),
    ("sinusoid_unphased_r27", lambda r: NL.Sinusoid(r, phased=False), 27, False),
    ("fourier3", lambda r: NL.FourierBasis(r, N=3), 4, False),
    ("fourier1_r32", lambda r: NL.FourierBasis(r, N=1), 32, False),
    ("fourier2_r40", lambda r: NL.FourierBasis(r, N=2), 40, False),
    ("scaled_walk_r47", lambda r: NL.ScaledWalk(r, bias=False), 47, False),
    ("sinusoid_r64", lambda r: NL.Sinusoid(r), 64, False),
    ("sinusoid_nonuniform_R", lambda r: NL.Sinusoid(r), 9, True),
    ("fourier1_r40_nonuniform_R", lambda r: NL.FourierBasis(r, N=1), 40, True),
]


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("name,make,r,rows", STEP_KINDS, ids=[k[0] for k in STEP_KINDS])
def test_device_dynamics_on_the_per_step_engine(name, make, r, rows, robust):
    c = _capi()
    d, T = 300, (60 if r < 40 else 24)          # (the oracle's complex-step Jacobians of a dense kind at r >= 40 are seconds per 10 timesteps)
    nl = make(r)
    rng = np.random.default_rng(311 + r)
    Y, C0 = _problem(d, r, T, 150 + r)
    theta = _theta_for(nl, rng, r)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mu0 = 0.2 * rng.standard_normal(r)
    rho = 0.3 + 2.0 * rng.random(d) if rows else 1.0
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=rho, lam=1.8, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=robust), O.CallableDyn(nl, nl.n_params))
    f = c.DeviceFilter(d, r, robust=robust, storage="f64", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags,
                       dyn_terms=nl.device_terms, engine="step", nonuniform_R=rows)
    if rows:
        f.set_row_noise(rho)
    assert f.geometry()["engine"] == "step" and f.n_theta == nl.n_params
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, mu0, rho=1.0, lambda0=1.8, theta=theta)
    assert f.geometry()["filter_kernel"] == "psmf_sweep_solve"
    f.zero_gradsum()
    f.run(0, T // 2)                   # two runs: the serial stage's first = 1 preparation in the middle of a series too
    f.run(T // 2, T)
    s = f.get_state()
    for k in ("C", "V", "mu", "P"):
        assert relerr(s[k], getattr(st, k)) < 1e-8, (name, k, relerr(s[k], getattr(st, k)))
    assert relerr(f.y_pred(0, T), Yp) < 1e-8
    assert relerr(s["gradsum"], st.gradsum) < 1e-7, (name, relerr(s["gradsum"], st.gradsum))
    assert relerr(f.predict(T, 5), O.predict_rollout(st.C, st.mu, theta, O.CallableDyn(nl, nl.n_params), T, 5)) < 1e-8
    f.close()


@pytest.mark.parametrize("optimiser", ["adam", "sgd"])
def test_recursive_general_kind_per_step_engine_matches_the_blocked_engine(optimiser):
    """The in-loop optimiser with a dense-Jacobian kind (psmf.py:287-304 with nonlinearities.py:81-114): the per-step engine's
    serial stage against the blocked engine's general kernel (itself pinned to the reference's run by the cos-phase fixtures and
    to the oracle's gradients above) -- theta after every update enters the next steps, so any difference grows."""
    c = _capi()
    d, r, T = 300, 12, 80
    nl = NL.Sinusoid(r)
    rng = np.random.default_rng(8)
    Y, C0 = _problem(d, r, T, 61)
    theta = _theta_for(nl, rng, r)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    out = {}
    for engine in ("block", "step"):
        f = c.DeviceFilter(d, r, storage="f64", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags, dyn_terms=nl.device_terms,
                           recursive=(2 if optimiser == "sgd" else 1), update_every=7, adam_lr=(1e-7 if optimiser == "sgd" else 1e-3),
                           engine=engine)                 # (SGD steps by lr x the raw gradient sum, which grows with d)
        f.upload_series(Y)
        f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=0.0, theta=theta)
        f.zero_gradsum()
        if optimiser == "adam":
            f.set_adam(np.zeros(nl.n_params), np.zeros(nl.n_params))
        f.run(0, T)
        out[engine] = f.get_state()
        out[engine]["yp"] = f.y_pred(0, T)
        f.close()
    assert np.max(np.abs(out["block"]["theta"] - theta)) > 1e-6          # the optimiser did move theta
    for k in ("theta", "C", "V", "mu", "P", "yp"):
        assert relerr(out["step"][k], out["block"][k]) < 1e-9, (k, relerr(out["step"][k], out["block"][k]))


@pytest.mark.parametrize("engine", ["block", "step"])
@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_recursive_adam_both_engines(robust, engine):
    """PSMFRecursive / rPSMFRecursive (theta updated by Adam inside the time loop, psmf.py:287-304) in the blocked engine
    (in-loop optimiser inside the block kernel) and in the per-step engine, against the reference's run."""
    c = _capi()
    g = load_golden("rpsmf_recursive" if robust else "psmf_recursive")
    T, n_pred, ue = int(g["T"]), int(g["n_pred"]), int(g["update_every"])
    d, r = g["C0"].shape
    f = c.DeviceFilter(d, r, storage="f64", robust=robust, dyn_kind=c.DYN_COS_PHASE, recursive=True, update_every=ue, adam_lr=1e-3,
                       engine=engine)
    assert f.geometry()["engine"] == engine
    f.upload_series(g["Y"])
    f.set_state(g["C0"], g["V0"], g["P0"], g["Q"], g["mu0"], rho=float(g["rho"]), lambda0=float(g["lambda0"]), theta=g["theta0"])
    f.zero_gradsum()
    f.set_adam(np.zeros(r), np.zeros(r))
    f.run(0, T)
    s = f.get_state()
    assert relerr(s["theta"], g["theta"][-1]) < 1e-6
    assert relerr(s["C"], g["C_T"]) < 1e-6 and relerr(s["mu"], g["mu_T"]) < 1e-6 and relerr(s["P"], g["P_T"]) < 1e-6
    yp = np.vstack([f.y_pred(0, T), f.predict(T, n_pred)])
    assert relerr(yp, g["y_pred"]) < 1e-6
    f.close()


def test_recursive_sinusoid_vs_oracle():
    """In-loop Adam with a dense-Jacobian kind and update_every = 3: theta (r^2 + 2 r parameters), its moments and the
    gradient restart all live in the block kernel's loop; oracle = step-by-step recursion + adam_update."""
    c = _capi()
    d, r, T, ue = 200, 6, 50, 3
    nl = NL.Sinusoid(r)
    rng = np.random.default_rng(2)
    Y, C0 = _problem(d, r, T, 13)
    theta = _theta_for(nl, rng, r)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    dyn = O.CallableDyn(nl, nl.n_params)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=0.0, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    m = v = np.zeros(nl.n_params)
    for k in range(1, T + 1):
        st, _ = O.lowrank_step(st, Y[k - 1], k, O.Mode(), dyn)
        if k % ue == 0:
            st.theta, m, v = O.adam_update(st.theta, st.gradsum, m, v, k, lr=2e-3)
            st.gradsum = np.zeros(nl.n_params)
    f = c.DeviceFilter(d, r, storage="f64", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags, recursive=True, update_every=ue, adam_lr=2e-3)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=0.0, theta=theta)
    f.zero_gradsum()
    f.set_adam(np.zeros(nl.n_params), np.zeros(nl.n_params))
    f.run(0, T)
    s = f.get_state()
    assert relerr(s["theta"], st.theta) < 1e-8 and relerr(s["C"], st.C) < 1e-8 and relerr(s["P"], st.P) < 1e-8
    f.close()


@pytest.mark.parametrize("engine", ["block", "step"])
@pytest.mark.parametrize("r,iso", [(12, False), (12, True), (20, True), (32, True)])
def test_per_step_R_and_Q_schedules(engine, r, iso):
    """PSMFIter reads R[k], Q[k] of the step (psmf.py:115,123,141): R_k = rho_k I, Q_k = q_k Q as device schedules, through
    the class surface, against the oracle with the same dictionaries.  iso: Q = q I, the usual case -- where the blocked
    engine would otherwise pick the two-inversion kernels (filter3 / filter3s / filter2), which read rho and q once per
    block; with schedules the library must select the general kernel."""
    d, T = 700, 75
    Y, C0 = _problem(d, r, T, 21)
    rng = np.random.default_rng(8)
    Q0 = 0.1 * np.eye(r) + (0.0 if iso else 0.01) * np.ones((r, r))
    rho_k = 0.5 + rng.random(T + 1)
    q_k = 0.5 + rng.random(T + 1)
    q_k[1] = 1.0
    Qs = {k: q_k[k] * Q0 for k in range(T + 1)}
    Rs = {k: rho_k[k] for k in range(T + 1)}
    V0, P0 = 0.1 * np.eye(r), np.eye(r)
    f = psmf.PSMFIter(np.zeros((0, 1)), C0, V0, np.zeros((r, 1)), P0, Qs, Rs, psmf.RandomWalk(), storage="f64", engine=engine)
    f.optim_init()
    f.step(ydict(Y), 1, T)
    assert f._dev.geometry()["engine"] == engine
    if engine == "block":      # schedules keep the run off filter3 (rho, q read once per block): filter4 reads them per step
        assert f._dev.geometry()["filter_kernel"] == ("psmf_blk_filter6" if r <= 16 else
                                                      (("psmf_blk_filter4" if r > 16 else "psmf_blk_filter4s") if iso else GENERAL_17_32))
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q0, rho=1.0, lam=0.0)
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(), O.RandomWalkDyn(), Qs=lambda k: Qs[k], rhos=lambda k: Rs[k])
    assert relerr(f._C[T], st.C) < 1e-9 and relerr(f._V[T], st.V) < 1e-9 and relerr(f._P[T], st.P) < 1e-9
    yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + 1)])
    assert relerr(yp, Yp) < 1e-9


@pytest.mark.parametrize("where", ["device", "host"])
def test_arbitrary_Q_schedule(where, monkeypatch):
    """A Q[k] that is not a scalar multiple of Q[1] (psmf.py:115 reads a matrix per step): uploaded matrix by matrix, P_bar = P + Q_k
    formed in the per-step engine's serial stage (psmf_set_q_matrix_schedule); beyond the upload limit the host forms P_bar
    (psmf_step_host), as before round 5."""
    d, r, T = 400, 4, 30
    Y, C0 = _problem(d, r, T, 23)
    rng = np.random.default_rng(9)
    Qs = {}
    for k in range(T + 1):
        A = rng.standard_normal((r, r))
        Qs[k] = 0.05 * np.eye(r) + 0.01 * A @ A.T
    V0, P0 = 0.1 * np.eye(r), np.eye(r)
    if where == "host":
        monkeypatch.setattr(psmf.PSMFIter, "_Q_MATRIX_SCHEDULE_MAX_BYTES", 0)
    f = psmf.PSMFIter(np.zeros((0, 1)), C0, V0, np.zeros((r, 1)), P0, Qs, {k: 2.0 for k in range(T + 1)}, psmf.RandomWalk(), storage="f64")
    f.optim_init()
    f.step(ydict(Y), 1, T)
    if where == "host":
        assert f._dev.dyn_kind == _capi().DYN_HOST
    else:
        assert f._dev.dyn_kind == _capi().DYN_RANDOM_WALK and f._dev.geometry()["filter_kernel"] == "psmf_sweep_solve"
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Qs[1], rho=2.0, lam=0.0)
    st, _, _ = O.run_epoch(st, Y, O.Mode(), O.RandomWalkDyn(), Qs=lambda k: Qs[k])
    assert relerr(f._C[T], st.C) < 1e-9 and relerr(f._P[T], st.P) < 1e-9


def test_tracking_error_norms_on_device():
    """TrackingMixin.errors_update (tracking.py:63-76): the three Frobenius norms of Y_pred - Y (full / train / pred windows)
    reduced on the device, against the host computation from `_y_pred`."""
    from rpsmf_amd.tracking import TrackingMixin

    class Tracked(TrackingMixin, psmf.PSMFIter):
        pass

    d, r, T, n_pred = 1200, 6, 80, 20
    Y, C0 = _problem(d, r, T + n_pred, 31)
    y_full = ydict(Y)
    y_train = {k: y_full[k] for k in range(1, T + 1)}
    nl = psmf.CosPhase(r)
    theta0 = (1e-3 * np.arange(1, r + 1)).reshape(-1, 1)
    f = Tracked(theta0, C0, 0.1 * np.eye(r), np.zeros((r, 1)), np.eye(r), {k: 0.1 * np.eye(r) for k in range(T + 1)},
                {k: 1.0 for k in range(T + 1)}, nl, storage="f64")
    f.adam_init()
    f.errors_init(y_full, T, 2, n_pred)
    for i in (1, 2):
        f.step(y_train, i, T)
        f.predict(i, T, n_pred)
        f.adam_update(i)
        f.errors_update(i, y_full, T, n_pred)
        f.log(i, 2, 0.0, verbose=False)
        Yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])
        assert relerr(f._E_y[i], np.linalg.norm(Yp - Y)) < 1e-10
        assert relerr(f._E_train[i], np.linalg.norm(Yp[:T] - Y[:T])) < 1e-10
        assert relerr(f._E_pred[i], np.linalg.norm(Yp[T:] - Y[T:])) < 1e-10
    assert f._tracking_on_device == 2 and len(f._logs) == 2


@pytest.mark.parametrize("engine", ["step", "block"])
@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_recursive_sgd_in_the_device_loop(robust, engine):
    """SGD inside the recursive loop (psmf.py:244-248,299-304) runs in the device loops since round 5 (psmf_config.recursive = 2:
    the persistent per-step kernel, the launched serial stage and the blocked kernels that share dyn_adam_step) -- same numbers as the
    numpy back end, which executes the reference's hook sequence."""
    g = load_golden("rpsmf_recursive" if robust else "psmf_recursive")
    T, n_pred, ue = int(g["T"]), int(g["n_pred"]), int(g["update_every"])
    d, r = g["C0"].shape
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    out = []
    for kw in (dict(storage="f64", engine=engine), dict(backend="numpy")):
        nl = psmf.CosPhase(r)
        if robust:
            f = psmf.rPSMFRecursive(theta0, g["C0"], g["V0"], mu0, g["P0"], g["Q"], np.eye(d), 1.8, nl, optim="sgd", **kw)
        else:
            f = psmf.PSMFRecursive(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: g["Q"] for k in range(T + 1)},
                                   {k: np.eye(d) for k in range(T + 1)}, nl, optim="sgd", **kw)
        f._update_every = ue                 # run() with a small step size: plain SGD at the default 1e-3 moves theta by O(1)
        f.optim_init(gam=1e-7)               # per observation here (gradients of O(10^3)) and the comparison turns chaotic
        f.step(ydict(g["Y"]), T)
        f.predict(T, n_pred)
        out.append((f._theta[T].reshape(-1), f._C[T], np.array([f._y_pred[k].reshape(-1) for k in range(1, T + n_pred + 1)])))
        if "storage" in kw:
            assert f._dev.dyn_kind == _capi().DYN_COS_PHASE          # evaluated on the device, theta stepped there
            kern = f._dev.geometry()["filter_kernel"]
            assert kern == ("psmf_pstep_k" if engine == "step" else kern) and kern not in ("psmf_blk_filter4", "psmf_blk_filter4s", "psmf_blk_filter5")
    for a, b in zip(*out):
        assert relerr(a, b) < 1e-9


def test_recursive_custom_learning_rate_schedule_stays_host_stepped():
    """An in-loop optimiser with a learning-rate schedule the device does not implement (here: a step function of k) keeps theta and the
    optimiser on the host, the device advances one step at a time -- same numbers as the numpy back end."""
    from rpsmf_amd.learning_rate import BaseLearningRate

    class Steps(BaseLearningRate):
        def get(self, t):
            return 1e-7 if t < 20 else 5e-8

    g = load_golden("psmf_recursive")
    T, n_pred, ue = int(g["T"]), int(g["n_pred"]), int(g["update_every"])
    d, r = g["C0"].shape
    theta0, mu0 = g["theta0"].reshape(-1, 1), g["mu0"].reshape(-1, 1)
    out = []
    for kw in (dict(storage="f64"), dict(backend="numpy")):
        f = psmf.PSMFRecursive(theta0, g["C0"], g["V0"], mu0, g["P0"], {k: g["Q"] for k in range(T + 1)},
                               {k: np.eye(d) for k in range(T + 1)}, psmf.CosPhase(r), optim="sgd", **kw)
        f._update_every = ue
        f.optim_init(gam=Steps())
        f.step(ydict(g["Y"]), T)
        out.append((f._theta[T].reshape(-1), f._C[T]))
        if "storage" in kw:
            assert f._dev.dyn_kind == _capi().DYN_HOST
    for a, b in zip(*out):
        assert relerr(a, b) < 1e-9


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_nonuniform_diagonal_R_on_device(robust):
    """R = diag(rho_i) with different rho_i (psmf.py:140-153 takes any diagonal R): row weights 1 / (rho_i + s_k) change with every
    step, so the per-step engine recomputes the weighted Gram each step and the sweep carries the weighted sums; through the
    class surface (a (d,) vector where the reference wants a dense d x d matrix), against the oracle."""
    d, r, T = 1500, 7, 70
    Y, C0 = _problem(d, r, T, 41)
    rng = np.random.default_rng(12)
    rho = 0.3 + 2.0 * rng.random(d)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    if robust:
        f = psmf.rPSMFIter(np.zeros((0, 1)), C0, V0, np.zeros((r, 1)), P0, Q, rho, 1.8, psmf.RandomWalk(), storage="f64")
    else:
        f = psmf.PSMFIter(np.zeros((0, 1)), C0, V0, np.zeros((r, 1)), P0, {k: Q for k in range(T + 1)}, {k: rho for k in range(T + 1)},
                          psmf.RandomWalk(), storage="f64")
    f.optim_init()
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=rho.copy(), lam=1.8)
    for i in (1, 2):
        f.step(ydict(Y), i, T)
        assert f._dev.geometry()["engine"] == "step"
        if robust:
            st.Q, st.rho, st.lam = Q, rho.copy(), 1.8
        st, Yp, _ = O.run_epoch(st, Y, O.Mode(robust=robust), O.RandomWalkDyn(), want_grad=False)
        assert relerr(f._C[T], st.C) < 1e-9 and relerr(f._V[T], st.V) < 1e-9 and relerr(f._P[T], st.P) < 1e-9
        assert relerr(f._mu[T], st.mu.reshape(-1, 1)) < 1e-9
        yp = np.array([f._y_pred[k].reshape(-1) for k in range(1, T + 1)])
        assert relerr(yp, Yp) < 1e-9
        f.optim_update(i)          # theta is empty; creates _theta[i] for the next epoch as the reference's run() does


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("kind", ["cos_phase", "sinusoid_unscaled"])
@pytest.mark.parametrize("recursive", [False, True], ids=["epoch", "recursive"])
def test_simplified_hooks_kernel_vs_oracle(kind, robust, recursive):
    """psmf_blk_filter5: the simplified hook configuration (synthetic_psmf.py:78-100, synthetic_rpsmf.py:82-118) with a
    diagonal-Jacobian f, PSMF and rPSMF, with the gradient summed over the epoch or Adam stepping theta inside the loop
    (update_every = 2) -- against the oracle on the same callable (complex-step derivatives), float64 storage."""
    c = _capi()
    d, r, T, ue = 500, 12, 110, 2
    nl = NL.CosPhase(r) if kind == "cos_phase" else NL.Sinusoid(r, scaled=False)
    rng = np.random.default_rng(31 + r)
    Y, C0 = _problem(d, r, T, 77)
    theta = 0.1 * rng.random(nl.n_params)
    V0, P0, Q = 0.1 * np.eye(r), 0.3 * np.eye(r), 0.05 * np.eye(r)
    mu0 = 0.2 * rng.standard_normal(r)
    mode = O.Mode(robust=robust, coef_update=False, eta_full=False, pbar_predict=False)
    dyn = O.CallableDyn(nl, nl.n_params)
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    Yp = np.empty((T, d))
    m = v = np.zeros(nl.n_params)
    for k in range(1, T + 1):
        st, info = O.lowrank_step(st, Y[k - 1], k, mode, dyn)
        Yp[k - 1] = info.y_pred
        if recursive and k % ue == 0:
            st.theta, m, v = O.adam_update(st.theta, st.gradsum, m, v, k)
            st.gradsum = np.zeros(nl.n_params)
    kw = dict(recursive=True, update_every=ue, adam_lr=1e-3) if recursive else {}
    f = c.DeviceFilter(d, r, robust=robust, storage="f64", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags, dyn_terms=nl.device_terms,
                       coef_update=False, eta_full=False, pbar_predict=False, **kw)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, mu0, rho=1.0, lambda0=1.8, theta=theta)
    assert f.geometry()["filter_kernel"] == "psmf_blk_filter5"
    f.zero_gradsum()
    if recursive:
        f.set_adam(np.zeros(nl.n_params), np.zeros(nl.n_params))
    f.run(0, 60)
    f.run(60, T)                      # carried state, a second chain
    s = f.get_state()
    for name in ("C", "V", "mu", "P"):
        assert relerr(s[name], getattr(st, name)) < 1e-8, name
    assert relerr(f.y_pred(0, T), Yp) < 1e-8
    assert relerr(s["theta"], st.theta) < 1e-8
    if st.gradsum is not None and np.max(np.abs(st.gradsum)) > 0:
        assert relerr(s["gradsum"], st.gradsum) < 1e-7
    if robust:
        assert relerr(s["rho"], st.rho) < 1e-9 and relerr(s["lam"], st.lam) < 1e-12
    f.close()


def test_q_matrix_schedule_refusals_and_dense_dynamics_with_it():
    """psmf_set_q_matrix_schedule: refused where it cannot hold (blocked engine, rPSMF's own running Q), range-checked by psmf_run,
    dropped again by None; and together with a dense-Jacobian kind (P_bar = F P F^T + Q_k, both in the serial stage) against the oracle."""
    c = _capi()
    d, r, T = 350, 5, 24
    Y, C0 = _problem(d, r, T, 31)
    rng = np.random.default_rng(4)
    Qm = np.empty((T + 1, r, r))
    for k in range(T + 1):
        A = rng.standard_normal((r, r))
        Qm[k] = 0.05 * np.eye(r) + 0.01 * A @ A.T
    f = c.DeviceFilter(d, r, storage="f64", engine="block")
    with pytest.raises(ValueError):
        f.set_q_matrix_schedule(Qm)
    f.close()
    f = c.DeviceFilter(d, r, storage="f64", engine="step", robust=True)
    with pytest.raises(ValueError):
        f.set_q_matrix_schedule(Qm)
    f.close()
    nl = NL.Sinusoid(r)
    theta = _theta_for(nl, rng, r)
    V0, P0 = 0.1 * np.eye(r), np.eye(r)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Qm[1], rho=1.0, lam=0.0, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    st, Yp, _ = O.run_epoch(st, Y, O.Mode(), O.CallableDyn(nl, nl.n_params), Qs=lambda k: Qm[k])
    f = c.DeviceFilter(d, r, storage="f64", engine="step", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags, dyn_terms=nl.device_terms)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Qm[1], np.zeros(r), rho=1.0, lambda0=0.0, theta=theta)
    f.set_q_matrix_schedule(Qm[:T])                      # one matrix short: the last step is beyond the schedule
    with pytest.raises(ValueError):
        f.run(0, T)
    f.set_q_matrix_schedule(Qm)
    f.set_state(C0, V0, P0, Qm[1], np.zeros(r), rho=1.0, lambda0=0.0, theta=theta)
    f.zero_gradsum()
    f.run(0, T)
    s = f.get_state()
    for n in ("C", "V", "mu", "P"):
        assert relerr(s[n], getattr(st, n)) < 1e-8, n
    assert relerr(f.y_pred(0, T), Yp) < 1e-8 and relerr(s["gradsum"], st.gradsum) < 1e-7
    f.set_q_matrix_schedule(None)                        # back to the constant Q of the state
    f.close()
