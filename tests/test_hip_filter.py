"""Parity of the HIP large-d engine (through the C ABI) with the CPU oracle and the golden
fixtures.  GPU only: `pytest -m gpu`.

Tolerances: storage f64 -> 1e-9 relative (pure float64 arithmetic on both sides, different
summation order + algebraically tracked Gram); storage f32 -> 1e-5 relative, the bar
BASELINE.json's north_star states.
"""

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import psmf_oracle as O

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-9, "f32": 1e-5}


def _capi():
    from rpsmf_amd import _capi

    return _capi


@pytest.fixture(params=["step", "block"])
def engine(request):
    """Both device engines: one row sweep per timestep, and the exact time-blocked engine."""
    return request.param


def _mode_kwargs(mode):
    return dict(robust=mode.robust, coef_update=mode.coef_update, eta_full=mode.eta_full,
                pbar_predict=mode.pbar_predict, fixed_lambda=mode.fixed_lambda, alpha=mode.alpha, beta=mode.beta)


def _compare(dev_state, st, tol, what=("C", "V", "mu", "P")):
    for name in what:
        ref = getattr(st, name)
        if np.max(np.abs(ref)) == 0.0:
            assert np.max(np.abs(dev_state[name])) == 0.0, name
        else:
            assert relerr(dev_state[name], ref) < tol, (name, relerr(dev_state[name], ref))


@pytest.mark.parametrize("storage", ["f64", "f32"])
@pytest.mark.parametrize("name,robust", [("psmf_full_rw", False), ("rpsmf_full_rw", True)])
def test_golden_full_filter(name, robust, storage, engine):
    """PSMFIter / rPSMFIter as shipped (d=20, r=5, T=200, two epochs) against the reference's outputs."""
    c = _capi()
    g = load_golden(name)
    Y = g["Y"]
    T, d = Y.shape
    r = g["C0"].shape[1]
    f = c.DeviceFilter(d, r, robust=robust, storage=storage, engine=engine)
    assert f.geometry()["engine"] == engine
    f.upload_series(Y)
    f.set_state(g["C0"], g["V0"], g["P0"], g["Q"], g["mu0"], rho=float(g["rho"]), lambda0=float(g["lambda0"]))
    tol = TOL[storage]
    for ep in (1, 2):
        if ep == 2 and robust:
            f.set_state(Q=g["Q"], rho=float(g["rho"]), lambda0=float(g["lambda0"]))
        k_prev = 0
        for k in (1, 2, 10, 200):
            f.run(k_prev, k)
            k_prev = k
            s = f.get_state()
            p = f"s_e{ep}_k{k}_"
            assert relerr(s["C"], g[p + "C"]) < tol
            assert relerr(s["V"], g[p + "V"]) < tol
            assert relerr(s["mu"], g[p + "mu"].reshape(-1)) < tol
            assert relerr(s["P"], g[p + "P"]) < tol
            assert relerr(s["eta"], g[p + "eta"]) < tol
            assert relerr(s["N"], g[p + "N"]) < tol
            if robust:
                assert relerr(s["lam"], g[p + "lam"]) < tol
                assert relerr(s["rho"], g[p + "rho"]) < tol
                assert relerr(s["Q"], g[p + "Q"]) < tol
    assert relerr(f.y_pred(0, T), g["y_pred_e2"]) < tol
    f.close()


def _problem(d, r, T, seed, noise="normal"):
    Y = O.synthetic_series(d, r, T, seed, noise=noise, dtype=np.float64)
    rng = np.random.default_rng(seed + 1)
    return Y, 0.1 * rng.standard_normal((d, r))


@pytest.mark.parametrize("storage", ["f64", "f32"])
@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("d,r,T", [(3000, 20, 150), (777, 7, 60), (4096, 32, 80), (260, 64, 40), (50, 1, 30)])
def test_full_filter_vs_oracle(d, r, T, robust, storage, engine):
    """Full filter, random walk, ragged shapes (rows not a multiple of the tile, r not a multiple of 4)."""
    c = _capi()
    if engine == "block" and r > 32:
        pytest.skip("the blocked engine needs r <= 32")
    Y, C0 = _problem(d, r, T, 100 + d + r, "t" if robust else "normal")
    if storage == "f32":
        Y = Y.astype(np.float32).astype(np.float64)
        C0 = C0.astype(np.float32).astype(np.float64)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    mode = O.Mode(robust=robust)
    st, Yp, _ = O.run_epoch(st, Y, mode, O.RandomWalkDyn())
    f = c.DeviceFilter(d, r, storage=storage, engine=engine, **_mode_kwargs(mode))
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    f.run(0, T)
    s = f.get_state()
    tol = TOL[storage]
    _compare(s, st, tol)
    assert relerr(f.y_pred(0, T), Yp) < tol
    if robust:
        assert relerr(s["rho"], st.rho) < tol and relerr(s["lam"], st.lam) < 1e-12
    # squared-error reduction on the device == host value
    assert relerr(f.sq_error(0, T), np.sum((Yp - Y) ** 2)) < 10 * tol
    # predict roll-out (random walk: y_hat = C_T mu_T)
    assert relerr(f.predict(T, 3), O.predict_rollout(st.C, st.mu, None, O.RandomWalkDyn(), T, 3)) < tol
    f.close()


@pytest.mark.parametrize("storage", ["f64", "f32"])
@pytest.mark.parametrize("robust", [False, True])
def test_simplified_cos_mode(robust, storage, engine):
    """ExperimentSynthetic configuration on the device: P = 0, eta = tr(R)/d, no coefficient update,
    cos dynamics with theta, closed-form theta gradient (synthetic_psmf.py:78-100)."""
    c = _capi()
    g = load_golden("rpsmf_simplified_cos" if robust else "psmf_simplified_cos")
    T, n_pred = int(g["T"]), int(g["n_pred"])
    Y = g["Y_obs"][:T]
    d, r = g["C0"].shape
    mode = O.Mode(robust=robust, coef_update=False, eta_full=False, pbar_predict=False)
    f = c.DeviceFilter(d, r, storage=storage, dyn_kind=c.DYN_COS_PHASE, engine=engine, **_mode_kwargs(mode))
    f.upload_series(Y)
    f.set_state(g["C0"], g["V0"], g["P0"], np.zeros((r, r)), g["mu0"], rho=1.0, lambda0=1.8, theta=g["theta0"])
    if engine == "block":       # the simplified hooks with a diagonal-Jacobian f: the vector-program kernel (psmf_blk4.hip, filter5)
        assert f.geometry()["filter_kernel"] == "psmf_blk_filter5"
    f.zero_gradsum()
    f.run(0, T)
    s = f.get_state()
    tol = TOL[storage]
    p = f"s_e1_k{T}_"
    assert relerr(s["C"], g[p + "C"]) < tol
    assert relerr(s["V"], g[p + "V"]) < tol
    assert relerr(s["mu"], g[p + "mu"].reshape(-1)) < tol
    # the fixture's gradient is a finite-difference derivative of the reference's likelihood
    assert relerr(s["gradsum"], g["gradsum"][0]) < max(tol, 1e-6) * 10
    # epoch-1 predictions incl. the roll-out: theta is still theta0 at this point
    yp = np.vstack([f.y_pred(0, T), f.predict(T, n_pred)])
    st = O.State(C=g["C0"], V=g["V0"], mu=g["mu0"], P=g["P0"], Q=np.zeros((r, r)), rho=1.0, lam=1.8,
                 theta=g["theta0"].copy(), gradsum=np.zeros(r))
    st, Yp, _ = O.run_epoch(st, Y, mode, O.CosPhaseDyn(r))
    ref = np.vstack([Yp, O.predict_rollout(st.C, st.mu, st.theta, O.CosPhaseDyn(r), T, n_pred)])
    assert relerr(yp, ref) < tol
    assert relerr(s["gradsum"], st.gradsum) < tol * 10
    f.close()


@pytest.mark.parametrize("robust", [False, True])
def test_recursive_adam_on_device(robust):
    """PSMFRecursive / rPSMFRecursive: theta updated by Adam inside the device time loop."""
    c = _capi()
    g = load_golden("rpsmf_recursive" if robust else "psmf_recursive")
    T, n_pred, ue = int(g["T"]), int(g["n_pred"]), int(g["update_every"])
    d, r = g["C0"].shape
    f = c.DeviceFilter(d, r, storage="f64", robust=robust, dyn_kind=c.DYN_COS_PHASE, recursive=True,
                       update_every=ue, adam_lr=1e-3)
    f.upload_series(g["Y"])
    f.set_state(g["C0"], g["V0"], g["P0"], g["Q"], g["mu0"], rho=float(g["rho"]), lambda0=float(g["lambda0"]),
                theta=g["theta0"])
    f.zero_gradsum()
    f.set_adam(np.zeros(r), np.zeros(r))
    f.run(0, T)
    s = f.get_state()
    assert relerr(s["theta"], g["theta"][-1]) < 1e-6
    assert relerr(s["C"], g["C_T"]) < 1e-6
    assert relerr(s["mu"], g["mu_T"]) < 1e-6
    assert relerr(s["P"], g["P_T"]) < 1e-6
    yp = np.vstack([f.y_pred(0, T), f.predict(T, n_pred)])
    assert relerr(yp, g["y_pred"]) < 1e-6
    f.close()


def test_graph_and_eager_agree_bitwise():
    """The hipGraph replay and plain launches run the same kernels: identical bits."""
    c = _capi()
    d, r, T = 2000, 20, 600   # > one graph chunk of 256 steps
    Y, C0 = _problem(d, r, T, 5)
    outs = []
    for use_graph in (True, False):
        f = c.DeviceFilter(d, r, storage="f32", use_graph=use_graph, engine="step")
        f.upload_series(Y.astype(np.float32))
        f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=0.0)
        f.run(0, T)
        outs.append(f.get_state())
        f.close()
    for k in ("C", "V", "mu", "P"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_gram_refresh_changes_nothing_material():
    c = _capi()
    d, r, T = 3000, 16, 300
    Y, C0 = _problem(d, r, T, 9)
    res = []
    for refresh in (0, 64):
        f = c.DeviceFilter(d, r, storage="f32", gram_refresh=refresh, engine="step")
        f.upload_series(Y.astype(np.float32))
        f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=0.0)
        f.run(0, T)
        res.append(f.get_state())
        f.close()
    for k in ("C", "V", "mu", "P"):
        assert relerr(res[0][k], res[1][k]) < 1e-5  # fp32 storage: different rounding path, same bar


def test_collective_path_single_rank_matches_plain(engine):
    """The multi-GPU code path (local reduce kernel -> RCCL all-reduce of h, ee -> serial stage, all
    captured in the hipGraph) run with a one-rank communicator must equal the plain path."""
    c = _capi()
    d, r, T = 1500, 12, 130   # blocked engine: block length 52 -> 3 pipelined blocks (cross-Gram all-reduce on the bulk stream)
    Y, C0 = _problem(d, r, T, 21)
    full = c.DeviceFilter(d, r, storage="f64", engine=engine)
    full.upload_series(Y)
    full.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=0.0)
    full.run(0, T)
    s_full = full.get_state()
    full.close()
    # RCCL with a single rank exercises the collective code path (reduce kernel + all-reduce + graph capture)
    import os

    os.environ["PSMF_FORCE_COLLECTIVE"] = "1"
    try:
        one = c.DeviceFilter(d, r, storage="f64", engine=engine)
        one.comm_init(1, 0, c.DeviceFilter.comm_unique_id())
        one.upload_series(Y)
        one.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=0.0)
        one.run(0, T)
        s_one = one.get_state()
        one.close()
    finally:
        del os.environ["PSMF_FORCE_COLLECTIVE"]
    for k in ("C", "V", "mu", "P"):
        assert relerr(s_one[k], s_full[k]) < 1e-12, k


def test_singular_system_raises_linalgerror(engine):
    """Non-finite state -> the r x r solve has no usable pivot -> LinAlgError, as numpy.linalg.inv would raise."""
    c = _capi()
    d, r, T = 64, 4, 3
    Y = np.zeros((T, d))
    f = c.DeviceFilter(d, r, storage="f64", engine=engine)
    f.upload_series(Y)
    P0 = np.full((r, r), np.nan)
    f.set_state(np.ones((d, r)), np.eye(r), P0, np.eye(r), np.zeros(r), rho=1.0, lambda0=0.0)
    with pytest.raises(np.linalg.LinAlgError):
        f.run(0, T)
    f.close()


def test_engines_agree_and_auto_selects_blocked():
    """The two engines are the same recursion: float64 storage -> agreement to round-off, long series
    (several blocks + a ragged last one)."""
    c = _capi()
    d, r, T = 5000, 20, 200    # block length 64 - 20 = 44: 4 full blocks + 24
    Y, C0 = _problem(d, r, T, 77)
    out = {}
    for eng in ("step", "block", "auto"):
        f = c.DeviceFilter(d, r, storage="f64", robust=True, engine=eng)
        f.upload_series(Y)
        f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
        f.run(0, T)
        out[eng] = (f.get_state(), f.y_pred(0, T), f.geometry())
        f.close()
    assert out["auto"][2]["engine"] == "block" and out["block"][2]["block_steps"] == 44
    for k in ("C", "V", "mu", "P", "rho"):
        assert relerr(out["block"][0][k], out["step"][0][k]) < 1e-10, k
    assert relerr(out["block"][1], out["step"][1]) < 1e-10
    with pytest.raises(ValueError):
        c.DeviceFilter(100, 40, engine="block")      # r > 32
    with pytest.raises(ValueError):
        c.DeviceFilter(100, 8, engine="block", dyn_kind=c.DYN_HOST)      # host-stepped dynamics advance one step at a time
    f = c.DeviceFilter(100, 8, engine="step", dyn_kind=c.DYN_SINUSOID, dyn_flags=3)   # (round 5: the per-step engine evaluates it too)
    assert f.geometry()["engine"] == "step"
    f.close()


@pytest.mark.parametrize("robust", [False, True])
def test_mean_history_matches_oracle(robust, engine):
    """`_mu[k]`, k = 0..T (what TrackingMixin reads, tracking.py:140-142), against the oracle step by step;
    several blocks and a second run segment appended to the first."""
    c = _capi()
    d, r, T = 1500, 9, 140
    Y, C0 = _problem(d, r, T, 5, "t" if robust else "normal")
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mu0 = 0.3 * np.ones(r)
    st = O.State(C=C0, V=V0, mu=mu0, P=P0, Q=Q, rho=1.0, lam=1.8)
    mode = O.Mode(robust=robust)
    ref = [mu0.copy()]
    for k in range(1, T + 1):
        st, _ = O.lowrank_step(st, Y[k - 1], k, mode, O.RandomWalkDyn())
        ref.append(st.mu.copy())
    ref = np.array(ref)
    f = c.DeviceFilter(d, r, storage="f64", engine=engine, **_mode_kwargs(mode))
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, mu0, rho=1.0, lambda0=1.8)
    f.run(0, 100)
    f.run(100, T)
    H = f.mu_history(0, T + 1)
    assert H.shape == (T + 1, r)
    assert np.array_equal(H[0], mu0)
    assert relerr(H, ref) < 1e-9
    assert np.array_equal(H[T], f.get_state()["mu"])
    with pytest.raises(ValueError):
        f.mu_history(0, T + 2)
    f.close()


def test_argument_errors():
    c = _capi()
    with pytest.raises(ValueError):
        c.DeviceFilter(10, 65)
    f = c.DeviceFilter(10, 3)
    with pytest.raises(c.PsmfError):
        f.run(0, 1)   # no state / series yet
    with pytest.raises(ValueError):
        f.upload_series(np.zeros((4, 11)))
    f.close()


@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("d,r,T", [(1004, 17, 200), (2056, 31, 100), (4100, 32, 97), (1600, 24, 130)])
def test_streaming_block_kernels_edge_shapes(d, r, T, robust):
    """Role-specialised filter kernel + streaming cross-Gram / apply kernels (f32 storage, d % 4 == 0, 16 <= r <= 32):
    rows not a multiple of the 16-row tile, odd r (identity padding of the 32 x 32 iterates), r < 32 (three column
    tiles of the next series block), a short last block, several passes over the series (carried inverses)."""
    c = _capi()
    Y, C0 = _problem(d, r, T, 300 + d + r, "t" if robust else "normal")
    Y = Y.astype(np.float32).astype(np.float64)
    C0 = C0.astype(np.float32).astype(np.float64)
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    mode = O.Mode(robust=robust)
    f = c.DeviceFilter(d, r, storage="f32", engine="block", **_mode_kwargs(mode))
    assert f.geometry()["block_steps"] == 64 - r
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    for _ in range(2):                          # second pass: state (and the Newton-Schulz starts) carried
        st, Yp, _ = O.run_epoch(st, Y, mode, O.RandomWalkDyn())
        f.run(0, T)
        s = f.get_state()
        _compare(s, st, TOL["f32"])
        assert relerr(f.y_pred(0, T), Yp) < TOL["f32"]
    cnt = f.counters()
    assert cnt["ns_steps"] + cnt["sweep_steps"] == 2 * T       # every step inverted by exactly one of the two paths
    f.close()


def test_collective_path_single_rank_streaming_kernels():
    """As test_collective_path_single_rank_matches_plain, on the f32 path that runs the streaming bulk kernels
    (cross-Gram all-reduce of the pipelined blocks on the bulk stream)."""
    import os

    c = _capi()
    d, r, T = 2000, 32, 150
    Y, C0 = _problem(d, r, T, 23)
    Y = Y.astype(np.float32).astype(np.float64)
    C0 = C0.astype(np.float32).astype(np.float64)
    out = []
    for coll in (False, True):
        if coll:
            os.environ["PSMF_FORCE_COLLECTIVE"] = "1"
        try:
            f = c.DeviceFilter(d, r, storage="f32", engine="block")
            if coll:
                f.comm_init(1, 0, c.DeviceFilter.comm_unique_id())
            f.upload_series(Y)
            f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=0.0)
            f.run(0, T)
            out.append(f.get_state())
            f.close()
        finally:
            os.environ.pop("PSMF_FORCE_COLLECTIVE", None)
    for k in ("C", "V", "mu", "P"):
        assert relerr(out[1][k], out[0][k]) < 1e-12, k


@pytest.mark.parametrize("robust", [False, True])
def test_split_runs_carry_the_state(robust):
    """A series filtered in two psmf_run calls (split in the middle of a block, state read back in between) equals one call:
    the r x r state travels between blocks / runs as the kernels' register dump, the row-major copies are written by the
    last block of each run (f64 storage: agreement to round-off; different block boundaries, same recursion)."""
    c = _capi()
    d, r, T, T1 = 2048, 32, 330, 141
    Y, C0 = _problem(d, r, T, 77, "t" if robust else "normal")
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mode = O.Mode(robust=robust)
    out = []
    for split in (False, True):
        f = c.DeviceFilter(d, r, storage="f64", engine="block", **_mode_kwargs(mode))
        f.upload_series(Y)
        f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
        if split:
            f.run(0, T1)
            mid = f.get_state()
            assert np.all(np.isfinite(mid["P"])) and np.allclose(mid["P"], mid["P"].T, rtol=0, atol=1e-9 * np.max(np.abs(mid["P"])))
            f.run(T1, T)
        else:
            f.run(0, T)
        out.append((f.get_state(), f.y_pred(0, T)))
        f.close()
    for k in ("C", "V", "mu", "P"):
        assert relerr(out[1][0][k], out[0][0][k]) < 1e-9, k
    assert relerr(out[1][1], out[0][1]) < 1e-9
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    st, Yp, _ = O.run_epoch(st, Y, mode, O.RandomWalkDyn())
    _compare(out[1][0], st, TOL["f64"])


@pytest.mark.parametrize("r,T", [(32, 330), (20, 250)])
def test_chained_blocks_one_launch_per_run(r, T):
    """The blocked engine's default on a GPU box: the coefficient-space filter kernels of a run are ONE launch that hands over
    from block to block inside the kernel (psmf_blk_filter3, BlockParams.chain) -- not a silent per-block or per-step fallback.
    Counters: blocks = ceil(T / (64 - r)), launches = 1; the HIP-event timing of the launch is reported; results vs the oracle
    (a short last block, r < 32 masks and the state carried into a second run included)."""
    import os
    if os.environ.get("PSMF_BLOCK_CHAIN") == "0" or os.environ.get("PSMF_BLOCK_FLAGS") == "0" or os.environ.get("PSMF_BLOCK_PIPE") == "0" \
            or os.environ.get("PSMF_RESERVED_CUS") == "0" or os.environ.get("PSMF_FILTER3") == "0":
        pytest.skip("a fallback was selected in the environment")
    c = _capi()
    d = 2048
    Y, C0 = _problem(d, r, T, 4242 + r, "normal")
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    f = c.DeviceFilter(d, r, storage="f64", engine="block")
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    f.counters(reset=True)
    f.filter_kernel_time(reset=True)
    T1 = T - 100
    f.run(0, T1)
    cnt = f.counters()
    B = 64 - r
    assert cnt["filter_launches"] == -(-T1 // B)
    assert cnt["filter_kernel_launches"] == 1
    n, ms = f.filter_kernel_time()
    assert n == 1 and 0.0 < ms < 1e3
    f.run(T1, T)                               # carried state, another chain
    assert f.counters()["filter_kernel_launches"] == 2
    s = f.get_state()
    f.close()
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    st, _, _ = O.run_epoch(st, Y, O.Mode(), O.RandomWalkDyn())
    _compare(s, st, TOL["f64"])


@pytest.mark.parametrize("fixed_lambda", [False, True])
@pytest.mark.parametrize("d,r,T", [(1536, 32, 150), (1200, 20, 120), (900, 12, 110)])
def test_rpsmf_scaling_factors_and_fixed_lambda(d, r, T, fixed_lambda, engine):
    """rPSMF with use_scaling-style factors alpha, beta != 1 (rpsmf.py:45-51,133-171) and fixed_lambda (rpsmf.py:36-40,170-171):
    every filter kernel of both engines against the oracle (f64 storage)."""
    c = _capi()
    Y, C0 = _problem(d, r, T, 500 + d + r, "t")
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    mode = O.Mode(robust=True, alpha=0.93, beta=1.07, fixed_lambda=fixed_lambda)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    st, Yp, _ = O.run_epoch(st, Y, mode, O.RandomWalkDyn())
    f = c.DeviceFilter(d, r, storage="f64", engine=engine, **_mode_kwargs(mode))
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
    f.run(0, T)
    s = f.get_state()
    _compare(s, st, TOL["f64"])
    assert relerr(f.y_pred(0, T), Yp) < TOL["f64"]
    assert relerr(s["rho"], st.rho) < TOL["f64"] and relerr(s["lam"], st.lam) < 1e-12
    f.close()


@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("r,storage", [(32, "f32"), (20, "f32"), (12, "f64")])
def test_start_predictor_saves_iterations_not_accuracy(r, storage, robust):
    """psmf_blk_filter3's start predictor (rank-2 downdate of the previous inverse, rescaled by the kappa ratio):
    against the plain start (PSMF_NS_PREDICT=0) the same result within the engine's tolerance vs the oracle, with fewer
    Newton-Schulz residual evaluations and no more direct sweeps."""
    import os
    if os.environ.get("PSMF_FILTER3") == "0" or os.environ.get("PSMF_NS") == "0":
        pytest.skip("a fallback was selected in the environment")
    c = _capi()
    d, T = 3072, 400
    Y, C0 = _problem(d, r, T, 777 + r, "normal")
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    st, _, _ = O.run_epoch(st, Y, O.Mode(robust=robust), O.RandomWalkDyn())
    res = {}
    old = os.environ.get("PSMF_NS_PREDICT")
    try:
        for mode in ("0", "7"):
            os.environ["PSMF_NS_PREDICT"] = mode
            f = c.DeviceFilter(d, r, robust=robust, storage=storage, engine="block")
            f.upload_series(Y)
            f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
            if f.geometry()["filter_kernel"].startswith("psmf_blk_filter6"):
                f.close()
                pytest.skip("r <= 16: the sweep-based small-rank kernel, no Newton-Schulz iteration to predict a start for")
            f.counters(reset=True)
            f.run(0, T)
            res[mode] = (f.get_state(), f.counters())
            f.close()
    finally:
        if old is None:
            os.environ.pop("PSMF_NS_PREDICT", None)
        else:
            os.environ["PSMF_NS_PREDICT"] = old
    for mode in ("0", "7"):
        _compare(res[mode][0], st, TOL[storage])
    plain, pred = res["0"][1], res["7"][1]
    # one direct sweep of both matrices costs what ~10 Newton-Schulz iterations do (15 us against 0.9 + the plain step)
    cost = lambda cn: cn["ns_iterations"] + 10 * cn["sweep_steps"]
    assert cost(pred) < 0.8 * cost(plain), (plain, pred)
    assert pred["sweep_steps"] <= plain["sweep_steps"]


def test_bulk_kernel_grid_does_not_change_the_result_beyond_rounding():
    """The streaming bulk kernels are grid-stride over 16-row tiles; the library sizes their grid to the CUs the bulk stream's
    shader engines really have (224 on an MI355X with the filter's CUs reserved).  Any grid gives the same
    filter up to the rounding of a different grouping of the cross-Gram's partial sums."""
    import os
    c = _capi()
    d, r, T = 5000, 32, 130
    Y, C0 = _problem(d, r, T, 99, "normal")
    V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8)
    st, _, _ = O.run_epoch(st, Y, O.Mode(), O.RandomWalkDyn())
    old = os.environ.get("PSMF_BULK_WGS")
    states = []
    try:
        for wgs in ("8", "104", "256", None):
            if wgs is None:
                os.environ.pop("PSMF_BULK_WGS", None)
            else:
                os.environ["PSMF_BULK_WGS"] = wgs
            f = c.DeviceFilter(d, r, storage="f32", engine="block")
            f.upload_series(Y)
            f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
            f.run(0, T)
            states.append(f.get_state())
            f.close()
    finally:
        if old is None:
            os.environ.pop("PSMF_BULK_WGS", None)
        else:
            os.environ["PSMF_BULK_WGS"] = old
    for s in states:
        _compare(s, st, TOL["f32"])
    for s in states[:-1]:
        for k in ("C", "V", "mu", "P"):
            assert np.linalg.norm(s[k] - states[-1][k]) <= 1e-6 * np.linalg.norm(states[-1][k]) + 1e-300
