"""The masked filter of ExperimentImpute beyond one workgroup's LDS (d > 512 or r > 16) and per-replica outcomes.

* psmf_impute_run routes such shapes to the masked per-step engine of the large-d handle (psmf_masked.hip: cfg.masked,
  psmf_upload_mask, psmf_masked_metrics) -- the reference's functions take any d, r (ExperimentImpute/PSMF.py:40-95,
  rPSMF.py:40-148).  Against oracle/impute_oracle.py, float64 on both sides, two passes.
* The same engine on two row shards of one GPU under the host communicator (per step: one all-reduce of r^2 + 1 doubles for the
  masked Gram and the observed count, one of r + 1 for h, ee), replicated state bit-identical.
* One diverging replica in a batch no longer voids the batch (rPSMF.py:236-243: NaN for that repeat, carry on).
"""

import threading

import numpy as np
import pytest

from conftest import relerr
from oracle.impute_oracle import impute_filter
from rpsmf_amd import impute
from rpsmf_amd.sharding import shard_rows

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _problem(d, n, r, seed, miss=0.4, empty_column=True):
    rng = np.random.default_rng(seed)
    Yorig = np.cumsum(0.3 * rng.standard_normal((d, n)), axis=1) + 3.0 * rng.random((d, 1))
    M = (rng.random((d, n)) > miss).astype(int)
    M[3] = 0                 # a row that is never observed
    if empty_column:
        M[:, 5] = 0          # a column with no observation at all (MLE-SMF divides by eta = 0 there, in the reference too)
    Mmiss = ((1 - M) * (rng.random((d, n)) > 0.1)).astype(float)
    return Yorig, M, Mmiss, rng.random((d, r)), rng.random((r, n))


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
@pytest.mark.parametrize("d,n,r", [(2000, 120, 20), (20000, 70, 10), (40, 150, 24), (700, 90, 3)])
def test_impute_run_large_shapes_vs_oracle(d, n, r, robust):
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 100 + d + r)
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    Xo = X0.copy()
    ep, ef, ib, st = impute_filter(Yorig * M, C0, Xo, M, Mmiss, V, Q, 10.0, P, 2, 2, Yorig, 0.0, robust=robust, lambda0=1.8, return_state=True)
    res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8, want_bands=True)
    assert res["kernel"] == "masked per-step engine" and res["status"][0] == 0
    errs = dict(Epred=relerr(res["Epred"][0], ep[0, 1:]), Efull=relerr(res["Efull"][0], ef[0, 1:]), inside=abs(res["inside"][0] - ib),
                C=relerr(res["C"][0], st["C"]), X=relerr(res["X"][0], st["X"]), Yrec=relerr(res["Yrec"][0], st["Yrec"]),
                YrecL=relerr(res["YrecL"][0], st["YrecL"]), YrecH=relerr(res["YrecH"][0], st["YrecH"]))
    print(f"masked large d={d} r={r} n={n} robust={robust}: {errs}  [{res['elapsed_ms']:.1f} ms]")
    # rPSMF at (2 000, 20) is the one badly conditioned case of the set: two float64 routes to P+ inside the numpy oracle itself --
    # solve(I + Pbar G, Pbar) against inv(inv(Pbar) + G) -- end 1.2e-9 apart on C (5e-10 on X) after these 240 steps; every other
    # case agrees to 1e-12
    assert max(errs.values()) < (5e-9 if robust else 1e-9), errs


def test_drop_in_functions_beyond_the_small_engine():
    """The reference's two call signatures at d = 600 (> 512): X mutated in place, Einit in column 0, same return tuple."""
    d, n, r = 600, 60, 12
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 7)
    V, Q, P, R = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r), 10 * np.eye(d)
    for robust in (False, True):
        Xo, Xd = X0.copy(), X0.copy()
        ep, ef, ib = impute_filter(Yorig * M, C0, Xo, M, Mmiss, V, Q, 10.0, P, 2, 2, Yorig, 0.25, robust=robust, lambda0=1.8)
        if robust:
            dp, df, rt, di = impute.robust_PSMF(Yorig * M, C0.copy(), Xd, d, n, r, M, Mmiss, V, Q, R, P, 1.8, 2, 2, Yorig, 0.25)
        else:
            dp, df, rt, di = impute.ProbabilisticSequentialMatrixFactorizer(Yorig * M, C0.copy(), Xd, d, n, r, M, Mmiss, 10, V, Q, R, P, 2, 2, Yorig, 0.25)
        assert dp.shape == (1, 3) and dp[0, 0] == 0.25 and relerr(dp, ep) < 1e-9 and relerr(df, ef) < 1e-9 and abs(di - ib) < 1e-12
        assert relerr(Xd, Xo) < 1e-9


def test_baseline_filters_beyond_the_small_engine():
    """MLE-SMF (MLESMF.py:40-92) and TMF (TMF.py:30-73) at d = 600 / r = 12 and d = 40 / r = 20 on the masked per-step engine
    (cfg.masked = 2 / 3), through the drop-in functions, against the oracle's restatements (pinned to the reference functions'
    outputs by tests/golden/impute_baselines.npz)."""
    from oracle.impute_oracle import mle_smf_filter, tmf_filter

    for d, n, r in ((600, 60, 12), (40, 90, 20)):
        Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 21 + d, empty_column=False)
        Q, P, R = 0.1 * np.eye(r), np.eye(r), 10 * np.eye(d)
        Xo, Xd = X0.copy(), X0.copy()
        ep, ef, ib, st = mle_smf_filter(Yorig * M, C0, Xo, M, Mmiss, Q, 10.0, P, 2, 2, Yorig, 0.3, return_state=True)
        dp, df, rt, di = impute.stochasticGradientStateSpaceMF(Yorig * M, C0.copy(), Xd, d, n, r, M, Mmiss, 10, Q, R, P, 2, 2, Yorig, 0.3)
        assert relerr(dp, ep) < 1e-9 and relerr(df, ef) < 1e-9 and abs(di - ib) < 1e-12 and relerr(Xd, Xo) < 1e-9
        res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, np.eye(r), Q, 10.0, P, 2, 2, method="mle_smf", want_bands=True)
        assert res["kernel"] == "masked per-step engine"
        assert relerr(res["C"][0], st["C"]) < 1e-9 and relerr(res["YrecL"][0], st["YrecL"]) < 1e-9 and relerr(res["YrecH"][0], st["YrecH"]) < 1e-9
        Xo, Xd = X0.copy(), X0.copy()
        ep, ef, st = tmf_filter(Yorig * M, C0, Xo, M, Mmiss, 2, Yorig, 0.3, return_state=True)
        dp, df, rt = impute.temporalRegularizedMF(Yorig * M, C0.copy(), Xd, d, n, r, M, Mmiss, 10, R, 2, Yorig, 0.3)
        assert relerr(dp, ep) < 1e-9 and relerr(df, ef) < 1e-9 and relerr(Xd, Xo) < 1e-9
        res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, np.eye(r), np.eye(r), 1.0, np.eye(r), 0.0, 2, method="tmf")
        assert relerr(res["C"][0], st["C"]) < 1e-9 and res["inside"][0] == 0.0


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_masked_engine_two_shards_on_one_gpu(robust):
    """cfg.masked on two row shards (uneven), host communicator: gathered C, y_hat, summed metrics = the unsharded handle and the
    oracle; replicated V, P, mu, X bit-identical across the shards; message sizes r^2 + 1 and r + 1 per step (the Gram one step ahead)."""
    from rpsmf_amd import _capi as c
    from test_hip_multishard import HostGroup

    d, n, r, nsh = 2001, 50, 20, 2
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 11)
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    Xo = X0.copy()
    ep, ef, ib, st = impute_filter(Yorig * M, C0, Xo, M, Mmiss, V, Q, 10.0, P, 2, 1, Yorig, 0.0, robust=robust, lambda0=1.8, return_state=True)
    Yt, Mt, Mmt = np.ascontiguousarray(Yorig.T), np.ascontiguousarray(M.T), np.ascontiguousarray(Mmiss.T)

    def run(f, row0, dl):
        f.upload_series(np.ascontiguousarray(Yt[:, row0:row0 + dl]))
        f.upload_mask(np.ascontiguousarray(Mt[:, row0:row0 + dl]))
        f.set_state(C0[row0:row0 + dl], V, P, Q, X0[:, n - 1], rho=10.0, lambda0=1.8)
        f.run(0, n // 2)
        f.run(n // 2, n)
        return dict(s=f.get_state(), yp=f.y_pred(0, n), X=f.mu_history(1, n), m=f.masked_metrics(np.ascontiguousarray(Mmt[:, row0:row0 + dl]), 2.0))

    kw = dict(robust=robust, storage="f64", masked=True, engine="step")
    f = c.DeviceFilter(d, r, **kw)
    whole = run(f, 0, d)
    f.close()
    grp = HostGroup(nsh)
    out, errs = [None] * nsh, []

    def worker(rank):
        try:
            row0, dl = shard_rows(d, nsh, rank)
            g = c.DeviceFilter(d, r, row0=row0, d_local=dl, **kw)
            g.comm_init_host(nsh, rank, grp.allreduce(rank))
            out[rank] = run(g, row0, dl)
            g.close()
        except BaseException as e:      # noqa: BLE001
            errs.append((rank, e))
            grp.barrier.abort()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(nsh)]
    for t in th:
        t.start()
    for t in th:
        t.join(200)
    assert not any(t.is_alive() for t in th) and not errs, errs
    for key in ("V", "P", "mu"):
        assert np.array_equal(out[0]["s"][key], out[1]["s"][key]), key          # replicated state: same bits on every shard
    assert np.array_equal(out[0]["X"], out[1]["X"])
    Cg = np.concatenate([o["s"]["C"] for o in out])
    ypg = np.concatenate([o["yp"] for o in out], axis=1)
    m = out[0]["m"] + out[1]["m"]
    assert relerr(Cg, whole["s"]["C"]) < 1e-11 and relerr(ypg, whole["yp"]) < 1e-11 and relerr(out[0]["X"], whole["X"]) < 1e-11
    assert relerr(Cg, st["C"]) < 1e-9 and relerr(out[0]["X"].T, st["X"]) < 1e-9 and relerr(ypg.T, st["Yrec"]) < 1e-9
    assert relerr(np.sqrt(m[0] / m[3]), ep[0, 1]) < 1e-9 and relerr(np.sqrt(m[1] / m[3]), ef[0, 1]) < 1e-9 and abs(m[2] / m[3] - ib) < 1e-12
    assert m[3] == Mmiss.sum() == whole["m"][3] and m[2] == whole["m"][2]                      # the two counts are integers: equal exactly
    # one (h, ee) message per step; one masked Gram per step, computed a step ahead beside the serial stage: + the first step's at the
    # start of the run (the Gram of 'the step after the last' of each of the two runs is computed and unused: n + 1 in all)
    assert set(grp.sizes) == {r * r + 1, r + 1} and grp.sizes.count(r * r + 1) == n + 1 and grp.sizes.count(r + 1) == n
    # negative control: without the exchange the shards do NOT reproduce the filter
    solo = c.DeviceFilter(d, r, row0=0, d_local=shard_rows(d, nsh, 0)[1], **kw)
    bad = run(solo, 0, shard_rows(d, nsh, 0)[1])
    solo.close()
    assert relerr(bad["X"], whole["X"]) > 1e-6


def test_one_diverging_replica_does_not_void_the_batch():
    """status[batch] (ExperimentImpute/rPSMF.py:236-243: a diverged repeat is recorded as NaN, the loop carries on): a replica
    whose C0 overflows the r x r system gets PSMF_ERR_NUMERIC and NaN results; the other replicas are bit-identical to their
    single runs; the drop-in function returns NaN errors and NaN coverage for it."""
    rng = np.random.default_rng(3)
    d, n, r, B = 19, 300, 10, 4
    Yorig = np.cumsum(0.3 * rng.standard_normal((d, n)), axis=1)
    M = (rng.random((B, d, n)) > 0.4).astype(int)
    Mmiss = ((1 - M) * (rng.random((B, d, n)) > 0.1)).astype(float)
    C0, X0 = rng.random((B, d, r)), rng.random((B, r, n))
    C0[2] *= 1e160
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    for robust in (False, True):
        res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8)
        assert res["status"].tolist() == [0, 0, -4, 0]
        assert np.all(np.isnan(res["Epred"][2])) and np.all(np.isnan(res["Efull"][2])) and np.isnan(res["inside"][2])
        for b in (0, 1, 3):
            one = impute.impute_batch(Yorig, M[b], Mmiss[b], C0[b], X0[b], V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8)
            for key in ("Epred", "Efull", "inside", "C", "X"):
                assert np.array_equal(res[key][b], one[key][0]), (robust, b, key)
        fn = impute.robust_PSMF if robust else impute.ProbabilisticSequentialMatrixFactorizer
        args = (Yorig * M[2], C0[2].copy(), X0[2].copy(), d, n, r, M[2], Mmiss[2]) + ((V, Q, 10 * np.eye(d), P, 1.8) if robust else (10, V, Q, 10 * np.eye(d), P)) + (2, 2, Yorig, 0.5)
        ep, ef, rt, ib = fn(*args)
        assert ep[0, 0] == 0.5 and np.all(np.isnan(ep[0, 1:])) and np.all(np.isnan(ef[0, 1:])) and np.isnan(ib)
    # the same on the large-shape route (replicas one after the other on the masked per-step engine)
    d, n, r, B = 600, 40, 6, 3
    Yorig = np.cumsum(0.3 * rng.standard_normal((d, n)), axis=1)
    M = (rng.random((B, d, n)) > 0.4).astype(int)
    Mmiss = ((1 - M) * (rng.random((B, d, n)) > 0.1)).astype(float)
    C0, X0 = rng.random((B, d, r)), rng.random((B, r, n))
    C0[1] *= 1e160
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2, 2, robust=True, lambda0=1.8)
    assert res["status"].tolist() == [0, -4, 0] and np.isnan(res["inside"][1]) and np.all(np.isfinite(res["Epred"][[0, 2]]))
    one = impute.impute_batch(Yorig, M[2], Mmiss[2], C0[2], X0[2], V, Q, 10.0, P, 2, 2, robust=True, lambda0=1.8)
    assert np.array_equal(res["Epred"][2], one["Epred"][0]) and np.array_equal(res["X"][2], one["X"][0])


@pytest.mark.parametrize("robust", [False, True], ids=["PSMF", "rPSMF"])
def test_masked_handle_f32_storage(robust):
    """cfg.masked with float32 storage of C, y, y_hat (one rounding of C per timestep; the r x r state and every accumulator
    float64): within the north-star 1e-5 of the float64 oracle over one pass; cos-phase dynamics and R / Q schedules are refused on
    masked handles (the ExperimentImpute filters are random walks with constant noise levels)."""
    from rpsmf_amd import _capi as c

    d, n, r = 6000, 150, 16
    Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 5)
    Yorig = Yorig.astype(np.float32).astype(np.float64)
    C0 = C0.astype(np.float32).astype(np.float64)
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    Xo = X0.copy()
    ep, ef, ib, st = impute_filter(Yorig * M, C0, Xo, M, Mmiss, V, Q, 10.0, P, 2, 1, Yorig, 0.0, robust=robust, lambda0=1.8, return_state=True)
    f = c.DeviceFilter(d, r, robust=robust, storage="f32", masked=True, engine="step")
    f.upload_series(np.ascontiguousarray(Yorig.T))
    f.upload_mask(np.ascontiguousarray(M.T))
    f.set_state(C0, V, P, Q, X0[:, n - 1], rho=10.0, lambda0=1.8)
    f.run(0, n)
    s, X, m = f.get_state(), f.mu_history(1, n), f.masked_metrics(np.ascontiguousarray(Mmiss.T), 2.0)
    with pytest.raises(ValueError, match="rPSMF" if robust else "masked"):
        f.set_schedules(np.ones(n + 1), None)
    f.close()
    errs = dict(C=relerr(s["C"], st["C"]), X=relerr(X.T, st["X"]), P=relerr(s["P"], st["P"]), V=relerr(s["V"], st["V"]),
                Epred=relerr(np.sqrt(m[0] / m[3]), ep[0, 1]), Efull=relerr(np.sqrt(m[1] / m[3]), ef[0, 1]))
    print(f"masked f32 storage d={d} r={r} robust={robust}: {errs}")
    assert max(errs.values()) < 1e-5, errs
    assert abs(m[2] / m[3] - ib) < 5e-4          # coverage: entries at a band edge may change sides at float32 resolution
    with pytest.raises(ValueError, match="random-walk"):
        c.DeviceFilter(d, r, masked=True, dyn_kind=c.DYN_COS_PHASE, engine="step", storage="f64")


def test_masked_engine_random_shapes_and_methods():
    """Shapes, ranks, methods and pass counts drawn at random (fixed seed) beyond the small-shape engine: d in 513 .. 3000 with any
    r <= 48, or r in 17 .. 48 with a small d -- every rank parity (odd / even: the identity-padded 2 x 2 pivot), both tile counts of
    the wave-local solve and the LDS solve (r > 32), the four methods -- against the oracle."""
    from oracle.impute_oracle import mle_smf_filter, tmf_filter

    rng = np.random.default_rng(20261004)
    for case in range(8):
        if case % 2 == 0:
            d, r = int(rng.integers(513, 3000)), int(rng.integers(1, 49))
        else:
            d, r = int(rng.integers(20, 200)), int(rng.integers(17, 49))
        n, iters = int(rng.integers(25, 70)), int(rng.integers(1, 3))
        method = ("psmf", "rpsmf", "mle_smf", "tmf")[case % 4]
        Yorig, M, Mmiss, C0, X0 = _problem(d, n, r, 1000 + case, empty_column=(method != "mle_smf"))
        V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
        Xo = X0.copy()
        if method in ("psmf", "rpsmf"):
            ep, ef, ib, st = impute_filter(Yorig * M, C0, Xo, M, Mmiss, V, Q, 10.0, P, 2, iters, Yorig, 0.0, robust=(method == "rpsmf"), lambda0=1.8, return_state=True)
        elif method == "mle_smf":
            ep, ef, ib, st = mle_smf_filter(Yorig * M, C0, Xo, M, Mmiss, Q, 10.0, P, 2, iters, Yorig, 0.0, return_state=True)
        else:
            ep, ef, st = tmf_filter(Yorig * M, C0, Xo, M, Mmiss, iters, Yorig, 0.0, return_state=True)
            ib = 0.0
        res = impute.impute_batch(Yorig, M, Mmiss, C0, X0, V, Q, 10.0, P, 2 if method != "tmf" else 0.0, iters, method=method, lambda0=1.8)
        assert res["kernel"] == "masked per-step engine" and res["status"][0] == 0
        errs = dict(Epred=relerr(res["Epred"][0], ep[0, 1:]), Efull=relerr(res["Efull"][0], ef[0, 1:]), inside=abs(res["inside"][0] - ib),
                    C=relerr(res["C"][0], st["C"]), X=relerr(res["X"][0], st["X"]))
        print(f"masked random case {case}: d={d} r={r} n={n} passes={iters} {method}: {errs}")
        assert max(errs.values()) < 5e-9, (case, d, r, method, errs)
