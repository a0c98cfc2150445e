#!/usr/bin/env python
"""Headline benchmark: PSMF filter timesteps/s at d=100k, r=32 (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" of this harness = one pass of the hot path (T filter timesteps, full PSMF filter:
predict / y_hat = C mu_bar / innovation / r x r solve / low-rank updates of C, mu, P, V) over a
synthetic series that is already resident in HBM.  value = K * T / elapsed  (timesteps per
second of the whole job).  For N > 1 the d rows of C and y are sharded over the ranks
(strong scaling; blocked engine: one RCCL all-reduce of a 128 x 64 cross-Gram per block of 64 - r
timesteps, off the critical path; per-step engine: r+1 doubles per timestep); launch with
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- torch is used
only as rendezvous plumbing (gloo: id broadcast, barrier, max over ranks), never for compute.

Extra objects on the JSON line (rank 0; the N = 1 run carries all of them):
  roofline      dominant kernel (SURVEY 8(d) definition): algorithmic bytes 8 d (r+1) per timestep x timesteps per
                launch / its average duration measured with HIP events on the stream it runs on, vs 8 TB/s HBM --
                kept as the contract defines it, and labelled for what it is: the blocked engine REMOVES that
                traffic, so this ratio is a speed-up over the step-at-a-time algorithm's roofline, not a bandwidth.
                The distances to the actual limits are beside it:
                  real_hbm_GBps       HBM bytes all kernels of a pass really move (rocprofv3 PMC passes, profiles/) / wall
                  filter_chain_frac   matrix-core issue time of the Newton-Schulz products of a timestep / the measured
                                      time per timestep inside the filter kernel (its own s_memrealtime stamps)
                  bulk_kernels        the two d-sized kernels that do stream the data: measured GB/s vs HBM peak
                  hbm_peak_measured   a plain copy kernel on this box, next to the nominal 8 TB/s
  cold_pass_steps_per_s   config E as literally stated: ONE pass of T timesteps from the initial state (the first
                ~400 timesteps invert by direct sweeps), timed on its own before the steady-state passes of `value`
  cpu_baseline  the CPU oracle (numpy restatement of the reference algorithm, O(d r^2) form)
                timed on a bounded prefix of the same series on this box's host cores.
  cpu_baseline_literal    the reference-shaped O(d^2 r) algebra (dense R, d x d inverse innovation: what the
                reference itself costs) at d = 2000, where it still fits
  other_configs BASELINE configs B, C (d = 10 000, r = 20, T = 5 000, PSMF / rPSMF) and D (masked filter, 19 x 295 719,
                50 seeds in one launch): throughput + parity against the oracle, each; config E again with float64 storage, on the
                per-step engine (one persistent launch per pass) and masked; a small rank (r = 12) and r = 40 (beyond the blocked
                engine: the persistent per-step kernel with the hub's matrices in LDS); ExperimentSynthetic's hooks, Beijing's dynamics
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md)
# Newton-Schulz iteration of the filter kernel, per SIMD: one 32x32x32 float64 product (16 v_mfma_f64_16x16x4_f64, 64 cycles
# each) + one float32 correction product (16 v_mfma_f32_16x16x4_f32, 32 cycles each); both inverses of a step iterate in
# parallel on the four SIMDs of the workgroup's CU.  2.4 GHz.
NS_ITER_ISSUE_US = (16 * 64 + 16 * 32) / 2.4e3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    # (long aliases: under `python -m torch.distributed.run` a bare --d / --r is ambiguous with the launcher's own options)
    ap.add_argument("--d", "--rows", type=int, default=100_000, dest="d")
    ap.add_argument("--r", "--latent-rank", type=int, default=32, dest="r")
    ap.add_argument("--T", "--timesteps", type=int, default=10_000, dest="T")
    ap.add_argument("--robust", type=int, default=0)
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--cpu-steps", type=int, default=300, help="timesteps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-y-pred", action="store_true")
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--engine", default="auto", choices=["auto", "step", "block"])
    ap.add_argument("--no-extras", action="store_true", help="skip other_configs / literal baseline / copy peak (profiling runs)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "gloo"],
                    help="N > 1: RCCL all-reduce on the device streams (default), or the host-mediated communicator over gloo "
                         "(psmf_comm_init_host) -- the transport for rehearsing the N > 1 path with all ranks on ONE GPU")
    ap.add_argument("--one-device", action="store_true", help="every rank uses HIP device 0 (rehearsal on a one-GPU box; needs --comm gloo)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = BASELINE config E as stated, the d rows sharded over the ranks (default); weak = every rank holds "
                         "--rows rows, d = N x rows (what more GPUs buy here: rows at the same timesteps/s, DESIGN section 6)")
    ap.add_argument("--pid-dir", default=None, help="write rank<r>.pid / rank<r>.timed marker files there (fault-injection tests)")
    ap.add_argument("--parity-fixture", default=None,
                    help="stored oracle answers of this workload (format of tests/golden/make_golden_fullsize.py) for the in-run parity of a "
                         "sharded run; default: tests/golden/fullsize_E_{psmf,rpsmf}.npz when the workload is BASELINE config E")
    return ap.parse_args()


class Series:
    """Synthetic series of ExperimentSynthetic/data.py semantics, generated shard-by-shard in
    time chunks so that neither the host nor PCIe ever holds the whole (T, d) array."""

    def __init__(self, d, r, T, seed, row0, d_local, robust, global_noise=False):
        # global_noise: a row shard draws the noise of the WHOLE width (the N = 1 stream) and keeps its columns, so that the
        # series of a sharded run is bit for bit the series of the unsharded one (the stored full-horizon answers then apply to
        # every N); otherwise a shard has a noise stream of its own (weak scaling: d grows with N)
        self.d, self.row0, self.global_noise = d, row0, bool(global_noise) and d_local != d
        rng = np.random.default_rng(seed)
        C_true = rng.standard_normal((d, r))
        self.Ct = np.ascontiguousarray(C_true[row0:row0 + d_local].T)
        theta = 1e-3 * np.arange(1, r + 1)
        x = rng.standard_normal(r)
        self.X = np.empty((T, r))
        for t in range(1, T + 1):
            x = np.cos(2.0 * np.pi * theta * t + x)
            self.X[t - 1] = x
        self.T, self.d_local, self.robust = T, d_local, robust
        self.noise_seed = seed * 1000 + (0 if self.global_noise else row0)

    def chunks(self, chunk=500):
        rng = np.random.default_rng(self.noise_seed)
        sd = np.sqrt(0.1)
        for a in range(0, self.T, chunk):
            b = min(self.T, a + chunk)
            width = self.d if self.global_noise else self.d_local
            if self.robust:
                eps = rng.standard_t(3.0, (b - a, width)).astype(np.float32)
            else:
                eps = rng.standard_normal((b - a, width), dtype=np.float32)
            if self.global_noise:
                eps = np.ascontiguousarray(eps[:, self.row0:self.row0 + self.d_local])
            Y = (self.X[a:b] @ self.Ct).astype(np.float32)
            Y += np.float32(sd) * eps
            yield a, Y


def geo_engine_block(f):
    return f.geometry()["engine"] == "block"


def init_state(d, r, seed):
    rng = np.random.default_rng(seed + 7)
    C0 = (0.1 * rng.standard_normal((d, r))).astype(np.float32).astype(np.float64)
    return dict(C=C0, V=0.1 * np.eye(r), P=np.eye(r), Q=0.1 * np.eye(r), mu=np.zeros(r), rho=1.0, lam=1.8)


def blas_threads():
    try:
        import threadpoolctl

        return int(max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1]))
    except Exception:
        return os.cpu_count() or 1


def oracle_prefix(series, st0, n, robust):
    """CPU oracle over the first n timesteps of `series`; returns (final state, seconds)."""
    from oracle import psmf_oracle as O

    Y = np.vstack([Yc for _, Yc in series.chunks(chunk=n)][:1])[:n].astype(np.float64)
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=st0["P"].copy(), Q=st0["Q"].copy(),
                 rho=st0["rho"], lam=st0["lam"])
    t0 = time.perf_counter()
    st, _, _ = O.run_epoch(st, Y, O.Mode(robust=bool(robust)), O.RandomWalkDyn(), want_grad=False)
    return st, time.perf_counter() - t0


def parity_of(f, st_cpu, n):
    """max |a - b| / max |b| per array (the norm of the 1e-5 bar), and the same entry by entry over the entries >= 1e-3 of the largest"""
    s = f.get_state()
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    def elem(a, b):
        a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
        big = np.abs(b) >= 1e-3 * np.max(np.abs(b))
        return float(np.max(np.abs(a[big] - b[big]) / np.abs(b[big])))

    out = dict(steps=n, norm="max|a-b| / max|b|", C=rel(s["C"], st_cpu.C), V=rel(s["V"], st_cpu.V), mu=rel(s["mu"], st_cpu.mu), P=rel(s["P"], st_cpu.P))
    out["elementwise_over_entries_above_1e-3_of_max"] = dict(C=elem(s["C"], st_cpu.C), V=elem(s["V"], st_cpu.V), mu=elem(s["mu"], st_cpu.mu), P=elem(s["P"], st_cpu.P))
    return out


def cpu_baseline(args, series, st0):
    """Oracle (kind 'port') on the first cpu-steps timesteps; also returns its final state so the
    GPU result can be checked against it in the same run."""
    n = min(args.cpu_steps, args.T)
    st, dt = oracle_prefix(series, st0, n, args.robust)
    info = dict(value=n / dt, unit="timesteps/s", cores=blas_threads(), kind="port",
                sample=f"first {n} of {args.T} timesteps of the same series (d={args.d}, r={args.r}), numpy float64 "
                       f"O(d r^2) restatement of the reference algorithm, {dt:.1f} s")
    return info, st, n


def cpu_baseline_f32(args, series, st0, budget_s=10.0):
    """SURVEY 8(d) asks for the CPU path in float32 beside float64: the same O(d r^2) step (lowrank_step of the oracle: y_hat, e, the
    exact Gram, h, the rank-1 update of C) with every d-sized array and product in float32 (sgemm / sgemv), the r x r algebra in
    float64.  A timing baseline only -- nothing is checked against it."""
    d, r = args.d, args.r
    n_max = min(args.cpu_steps, args.T)
    Y = np.vstack([Yc for _, Yc in series.chunks(chunk=n_max)][:1])[:n_max].astype(np.float32)
    C = st0["C"].astype(np.float32)
    V, P, Q, mu = st0["V"].copy(), st0["P"].copy(), st0["Q"].copy(), st0["mu"].copy()
    rho, lam, robust = float(st0["rho"]), float(st0["lam"]), bool(args.robust)
    n, t0 = 0, time.perf_counter()
    while n < n_max and time.perf_counter() - t0 < budget_s:
        P_bar = P + Q
        mb32 = mu.astype(np.float32)
        e = Y[n] - C @ mb32
        w = V @ mu
        s_ = float(mu @ w)
        G = (C.T @ C).astype(np.float64)
        eta = rho + float(np.sum(G * P_bar)) / d
        N = s_ + eta
        kap = 1.0 / (rho + s_)
        h = (C.T @ e).astype(np.float64)
        ee = float(e @ e)
        P_plus = np.linalg.solve(np.eye(r) + kap * (P_bar @ G), P_bar)
        P_plus = 0.5 * (P_plus + P_plus.T)
        b = kap * h
        mu = mu + P_plus @ b
        C += np.outer(e, (w / N).astype(np.float32))
        V = V - np.outer(w, w) / N
        if robust:
            om = (lam + kap * ee - float(b @ P_plus @ b)) / (lam + d)
            V *= (lam + ee / N) / (lam + d)
            P_plus *= om
            Q = om * Q
            rho *= om
            lam += d
        P = P_plus
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="timesteps/s", cores=blas_threads(), kind="port",
                sample=f"first {n} of {args.T} timesteps of the same series (d={d}, r={r}), numpy float32 d-sized arrays and products "
                       f"(float64 r x r algebra), the oracle's O(d r^2) step restated in bench.py, {dt:.1f} s; timing only, not a checker")


def cpu_baseline_literal(r, seed, budget_s=12.0):
    """The reference-shaped algebra (oracle.literal_step: dense d x d R, kron, d x d inverse innovation, d x d trace --
    psmf.py:121-165 op for op) at d = 2000, the largest size at which it is still practical."""
    from oracle import psmf_oracle as O

    d = 2000
    Y = O.synthetic_series(d, r, 64, seed, dtype=np.float64)
    st0 = init_state(d, r, seed)
    st = O.State(C=st0["C"], V=st0["V"], mu=st0["mu"], P=st0["P"], Q=st0["Q"], rho=1.0, lam=1.8)
    mode, dyn = O.Mode(), O.RandomWalkDyn()
    st, _ = O.literal_step(st, Y[0], 1, mode, dyn)           # warm the BLAS threads
    n, t0 = 0, time.perf_counter()
    while n < 63 and time.perf_counter() - t0 < budget_s:
        st, _ = O.literal_step(st, Y[n + 1], n + 2, mode, dyn)
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="timesteps/s", cores=blas_threads(), kind="port",
                sample=f"{n} timesteps at d={d}, r={r} (one d x d float64 temporary at d=100 000 would be 80 GB), numpy float64 "
                       f"O(d^2 r) literal restatement of pypsmf/psmf/psmf.py:121-165, {dt:.1f} s")


def run_filter_config(_capi, name, d, r, T, robust, passes=3, storage="f32", engine="auto", bulk_times=False, oracle=None):
    """Configs B / C (and config E under another storage type / engine): cold pass, steady passes, parity against the oracle on the
    first 300 timesteps."""
    seed = 35833 if robust else 35853
    series = Series(d, r, T, seed, 0, d, robust)
    st0 = init_state(d, r, seed)
    f = _capi.DeviceFilter(d, r, robust=robust, storage=storage, engine=engine)
    for a, Yc in series.chunks(chunk=1000):
        f.upload_series(Yc, t0=a, T_total=T)
    reset = lambda: f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
    n_par = 300
    if oracle is not None:      # (state after n steps, n, seconds): the headline's own oracle prefix -- same seed, same series generator
        st_cpu, n_par, dt_cpu = oracle
    else:
        st_cpu, dt_cpu = oracle_prefix(series, st0, n_par, robust)
    reset()
    f.run(0, n_par)
    par = parity_of(f, st_cpu, n_par)
    reset()
    f.run(0, T)                                    # untimed: runtime pools, instruction caches
    reset()
    f.sync()
    t0 = time.perf_counter()
    f.run(0, T)
    cold = time.perf_counter() - t0
    if robust:
        f.set_state(Q=st0["Q"], rho=st0["rho"], lambda0=st0["lam"])    # rPSMF epochs restart Q, R, lambda (rpsmf.py:106-114)
    t0 = time.perf_counter()
    for _ in range(passes):
        f.run(0, T, sync=False)
    f.sync()
    steady = (time.perf_counter() - t0) / passes
    geo = f.geometry()
    out = {"workload": f"{'rPSMF' if robust else 'PSMF'} full filter d={d} r={r} T={T}, {storage} storage, 1 GPU", "value": T / steady,
           "unit": "timesteps/s", "cold_pass_steps_per_s": T / cold, "us_per_timestep": 1e6 * steady / T, "engine": geo["engine"],
           "kernel": geo.get("filter_kernel"),
           "parity_vs_cpu_oracle": par, "cpu_oracle_steps_per_s": n_par / dt_cpu,
           "hbm_frac_step_at_a_time": (T / steady) * (8.0 if storage == "f32" else 16.0) * d * (r + 1) / (HBM_PEAK_GBS * 1e9)}
    if bulk_times and geo["engine"] == "block":      # the two d-sized kernels of a block, timed alone (they run beside the filter chain)
        out["bulk_kernels_us"] = {"cross_gram+reduce": f.time_kernel(1, 50), "apply": f.time_kernel(2, 50), "block_steps": geo["block_steps"]}
    f.close()
    return out


def run_masked_config(_capi, d=100_000, r=32, T=1_000, missing=0.4, n_par=40, passes=3):
    """The masked filter north_star describes ("masked innovation-covariance assembly over the observed-index set") at config E's shape:
    40 % of the entries missing at random, the large-d masked handle (cfg.masked = 1) -- one persistent launch per pass, the masked Gram
    of every timestep formed from the on-chip C -- with parity against the oracle's masked step on the first timesteps."""
    from oracle import psmf_oracle as O

    seed = 35871
    series = Series(d, r, T, seed, 0, d, False)
    st0 = init_state(d, r, seed)
    Y = np.vstack([Yc for _, Yc in series.chunks(chunk=T)])[:T].astype(np.float64)
    rng = np.random.default_rng(seed)
    M = (rng.random((T, d)) >= missing).astype(np.uint8)
    Y *= M
    f = _capi.DeviceFilter(d, r, storage="f64", engine="step", masked=True)
    f.upload_series(Y)
    f.upload_mask(M)
    reset = lambda: f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=st0["P"].copy(), Q=st0["Q"].copy(), rho=st0["rho"], lam=st0["lam"])
    t0 = time.perf_counter()
    for k in range(n_par):
        st, _ = O.lowrank_step(st, Y[k], k + 1, O.Mode(), O.RandomWalkDyn(), mask=M[k].astype(float), want_grad=False)
    dt_cpu = time.perf_counter() - t0
    reset()
    f.run(0, n_par)
    par = parity_of(f, st, n_par)
    reset()
    f.run(0, T)
    f.sync()
    t0 = time.perf_counter()
    for _ in range(passes):
        f.run(0, T, sync=False)
    f.sync()
    steady = (time.perf_counter() - t0) / passes
    geo = f.geometry()
    f.close()
    return {"workload": f"masked PSMF (ExperimentImpute/PSMF.py:59-84 semantics), d={d} r={r} T={T}, {int(100 * missing)} % missing, f64 storage, 1 GPU",
            "value": T / steady, "unit": "masked timesteps/s", "us_per_timestep": 1e6 * steady / T, "kernel": geo.get("filter_kernel"),
            "parity_vs_cpu_oracle": par, "cpu_oracle_steps_per_s": n_par / dt_cpu}


def run_synthetic_simplified(_capi, d=10_000, r=20, T=5_000, robust=False, passes=3):
    """ExperimentSynthetic's own hook configuration (synthetic_psmf.py:78-106: P_bar = P_{k-1}, eta = tr R / d, no coefficient
    update, f = cos(2 pi theta t + x)) at config B's size: throughput and parity against the oracle on the first 300 timesteps."""
    from oracle import psmf_oracle as O

    seed = 35833 if robust else 35853
    series = Series(d, r, T, seed, 0, d, robust)
    st0 = init_state(d, r, seed)
    theta0 = 0.1 * np.random.default_rng(seed + 1).random(r)        # synthetic_psmf.py:128: theta0 = 0.1 * rand(r, 1)
    P0, Q0 = np.zeros((r, r)), np.zeros((r, r))                     # synthetic_psmf.py:131-132
    f = _capi.DeviceFilter(d, r, robust=robust, storage="f32", dyn_kind=_capi.DYN_COS_PHASE, coef_update=False, eta_full=False, pbar_predict=False)
    for a, Yc in series.chunks(chunk=1000):
        f.upload_series(Yc, t0=a, T_total=T)
    reset = lambda: f.set_state(st0["C"], st0["V"], P0, Q0, st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta0)
    n_par = 300
    Y = np.vstack([Yc for _, Yc in series.chunks(chunk=n_par)][:1])[:n_par].astype(np.float64)
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=P0, Q=Q0, rho=st0["rho"], lam=st0["lam"], theta=theta0.copy(), gradsum=np.zeros(r))
    st, _, _ = O.run_epoch(st, Y, O.Mode(robust=robust, coef_update=False, eta_full=False, pbar_predict=False), O.CosPhaseDyn(r))
    reset(); f.zero_gradsum()
    f.run(0, n_par)
    s = f.get_state()
    rel = lambda a, b_: float(np.max(np.abs(a - b_)) / np.max(np.abs(b_)))
    par = dict(steps=n_par, C=rel(s["C"], st.C), V=rel(s["V"], st.V), mu=rel(s["mu"], st.mu), gradsum=rel(s["gradsum"], st.gradsum))
    reset(); f.run(0, T); reset(); f.sync()
    t0 = time.perf_counter()
    for _ in range(passes):
        f.run(0, T, sync=False)
    f.sync()
    dt = (time.perf_counter() - t0) / passes
    geo = f.geometry()
    f.close()
    return {"workload": f"{'rPSMF' if robust else 'PSMF'} with ExperimentSynthetic's simplified hooks, f = cos(2 pi theta t + x), d={d} r={r} T={T}, f32 storage, 1 GPU",
            "value": T / dt, "unit": "timesteps/s", "us_per_timestep": 1e6 * dt / T, "kernel": geo["filter_kernel"], "parity_vs_cpu_oracle": par}


def run_fourier_config(_capi, d=10_000, r=10, T=5_000, N=2, passes=3):
    """ExperimentBeijing's dynamics family (beijing_psmf.py:97-140: full PSMF filter, f = FourierBasis, r = 10) at config B's d and
    T: throughput of the small-rank block filter (psmf_blk_filter6) and parity (state and theta gradient) against the oracle run on
    the same callable (complex-step derivatives) over the first 200 timesteps."""
    from oracle import psmf_oracle as O
    from rpsmf_amd import nonlinearities as NL

    seed = 35861
    nl = NL.FourierBasis(r, N=N)
    series = Series(d, r, T, seed, 0, d, False)
    st0 = init_state(d, r, seed)
    theta0 = 0.1 * np.random.default_rng(seed + 1).random(nl.n_params)          # beijing_psmf.py:117: theta0 = 0.1 * rand
    f = _capi.DeviceFilter(d, r, storage="f32", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags, dyn_terms=nl.device_terms)
    for a, Yc in series.chunks(chunk=1000):
        f.upload_series(Yc, t0=a, T_total=T)
    reset = lambda: f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta0)
    n_par = 200
    Y = np.vstack([Yc for _, Yc in series.chunks(chunk=n_par)][:1])[:n_par].astype(np.float64)
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=st0["P"].copy(), Q=st0["Q"].copy(), rho=st0["rho"], lam=st0["lam"],
                 theta=theta0.copy(), gradsum=np.zeros(nl.n_params))
    st, _, _ = O.run_epoch(st, Y, O.Mode(robust=False), O.CallableDyn(nl, nl.n_params))
    reset(); f.zero_gradsum()
    f.run(0, n_par)
    s = f.get_state()
    rel = lambda a, b_: float(np.max(np.abs(a - b_)) / np.max(np.abs(b_)))
    par = dict(steps=n_par, C=rel(s["C"], st.C), V=rel(s["V"], st.V), P=rel(s["P"], st.P), mu=rel(s["mu"], st.mu), gradsum=rel(s["gradsum"], st.gradsum))
    reset(); f.run(0, T); reset(); f.sync()
    t0 = time.perf_counter()
    for _ in range(passes):
        f.run(0, T, sync=False)
    f.sync()
    dt = (time.perf_counter() - t0) / passes
    geo = f.geometry()
    f.close()
    return {"workload": f"PSMF full filter, f = FourierBasis(N={N}) ({nl.n_params} parameters), d={d} r={r} T={T}, f32 storage, 1 GPU",
            "value": T / dt, "unit": "timesteps/s", "us_per_timestep": 1e6 * dt / T, "kernel": geo["filter_kernel"], "parity_vs_cpu_oracle": par}


def run_impute_config(seeds=50, n=295_719, d=19, r=10, n_par=3000, variants=(False, True)):
    """Config D: masked filter, gas-sensor shape (the CSV is not in the reference checkout: synthetic stand-in of the same
    shape), 40 % missing, `seeds` replicas in one launch; parity of replica 0 on an n_par-column prefix against the oracle."""
    from oracle.impute_oracle import impute_filter
    from rpsmf_amd import impute, impute_harness as H

    rng = np.random.default_rng(20160930)
    Yorig = np.cumsum(0.05 * rng.standard_normal((d, n)), axis=1) + 10.0 * rng.random((d, 1))
    Yorig[rng.random((d, n)) < 0.01] = np.nan
    Yint = np.nan_to_num(Yorig, nan=0.0)
    np.random.seed(123)
    M, Mm, C0, X0 = [], [], [], []
    for _ in range(seeds):
        p = H.draw_problem(Yorig, 40, r)
        M.append(p["M"].astype(np.uint8)); Mm.append(p["Mmiss"].astype(np.uint8)); C0.append(p["C"]); X0.append(p["X"])
    M, Mm, C0, X0 = np.stack(M), np.stack(Mm), np.stack(C0), np.stack(X0)
    V, Q, P = 2 * np.eye(r), 0.1 * np.eye(r), np.eye(r)
    out = {}
    for robust in variants:
        res = impute.impute_batch(Yint, M, Mm, C0, X0, V, Q, 10.0, P, 2, 2, robust=robust, lambda0=1.8)
        steps = seeds * 2 * n
        # parity: replica 0 on a prefix, device vs oracle
        Xo = X0[0][:, :n_par].copy()
        t0 = time.perf_counter()
        ep, ef, ib = impute_filter(Yint[:, :n_par] * M[0][:, :n_par], C0[0], Xo, M[0][:, :n_par], np.maximum(Mm[0][:, :n_par], 1e-300).astype(float),
                                   V, Q, 10.0, P, 2, 2, Yint[:, :n_par], 0.0, robust=robust, lambda0=1.8)
        dt = time.perf_counter() - t0
        one = impute.impute_batch(Yint[:, :n_par], M[0][:, :n_par], np.maximum(Mm[0][:, :n_par], 0), C0[0], X0[0][:, :n_par], V, Q, 10.0, P, 2, 2,
                                  robust=robust, lambda0=1.8)
        rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))
        out["rPSMF" if robust else "PSMF"] = {
            "workload": f"{'rPSMF' if robust else 'PSMF'} masked filter {d}x{n}, r={r}, Iter=2, 40% missing, {seeds} seeds in one launch",
            "value": steps / (res["elapsed_ms"] * 1e-3), "unit": "timesteps/s (all replicas)", "kernel_s": res["elapsed_ms"] * 1e-3,
            "us_per_column_per_replica": 1e3 * res["elapsed_ms"] / (2 * n),
            "parity_vs_cpu_oracle": {"columns": n_par, "Epred": rel(one["Epred"][0], ep[0, 1:]), "Efull": rel(one["Efull"][0], ef[0, 1:]),
                                     "inside_abs": abs(float(one["inside"][0]) - ib)},
            "cpu_oracle_steps_per_s": 2 * n_par / dt,
            "reference_published_s_per_seed": 100.68 if not robust else 109.38}
    return out


def _digest(*arrays):
    import hashlib

    hsh = hashlib.blake2b(digest_size=16)
    for a in arrays:
        hsh.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return hsh.hexdigest()


def sharded_parity_vs_fixture(f, path, d, r, T, seed, robust, row0, d_local, world, dist, reset):
    """In-run parity of a SHARDED run at a size the oracle cannot be re-run at (BASELINE config E: 0.1 s of CPU per timestep):
    stored oracle answers of the same workload (tests/golden/make_golden_fullsize.py: r-sized state, eta, N and a fixed 64-column
    sketch S^T C at checkpoints).  Every rank filters to the first two checkpoints; the replicated V, P, mu, eta, N must be
    BIT-IDENTICAL across the ranks (digests gathered over the rendezvous) and within 1e-5 of the stored answers; the sketch is
    summed over the ranks' row shards (S[row0 : row0 + d_local]^T C_local) and compared as a whole."""
    import torch

    g = np.load(path)
    assert (int(g["d"]), int(g["r"]), int(g["T"]), int(g["seed"]), bool(g["robust"])) == (d, r, T, seed, robust), "fixture is of another workload"
    S = np.random.default_rng(20240607).standard_normal((d, 64))[row0:row0 + d_local]        # make_golden_fullsize.sketch_matrix
    cps = [int(k) for k in g["checkpoints"] if int(k) <= T][:2]
    rel = lambda a, b_: float(np.max(np.abs(np.asarray(a) - np.asarray(b_))) / np.max(np.abs(b_)))
    worst, identical, k_prev = {}, True, 0
    reset()
    for k in cps:
        f.run(k_prev, k)
        k_prev = k
        sdev = f.get_state()
        digests = [None] * world
        dist.all_gather_object(digests, _digest(sdev["V"], sdev["P"], sdev["mu"], [sdev["eta"], sdev["N"], sdev["rho"], sdev["lam"]]))
        identical = identical and len(set(digests)) == 1
        sk = torch.from_numpy(np.ascontiguousarray(S.T @ sdev["C"]))
        dist.all_reduce(sk, op=dist.ReduceOp.SUM)
        p = f"k{k}_"
        errs = dict(V=rel(sdev["V"], g[p + "V"]), P=rel(sdev["P"], g[p + "P"]), mu=rel(sdev["mu"], g[p + "mu"]), eta=rel(sdev["eta"], g[p + "eta"]),
                    N=rel(sdev["N"], g[p + "N"]), StC=rel(sk.numpy(), g[p + "StC"]))
        for name, e in errs.items():
            worst[name] = max(worst.get(name, 0.0), e)
    tm = torch.tensor([worst[n] for n in sorted(worst)], dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    out = dict(zip(sorted(worst), [float(v) for v in tm]))
    out.update(checkpoints=cps, ranks=world, replicated_state_bit_identical=bool(identical), against=os.path.relpath(path, ROOT),
               ok=bool(identical and max(float(v) for v in tm) < 1e-5))
    return out


def sharded_parity_vs_oracle(f, args, series, st0, d, r, T, seed, row0, d_local, world, dist, reset):
    """Sharded parity at small sizes (rehearsals, tests): every rank runs the oracle on the WHOLE problem -- the series is the
    concatenation of the shards' series -- and checks its own rows of C and the replicated V, mu, P; worst over the ranks."""
    import torch

    from oracle import psmf_oracle as O
    from rpsmf_amd.sharding import shard_rows

    n_cpu = min(args.cpu_steps, T)
    gn = args.scaling == "strong"
    parts = [Series(d, r, T, seed, *shard_rows(d, world, q), bool(args.robust), global_noise=gn) for q in range(world)]
    Yfull = np.hstack([np.vstack([Yc for _, Yc in p_.chunks(chunk=n_cpu)][:1])[:n_cpu] for p_ in parts]).astype(np.float64)
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=st0["P"].copy(), Q=st0["Q"].copy(), rho=st0["rho"], lam=st0["lam"])
    st, _, _ = O.run_epoch(st, Yfull, O.Mode(robust=bool(args.robust)), O.RandomWalkDyn(), want_grad=False)
    reset()
    f.run(0, n_cpu)
    sdev = f.get_state()
    rel = lambda a, b_: float(np.max(np.abs(a - b_)) / np.max(np.abs(b_)))
    mine = np.array([rel(sdev["C"], st.C[row0:row0 + d_local]), rel(sdev["V"], st.V), rel(sdev["mu"], st.mu), rel(sdev["P"], st.P)])
    tm = torch.from_numpy(mine)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    digests = [None] * world
    dist.all_gather_object(digests, _digest(sdev["V"], sdev["P"], sdev["mu"]))
    return dict(steps=n_cpu, ranks=world, C=float(tm[0]), V=float(tm[1]), mu=float(tm[2]), P=float(tm[3]),
                replicated_state_bit_identical=len(set(digests)) == 1)


def exchange_info(world, args, per_rank, geo, r):
    """config.exchange: None for a plain N = 1 run, else what carried the sum over the row shards and how many ranks it spanned."""
    c0 = per_rank[0]["comm"]
    if c0["transport"] is None:
        return None
    if geo["engine"] == "block":
        msgs = {"first_block_gram_bytes": 64 * 64 * 8, "cross_gram_per_block_bytes": 128 * 64 * 8, "blocks_per_pass": -(-args.T // geo["block_steps"]),
                "where": "bulk stream, one block ahead of its use (off the filter chain's critical path)"}
    else:
        msgs = {"per_timestep_bytes": (r + 1) * 8, "where": "between the row sweep and the serial stage of every timestep"}
    return {"transport": "RCCL all-reduce (ncclDouble, sum)" if c0["transport"] == "rccl" else "host-mediated all-reduce over gloo (rehearsal transport)",
            "rccl_ranks": c0["ranks"] if c0["transport"] == "rccl" else None,        # ncclCommCount of rank 0's communicator
            "ranks": [p["comm"]["ranks"] for p in per_rank], "world_size": world,
            "rank_devices": [{"rank": p["rank"], "hip_device": p["device"], "pci_bus_id": p["pci_bus_id"]} for p in per_rank],
            "distinct_gpus": len({p["pci_bus_id"] for p in per_rank}), "messages": msgs,
            "forced_single_rank": bool(os.environ.get("PSMF_FORCE_COLLECTIVE")) and world == 1,
            "rccl_init_error": per_rank[0].get("rccl_error")}          # not None: RCCL was asked for and failed; the transport above is the fallback


def load_pmc(name):
    p = os.path.join(ROOT, "profiles", name)
    return json.load(open(p)) if os.path.exists(p) else None


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Libraries underneath write there too (gloo's "Rank 0 is connected to ..." line, RCCL's
    # version banner at communicator creation -- from C, straight to file descriptor 1): everything but the final line goes to stderr.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist  # rendezvous plumbing only (gloo, CPU tensors)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from rpsmf_amd import _capi
    from rpsmf_amd.sharding import shard_rows

    d, r, T = args.d, args.r, args.T
    if args.scaling == "weak":
        d = args.d * world              # per-GPU work fixed: every rank holds args.d rows of a filter with N x as many
    if args.pid_dir:
        with open(os.path.join(args.pid_dir, f"rank{rank}.pid"), "w") as fp:
            fp.write(str(os.getpid()))
    row0, d_local = shard_rows(d, world, rank)
    seed = 35833 if args.robust else 35853  # Makefile:55,64 of the reference
    series = Series(d, r, T, seed, row0, d_local, bool(args.robust), global_noise=(args.scaling == "strong"))
    st0 = init_state(d, r, seed)

    if args.one_device:
        local_rank = 0      # (with --comm rccl RCCL refuses the second rank on the device: that run rehearses the fallback below)
    def make_filter():
        return _capi.DeviceFilter(d, r, robust=bool(args.robust), storage=args.storage, store_y_pred=not args.no_y_pred,
                                  device=local_rank, row0=row0, d_local=d_local, n_workgroups=args.workgroups, engine=args.engine)

    def host_allreduce(v):          # same bits on every rank (gloo reduces in a fixed order)
        import torch

        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    f = make_filter()
    rccl_error = None
    if world > 1 and args.comm == "gloo":
        f.comm_init_host(world, rank, host_allreduce)
    elif world > 1:
        import torch

        if rank == 0:
            uid = np.frombuffer(_capi.DeviceFilter.comm_unique_id(), dtype=np.uint8).copy()
        else:
            uid = np.zeros(_capi.UNIQUE_ID_BYTES, dtype=np.uint8)
        t = torch.from_numpy(uid)
        dist.broadcast(t, 0)
        # ncclCommInitRank is a collective: a peer that never calls it leaves every rank waiting inside RCCL, where no Python
        # timeout reaches.  The call runs in a helper thread (ctypes releases the GIL); if it has not returned within the bound
        # this rank says why and ends the process with a non-zero code -- a hung start must never look like a slow one.
        import threading

        init_box = {}

        def _init():
            try:
                f.comm_init(world, rank, t.numpy().tobytes())
            except Exception as e:      # noqa: BLE001 -- whatever RCCL reports, every rank has to learn of it
                init_box["err"] = f"rank {rank}: {e}"

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(float(os.environ.get("PSMF_COMM_INIT_TIMEOUT", "240")))
        if th.is_alive():
            print(f"[bench] rank {rank}: RCCL communicator initialisation (ncclCommInitRank + warm-up all-reduces over {world} ranks) "
                  f"did not return within {os.environ.get('PSMF_COMM_INIT_TIMEOUT', '240')} s; a peer is missing or the fabric is down. "
                  "No number is reported.", file=sys.stderr, flush=True)
            os._exit(3)
        rccl_error = init_box.get("err")
        errs = [None] * world
        dist.all_gather_object(errs, rccl_error)
        if any(errs):
            # RCCL did not come up on every rank: rather than lose the N-GPU measurement, carry the r-sized sums over the host
            # (gloo) -- a different, slower transport, and config.exchange says so with RCCL's own message
            rccl_error = "; ".join(e for e in errs if e)
            if rank == 0:
                print(f"[bench] RCCL communicator failed ({rccl_error}); falling back to the host-mediated all-reduce over gloo", file=sys.stderr, flush=True)
            try:
                f.comm_abort()          # a communicator whose peers never joined: destroying it could wait for them
            except Exception:           # noqa: BLE001
                pass
            f.close()
            f = make_filter()
            f.comm_init_host(world, rank, host_allreduce)

    elif os.environ.get("PSMF_FORCE_COLLECTIVE"):
        # N = 1 with the RCCL calls of the sharded engine in the loop (a 1-rank communicator: the all-reduce kernels really sit on
        # the bulk stream) -- what every rank of an N > 1 run does besides waiting for its peers
        f.comm_init(1, 0, _capi.DeviceFilter.comm_unique_id())

    for a, Yc in series.chunks():
        f.upload_series(Yc, t0=a, T_total=T)

    def reset():
        f.set_state(st0["C"][row0:row0 + d_local], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"],
                    lambda0=st0["lam"])

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch

        te = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        return float(te.item())

    # ---- CPU baseline + in-run parity of the GPU path against it (rank 0, N = 1 only)
    cpu = None
    parity = None
    if world == 1 and args.cpu_steps > 0:
        cpu, st_cpu, n_cpu = cpu_baseline(args, series, st0)
        reset()
        f.run(0, n_cpu)
        parity = parity_of(f, st_cpu, n_cpu)
    elif world > 1 and args.cpu_steps > 0:
        fixture = args.parity_fixture
        if fixture is None and (d, r, T, args.scaling) == (100_000, 32, 10_000, "strong"):
            fixture = os.path.join(ROOT, "tests", "golden", f"fullsize_E_{'rpsmf' if args.robust else 'psmf'}.npz")
        if fixture is not None and os.path.exists(fixture):
            parity = sharded_parity_vs_fixture(f, fixture, d, r, T, seed, bool(args.robust), row0, d_local, world, dist, reset)
        elif d * r <= 1_000_000:
            parity = sharded_parity_vs_oracle(f, args, series, st0, d, r, T, seed, row0, d_local, world, dist, reset)

    # ---- untimed pre-warm: the HIP runtime grows its signal / kernel-argument pools the first time a whole pass worth of
    # launches is queued ahead of the GPU (a one-off ~80 ms stall in the second pass, tools/probe_stall.py)
    reset()
    for _ in range(2):
        f.run(0, T, sync=False)
    f.sync()
    # ---- config E as literally stated: ONE pass of T timesteps from the initial state, timed on its own
    # (twice, each from the initial state: a single 40 ms measurement is at the mercy of one host hiccup; both are reported)
    cold_runs = []
    for _ in range(2):
        reset()
        f.sync()
        barrier()
        t0 = time.perf_counter()
        f.run(0, T, sync=False)
        f.sync()
        barrier()
        cold_runs.append(max_over_ranks(time.perf_counter() - t0))
    cold_elapsed = min(cold_runs)
    # ---- timed region: K passes of T timesteps, state carried from pass to pass
    for _ in range(args.warmup):
        f.run(0, T, sync=False)
    f.sync()
    if geo_engine_block(f):
        f.counters(reset=True)
        f.filter_kernel_time(reset=True)
    barrier()
    if args.pid_dir:
        open(os.path.join(args.pid_dir, f"rank{rank}.timed"), "w").close()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f.run(0, T, sync=False)
    f.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    insitu = f.counters() if geo_engine_block(f) else None      # device-timer durations of the filter kernels of the timed region
    chained = f.filter_kernel_time() if geo_engine_block(f) else (0, 0.0)   # HIP events around the chained filter launches of the timed region
    elapsed = max_over_ranks(elapsed)
    value = args.steps * T / elapsed
    # ---- what the exchange really was: the communicator as RCCL reports it, which GPU every rank drives, and per rank the time
    # between consecutive blocks of the chained filter launch (an all-reduce that landed on the critical path shows there)
    cinfo = f.comm_info()
    mine = dict(rank=rank, device=local_rank, pci_bus_id=_capi.device_pci_bus_id(local_rank), comm=cinfo, rccl_error=rccl_error,
                gap_between_blocks_us=insitu["filter_gap_us_mean"] if insitu else None,
                block_us_in_kernel=insitu["filter_us_mean"] if insitu and insitu["filter_launches"] > 0 else None)
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # ---- roofline of the dominant kernel, HIP events on the library's stream (psmf_time_kernel)
    es = 4.0 if args.storage == "f32" else 8.0
    bytes_per_step = 2.0 * es * d_local * (r + 1)   # read C, write C, read y, write y_hat
    geo = f.geometry()
    traffic = None
    if geo["engine"] == "block":
        # one launch of the coefficient-space filter kernel = every block of a pass (chained); the two d-sized
        # products of each block (cross-Gram, apply) run concurrently on a second stream
        B = geo["block_steps"]
        t_filter = f.time_kernel(0, 20)
        t_gram = f.time_kernel(1, 50)
        t_apply = f.time_kernel(2, 50)
        t_block_insitu = insitu["filter_us_mean"] if insitu and insitu["filter_launches"] > 0 else t_filter
        if chained[0] > 0:
            # HIP events around every chained launch of the timed region, on the stream it runs on: reported as measured (the host
            # wall time of a pass is beside it as ms_per_step; the two clocks differ by ~0.2 %)
            kernel, kernel_us, steps_per_launch = geo.get("filter_kernel", "psmf_blk_filter3"), 1e3 * chained[1] / chained[0], float(T)
        else:
            full_blocks_only = (T % B == 0)
            kernel, kernel_us = geo.get("filter_kernel", "psmf_blk_filter3"), t_block_insitu
            steps_per_launch = B if full_blocks_only else T / (insitu["filter_launches"] / args.steps)
        zbytes = es * d_local * 64
        nb1 = min(B, 64)
        xg_bytes = es * d_local * (r + B + nb1)          # cross-Gram reads C, the current and the next series block
        ap_bytes = 2.0 * es * d_local * (r + B)          # apply reads C, Y_cur and writes C, Y_hat
        pmc_name = next((n for n in ("r4_pmc_traffic_block_engine.json", "r3_pmc_traffic_block_engine.json", "r2_pmc_traffic_block_engine.json", "r1_pmc_traffic_block_engine.json")
                         if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        pmc = load_pmc(pmc_name) if pmc_name else None
        real_hbm = None
        traffic_source = None
        if (d, r, args.storage, world) == (100_000, 32, "f32", 1) and pmc:
            # rocprofv3 --pmc passes of this workload (counter collection serialises the kernels; the library then runs
            # one filter launch per block): bytes per block of B timesteps, scaled to the blocks one launch advances
            traffic = pmc["traffic_bytes_per_launch"] * (steps_per_launch / B)
            traffic_source = (f"profiles/{pmc_name} (rocprofv3 --pmc passes of this workload, committed; NOT measured in this run; counter "
                              "collection serialises kernels, so those passes ran the per-block schedule -- one filter launch per block, event "
                              "hand-off -- not the chained launch timed here: same kernels and bytes per block, different launch structure)")
            if "all_kernels_bytes_per_block" in pmc:
                real_hbm = pmc["all_kernels_bytes_per_block"] * (T / B) / (elapsed / args.steps) / 1e9
        steps_timed = max(1, (insitu["ns_steps"] + insitu["sweep_steps"])) if insitu else 1
        iters_per_step = insitu["ns_iterations"] / steps_timed if insitu else None
        step_us = t_block_insitu / B
        extra = {"steps_per_launch": steps_per_launch,
                 "blocks_per_launch": steps_per_launch / B,
                 "block_us_in_kernel": t_block_insitu,
                 "one_block_us_hip_events_standalone": t_filter,
                 "gap_between_blocks_us": insitu["filter_gap_us_mean"] if insitu else None,
                 "gap_between_blocks_us_per_rank": [p["gap_between_blocks_us"] for p in per_rank],
                 "block_us_in_kernel_per_rank": [p["block_us_in_kernel"] for p in per_rank],
                 "real_hbm_GBps": real_hbm,
                 "traffic_source": traffic_source,
                 # the blocked formulation's own minimum per block of B timesteps: cross-Gram reads C, Y_cur, Y_next; apply reads
                 # C, Y_cur and writes C, Y_hat (f32) -- against what the counters say the kernels of a block move
                 "blocked_min_bytes_per_block": xg_bytes + ap_bytes,
                 "blocked_traffic_over_min": (pmc["all_kernels_bytes_per_block"] / (xg_bytes + ap_bytes))
                                             if (traffic_source and pmc and "all_kernels_bytes_per_block" in pmc) else None,
                 "filter_chain_frac": (iters_per_step * NS_ITER_ISSUE_US / step_us) if iters_per_step else None,
                 "filter_chain": {"newton_schulz_iterations_per_timestep": iters_per_step, "mfma_issue_us_per_iteration": NS_ITER_ISSUE_US,
                                  "us_per_timestep_in_kernel": step_us,
                                  "timesteps_by_direct_sweep": insitu["sweep_steps"] if insitu else None},
                 "kernels_us": {"psmf_blk_filter3": t_filter, "psmf_blk_xgram2+xreduce2": t_gram, "psmf_blk_apply2": t_apply},
                 "bulk_kernels": {"psmf_blk_xgram2+xreduce2": {"bytes": xg_bytes, "us": t_gram, "GBps": xg_bytes / (t_gram * 1e-6) / 1e9,
                                                              "frac_of_hbm_peak": xg_bytes / (t_gram * 1e-6) / 1e9 / HBM_PEAK_GBS},
                                  "psmf_blk_apply2": {"bytes": ap_bytes, "us": t_apply, "GBps": ap_bytes / (t_apply * 1e-6) / 1e9,
                                                      "frac_of_hbm_peak": ap_bytes / (t_apply * 1e-6) / 1e9 / HBM_PEAK_GBS}},
                 "note": "filter_chain_frac is the distance of the timed kernel to ITS bound: matrix-core issue time of the Newton-Schulz products of a "
                         "timestep / measured time per timestep inside the filter kernel (one workgroup, a chain of r x r stages bound by its "
                         "instruction stream on one CU).  frac = SURVEY 8(d) bookkeeping: step-at-a-time algorithmic bytes of the timesteps one "
                         "launch advances / its duration / 8 TB/s -- the blocked engine does not move those bytes (traffic, real_hbm_GBps, "
                         "blocked_min_bytes_per_block); the kernels that do stream the data are under bulk_kernels"}
    elif geo.get("filter_kernel") == "psmf_pstep_k":
        # per-step engine as ONE persistent launch per pass (psmf_pstep.hip): HIP events around a pass on the stream it runs on
        kernel, steps_per_launch = "psmf_pstep_k", float(T)
        kernel_us = 1e3 * min(f.run_timed(0, T) for _ in range(2))
        extra = {"steps_per_launch": steps_per_launch, "us_per_timestep_in_kernel": kernel_us / T,
                 "note": "C stays on chip (float64, in the row workgroups' registers) for the whole launch: the kernel moves y and y_hat only; "
                         "frac is the SURVEY 8(d) bookkeeping ratio (step-at-a-time bytes / duration / 8 TB/s), not a bandwidth -- what bounds "
                         "the kernel is the latency of its two hand-offs and of the r x r stage per timestep (docs/MEASUREMENTS.md)"}
    else:
        kernel, kernel_us, steps_per_launch = "psmf_sweep_solve", f.time_kernel(0, 300), 1
        extra = {"steps_per_launch": 1, "kernels_us": {"psmf_sweep_solve": kernel_us, "psmf_serial": f.time_kernel(1, 300)}}
        pmc = load_pmc("r1_pmc_traffic.json")
        if (d, r, args.storage, world) == (100_000, 32, "f32", 1) and pmc:
            traffic = pmc["traffic_bytes_per_launch"]   # rocprofv3 --pmc passes of this workload
    alg_bytes = bytes_per_step * steps_per_launch
    achieved = alg_bytes / (kernel_us * 1e-6) / 1e9
    f.close()

    if rank == 0:
        line = {
            "metric": "PSMF filter timesteps/sec at d=100k r=32" if (d, r) == (100_000, 32) else f"PSMF filter timesteps/sec at d={d} r={r}",
            "value": value,
            "unit": "timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",      # the arithmetic type: float64 accumulation of every d -> r contraction and float64 r x r state
            "storage_dtype": args.storage,   # C, y, y_hat in HBM (f32: one rounding of C per block)
            "data": "synthetic",
            "config": {"workload": f"{'rPSMF' if args.robust else 'PSMF'} full filter, random-walk dynamics, d={d} r={r} "
                                   f"T={T} synthetic Gaussian series, rows sharded over {world} GPU(s)"
                                   + (f" ({args.d} rows per GPU: weak scaling, d = N x {args.d})" if args.scaling == "weak" else ""),
                       "d": d, "d_per_gpu": d_local, "r": r, "T": T, "timesteps_per_pass": T, "store_y_pred": not args.no_y_pred,
                       "exchange": exchange_info(world, args, per_rank, geo, r),
                       "us_per_timestep": 1e6 * elapsed / (args.steps * T), "engine": geo["engine"], "geometry": geo},
            # BASELINE config E as literally stated -- ONE pass of T timesteps from the initial state -- next to `value`, which is
            # the steady state of carried-state passes (epochs 2, 3, ... of PSMFIter.run)
            "value_config_E_literal": T / cold_elapsed,
            "cold_pass_steps_per_s": T / cold_elapsed,
            "cold_pass_runs_steps_per_s": [T / c for c in cold_runs],
            "roofline": dict({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel, "kernel_us": kernel_us,
                              "algorithmic_bytes_per_launch": alg_bytes,
                              "whole_job_frac": value * 2.0 * es * d * (r + 1) / (world * HBM_PEAK_GBS * 1e9)}, **extra),
        }
        if world > 1:
            # an N-GPU number carried over the host path must not be mistaken for an RCCL number
            line["transport_fallback"] = bool(per_rank[0].get("rccl_error")) or args.comm != "rccl"
            line["scaling_note"] = ("strong scaling at fixed d is flat BY DESIGN under the blocked engine: every rank repeats the r x r chain "
                                    "(the critical path) on all-reduced sums and shares only the d-sized products, which are hidden behind it at "
                                    "d <= 2.5e5 rows per GPU; whole_job_frac (value x step-at-a-time bytes / (N x 8 TB/s)) therefore falls as 1/N. "
                                    "More GPUs buy more ROWS at the same speed (--scaling weak).")
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if parity is not None:
            line["parity_vs_cpu_oracle"] = parity
        if world == 1 and not args.no_extras:
            try:
                line["roofline"]["hbm_peak_measured"] = {"GBps": _capi.measure_copy_bandwidth(local_rank, 1 << 32, 10),
                                                         "what": "copy kernel, 4 GiB read + 4 GiB written per launch, 16-byte accesses"}
            except Exception as e:      # never lose the headline over an extra
                line["roofline"]["hbm_peak_measured"] = {"error": repr(e)}
            if args.cpu_steps > 0:
                line["cpu_baseline_literal"] = cpu_baseline_literal(r, seed)
                try:
                    line["cpu_baseline_f32"] = cpu_baseline_f32(args, series, st0)
                except Exception as e:
                    line["cpu_baseline_f32"] = {"error": repr(e)}
                other = {}
                e_oracle = (st_cpu, n_cpu, n_cpu / cpu["value"]) if cpu is not None else None
                if (d, r, args.storage, args.engine) == (100_000, 32, "f32", "auto"):
                    try:     # the headline with float64 storage of C, y, y_hat on the same (blocked) engine: what the 1e-5 bar costs in either norm
                        other["E_f64_storage"] = run_filter_config(_capi, "E_f64", d, r, 4_000, bool(args.robust), passes=2, storage="f64", bulk_times=True, oracle=e_oracle)
                    except Exception as e:
                        other["E_f64_storage"] = {"error": repr(e)}
                    try:     # the step-at-a-time engine north_star describes, as one persistent launch per pass (C on chip in float64)
                        other["E_per_step_engine"] = run_filter_config(_capi, "E_step", d, r, 4_000, bool(args.robust), passes=2, storage="f32", engine="step", oracle=e_oracle)
                    except Exception as e:
                        other["E_per_step_engine"] = {"error": repr(e)}
                    try:     # ... and its masked form at the same shape
                        other["E_masked_per_step_engine"] = run_masked_config(_capi, d, r)
                    except Exception as e:
                        other["E_masked_per_step_engine"] = {"error": repr(e)}
                for name, rob in (("B", False), ("C", True)):
                    try:
                        other[name] = run_filter_config(_capi, name, 10_000, 20, 5_000, rob)
                    except Exception as e:
                        other[name] = {"error": repr(e)}
                try:     # the default model at a small rank (r <= 16: psmf_blk_filter6d, two sweep inversions side by side)
                    other["small_rank_r12"] = run_filter_config(_capi, "r12", 10_000, 12, 5_000, False)
                except Exception as e:
                    other["small_rank_r12"] = {"error": repr(e)}
                try:     # beyond the blocked engine's r <= 32: the persistent per-step kernel with the hub's matrices in LDS (33 <= r <= 48)
                    other["rank_40_per_step_engine"] = run_filter_config(_capi, "r40", 20_000, 40, 3_000, False, passes=2, storage="f64")
                except Exception as e:
                    other["rank_40_per_step_engine"] = {"error": repr(e)}
                try:
                    other["ExperimentSynthetic_hooks"] = run_synthetic_simplified(_capi)
                except Exception as e:
                    other["ExperimentSynthetic_hooks"] = {"error": repr(e)}
                try:
                    other["ExperimentBeijing_dynamics"] = run_fourier_config(_capi)
                except Exception as e:
                    other["ExperimentBeijing_dynamics"] = {"error": repr(e)}
                try:
                    other["D"] = run_impute_config()
                except Exception as e:
                    other["D"] = {"error": repr(e)}
                try:     # the reference's own batch: IMPUTE_REPEATS = 100 (Makefile:160,176-177) -- one workgroup per replica, 100 CUs busy
                    other["D_100_repeats"] = run_impute_config(seeds=100, variants=(False,))
                except Exception as e:
                    other["D_100_repeats"] = {"error": repr(e)}
                line["other_configs"] = other
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
