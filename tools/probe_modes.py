"""GPU-box probe: timesteps/s of the device filter in its modes (which kernel runs what), d = 20 000, T = 2 000."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench

d, T = 20000, 2000
cases = [
    ("random walk r=32 (filter3)", dict(r=32)),
    ("random walk r=20 (filter3, masked)", dict(r=20)),
    ("random walk r=12 (filter6d; filter3s with PSMF_FILTER6_DUAL=0)", dict(r=12)),
    ("rPSMF r=20", dict(r=20, robust=True)),
    ("cos-phase r=20 (general blocked kernel)", dict(r=20, dyn_kind=_capi.DYN_COS_PHASE)),
    ("cos-phase simplified r=20 (ExperimentSynthetic)", dict(r=20, dyn_kind=_capi.DYN_COS_PHASE, coef_update=False, eta_full=False, pbar_predict=False)),
    ("Fourier N=2 r=10 (ExperimentBeijing)", dict(r=10, dyn_kind=_capi.DYN_FOURIER, dyn_terms=2)),
    ("recursive cos-phase r=20 (in-loop Adam)", dict(r=20, dyn_kind=_capi.DYN_COS_PHASE, recursive=True)),
    ("cos-phase r=10 full filter", dict(r=10, dyn_kind=_capi.DYN_COS_PHASE)),
    ("recursive cos-phase r=10", dict(r=10, dyn_kind=_capi.DYN_COS_PHASE, recursive=True)),
    ("Fourier N=1 r=14", dict(r=14, dyn_kind=_capi.DYN_FOURIER, dyn_terms=1)),
    ("Fourier N=2 r=10, rPSMF", dict(r=10, dyn_kind=_capi.DYN_FOURIER, dyn_terms=2, robust=True)),
    ("scaled walk r=10", dict(r=10, dyn_kind=_capi.DYN_SCALED_WALK)),
    ("recursive sinusoid r=6 (in-loop Adam)", dict(r=6, dyn_kind=_capi.DYN_SINUSOID, dyn_flags=3, recursive=True)),
    ("Fourier N=2 r=16", dict(r=16, dyn_kind=_capi.DYN_FOURIER, dyn_terms=2)),
    ("Fourier N=1 r=20 (general blocked kernel)", dict(r=20, dyn_kind=_capi.DYN_FOURIER, dyn_terms=1)),
    ("per-step engine r=32 f64", dict(r=32, engine="step", storage="f64")),
    ("per-step engine r=40 f64", dict(r=40, storage="f64")),
]
only = os.environ.get("ONLY")
if only:
    cases = [c for c in cases if only in c[0]]
for name, kw in cases:
    r = kw.pop("r")
    ser = bench.Series(d, r, T, 4711, 0, d, bool(kw.get("robust", False)))
    st0 = bench.init_state(d, r, 4711)
    kw.setdefault("storage", "f32")
    f = _capi.DeviceFilter(d, r, **kw)
    for a, Yc in ser.chunks():
        f.upload_series(Yc, t0=a, T_total=T)
    theta = None
    if f.n_theta:
        theta = 0.05 + 0.1 * np.random.default_rng(3).random(f.n_theta)
    f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta)
    best = 1e9
    try:
        cn = {}
        for i in range(3):
            f.counters(reset=True)
            t0 = time.perf_counter(); f.run(0, T); best = min(best, time.perf_counter() - t0)
            cn = f.counters() if f.geometry()["engine"] == "block" else {}
        g = f.geometry()
        extra = f" ns/sweep/iter/failed={cn['ns_steps']}/{cn['sweep_steps']}/{cn['ns_iterations']}/{cn['ns_failed']}" if cn else ""
        print(f"{name:52s} {T / best:10.0f} timesteps/s  ({1e6 * best / T:.2f} us)  {g['filter_kernel']}{extra}", flush=True)
    except Exception as e:
        print(f"{name:52s} failed: {type(e).__name__}: {e}", flush=True)
    f.close()
