"""GPU-box probe: filter4 with Newton-Schulz off (direct sweeps only) vs on, hard regime (cos-phase, d = 2e4, q = 0.1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench
T = 2000
for r in (20, 32, 12):
  for env in ({"PSMF_NS": "0"}, {}, {"PSMF_FILTER4": "0"}):
    for k in ("PSMF_NS", "PSMF_FILTER4"):
        os.environ.pop(k, None)
    os.environ.update(env)
    d = 20000
    ser = bench.Series(d, r, T, 4711, 0, d, False)
    st0 = bench.init_state(d, r, 4711)
    f = _capi.DeviceFilter(d, r, storage="f32", dyn_kind=_capi.DYN_COS_PHASE)
    for a, Yc in ser.chunks():
        f.upload_series(Yc, t0=a, T_total=T)
    theta = 0.05 + 0.1 * np.random.default_rng(3).random(r)
    f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta)
    out = []
    for i in range(4):
        f.counters(reset=True)
        t0 = time.perf_counter(); f.run(0, T); dt = time.perf_counter() - t0
        c = f.counters()
        out.append(f"{1e6 * dt / T:.2f}us {c['ns_steps']}/{c['sweep_steps']}/{c['ns_iterations']}/{c['ns_failed']}")
    print(f"r={r}", env, f.geometry()["filter_kernel"], " | ".join(out), flush=True)
    f.close()
