"""GPU-box probe: filter4 (cos-phase full filter) inversion statistics under the Newton-Schulz switches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench

d, T, r = 20000, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 20
ser = bench.Series(d, r, T, 4711, 0, d, False)
st0 = bench.init_state(d, r, 4711)
chunks = list(ser.chunks())
for env in ({}, {"PSMF_NS_PREDICT": "0"}, {"PSMF_NS_FAR": "30"}, {"PSMF_NS_FAR": "30", "PSMF_NS_PREDICT": "0"}):
    for k in ("PSMF_NS_PREDICT", "PSMF_NS_FAR"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for truth in (False, True):
        f = _capi.DeviceFilter(d, r, storage="f32", dyn_kind=_capi.DYN_COS_PHASE)
        for a, Yc in chunks:
            f.upload_series(Yc, t0=a, T_total=T)
        theta = 1e-3 * np.arange(1, r + 1) if truth else 0.05 + 0.1 * np.random.default_rng(3).random(r)
        f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta)
        out = []
        for i in range(3):
            f.counters(reset=True)
            t0 = time.perf_counter(); f.run(0, T); dt = time.perf_counter() - t0
            c = f.counters()
            out.append(f"{1e6 * dt / T:.2f}us ns/sw/it/fail={c['ns_steps']}/{c['sweep_steps']}/{c['ns_iterations']}/{c['ns_failed']}")
        s = f.get_state()
        print(env, "theta=truth" if truth else "theta=random", f.geometry()["filter_kernel"], " | ".join(out),
              f"  trP={np.trace(s['P']):.3e} minG={np.linalg.eigvalsh(s['V']).min():.2e}", flush=True)
        f.close()
