"""GPU-box probe: filter4 speed by regime (P / q): cos-phase full filter, d and q varied."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench
T, r = 2000, 20
for d, q, f4 in ((20000, 0.1, "1"), (20000, 1.0, "1"), (20000, 10.0, "1"), (100000, 0.1, "1"), (100000, 1.0, "1"), (20000, 1.0, "0"), (100000, 1.0, "0")):
    os.environ["PSMF_FILTER4"] = f4
    ser = bench.Series(d, r, T, 4711, 0, d, False)
    st0 = bench.init_state(d, r, 4711)
    f = _capi.DeviceFilter(d, r, storage="f32", dyn_kind=_capi.DYN_COS_PHASE)
    for a, Yc in ser.chunks():
        f.upload_series(Yc, t0=a, T_total=T)
    theta = 0.05 + 0.1 * np.random.default_rng(3).random(r)
    f.set_state(st0["C"], st0["V"], st0["P"], q * np.eye(r), st0["mu"], rho=st0["rho"], lambda0=st0["lam"], theta=theta)
    out = []
    for i in range(3):
        f.counters(reset=True)
        t0 = time.perf_counter(); f.run(0, T); dt = time.perf_counter() - t0
        c = f.counters()
        out.append(f"{1e6 * dt / T:.2f}us ns/sw/it/fail={c['ns_steps']}/{c['sweep_steps']}/{c['ns_iterations']}/{c['ns_failed']}")
    s = f.get_state()
    print(f"d={d} q={q}", f.geometry()["filter_kernel"], " | ".join(out), f" max eig P / q = {np.linalg.eigvalsh(s['P']).max() / q:.3f}", flush=True)
    f.close()
