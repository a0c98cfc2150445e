"""GPU-box probe: BASELINE configs B and C (d=10000, r=20, T=5000, PSMF / rPSMF), several passes each, per-pass times."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench

for rob in (0, 1):
    d, r, T = 10000, 20, 5000
    seed = 35833 if rob else 35853
    ser = bench.Series(d, r, T, seed, 0, d, bool(rob))
    st0 = bench.init_state(d, r, seed)
    f = _capi.DeviceFilter(d, r, robust=bool(rob), storage="f32")
    for a, Yc in ser.chunks():
        f.upload_series(Yc, t0=a, T_total=T)
    f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
    times = []
    for i in range(12):
        t0 = time.perf_counter(); ms = f.run_timed(0, T); wall = time.perf_counter() - t0
        times.append((round(1e3 * ms / T, 2), round(1e6 * wall / T, 2)))
    print("robust", rob, "us/step (event, wall):", times, f.counters(reset=True), flush=True)
    f.close()
