import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from rpsmf_amd import _capi
import bench
d, r, T = 100000, 32, 3008
ser = bench.Series(d, r, T, 35853, 0, d, False); st0 = bench.init_state(d, r, 35853)
f = _capi.DeviceFilter(d, r, storage="f32")
for a, Yc in ser.chunks(): f.upload_series(Yc, t0=a, T_total=T)
f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=1.0, lambda0=1.8)
w = []
for i in range(120):
    t0 = time.perf_counter(); f.run(0, T, sync=False); f.sync(); w.append(1e6 * (time.perf_counter() - t0) / T)
med = sorted(w)[len(w) // 2]
print("wall us/step over %d passes: min %.3f median %.3f max %.3f; slow passes (index, us/step):" % (len(w), min(w), med, max(w)), [(i, round(x, 2)) for i, x in enumerate(w) if x > 1.1 * med])
