"""GPU-box probe: which r x r inversion path the blocked engine takes on the bench workloads (not a test)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi
import bench

def run(d, r, T, robust):
    seed = 35833 if robust else 35853
    ser = bench.Series(d, r, T, seed, 0, d, bool(robust))
    st0 = bench.init_state(d, r, seed)
    f = _capi.DeviceFilter(d, r, robust=bool(robust), storage="f32")
    for a, Yc in ser.chunks():
        f.upload_series(Yc, t0=a, T_total=T)
    f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"], lambda0=st0["lam"])
    out = []
    f.counters(reset=True)
    for a in range(0, T, 500):
        ms = f.run_timed(a, min(T, a + 500))
        c = f.counters(reset=True)
        c.update(k0=a, us_per_step=1e3 * ms / (min(T, a + 500) - a))
        out.append(c)
    # second epoch over the same series (state carried)
    ms = f.run_timed(0, T)
    c = f.counters(reset=True); c.update(k0="epoch2", us_per_step=1e3 * ms / T); out.append(c)
    f.close()
    return out

if __name__ == "__main__":
    for (d, r, T, rob) in ((100000, 32, 3000, 0), (10000, 20, 3000, 0), (10000, 20, 3000, 1)):
        print(d, r, T, rob)
        for c in run(d, r, T, rob):
            print("  ", json.dumps(c), flush=True)
