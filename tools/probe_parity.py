"""GPU-box probe: error of the f32 / f64 storage paths vs the float64 CPU oracle along the run."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import psmf_oracle as O
from rpsmf_amd import _capi

def go(d, r, T, marks, robust=False):
    ser = bench.Series(d, r, T, 35853, 0, d, robust)
    Y = np.vstack([y for _, y in ser.chunks(chunk=T)])
    st0 = bench.init_state(d, r, 35853)
    fs = {}
    for storage in ("f32", "f64"):
        f = _capi.DeviceFilter(d, r, storage=storage, robust=robust, engine=os.environ.get("PROBE_ENGINE", "auto"))
        f.upload_series(Y)
        f.set_state(st0["C"], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=1.0, lambda0=1.8)
        fs[storage] = f
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=st0["P"].copy(), Q=st0["Q"].copy(), rho=1.0, lam=1.8)
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    kprev = 0
    Y64 = Y.astype(np.float64)
    for k in marks:
        for j in range(kprev, k):
            st, info = O.lowrank_step(st, Y64[j], j + 1, O.Mode(robust=robust), O.RandomWalkDyn(), want_grad=False)
        out = dict(d=d, r=r, k=k, robust=robust, engine=fs["f32"].geometry()["engine"])
        for storage, f in fs.items():
            f.run(kprev, k)
            s = f.get_state()
            out[storage] = {n: "%.1e" % rel(s[n], getattr(st, n)) for n in ("C", "V", "mu", "P")}
            out[storage]["yhat"] = "%.1e" % rel(f.y_pred(k - 1, 1)[0], info.y_pred)
        kprev = k
        print(json.dumps(out), flush=True)
    for f in fs.values():
        f.close()

if __name__ == "__main__":
    go(20000, 32, 3000, (10, 100, 300, 1000, 3000))
    go(10000, 20, 2000, (100, 300, 1000, 2000), robust=True)
    go(100000, 32, 300, (100, 300))
