"""GPU-box probe: the persistent per-step kernel at 33 <= r <= 48 (hub with LDS-resident matrices) against the launched form."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpsmf_amd import _capi as c

def run(d, r, T, storage, robust, persistent, Y, C0, cuts=None):
    os.environ["PSMF_STEP_PERSISTENT"] = "1" if persistent else "0"
    f = c.DeviceFilter(d, r, storage=storage, robust=robust, engine="step")
    f.upload_series(Y)
    f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
    kern = f.geometry()["filter_kernel"]
    try:
        for a, b in zip((cuts or (0, T))[:-1], (cuts or (0, T))[1:]):
            f.run(a, b)
        s = f.get_state(); s["yp"] = f.y_pred(0, T); err = None
    except Exception as e:
        s, err = None, repr(e)[:100]
    best = None
    if err is None and T >= 200:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); f.run(0, T); best = min(best, (time.perf_counter() - t0) / T)
    f.close()
    return s, err, kern, best

rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
rng = np.random.default_rng(0)
for (d, r, T, storage, robust) in [(948, 48, 12, "f64", False), (947, 47, 12, "f64", False), (944, 44, 12, "f64", True), (933, 33, 12, "f64", False),
                                   (20000, 40, 400, "f64", False), (20000, 48, 400, "f64", False), (20000, 40, 400, "f32", True), (5000, 36, 400, "f64", False),
                                   (100000, 40, 300, "f32", False), (100000, 48, 300, "f64", True), (70000, 33, 300, "f32", False)]:
    Y = rng.standard_normal((T, d)).astype(np.float32).astype(np.float64)
    C0 = (0.1 * rng.standard_normal((d, r))).astype(np.float32).astype(np.float64)
    a, ea, ka, ta = run(d, r, T, storage, robust, True, Y, C0, cuts=(0, T // 3, T))
    b, eb, kb, tb = run(d, r, T, storage, robust, False, Y, C0)
    out = dict(d=d, r=r, T=T, storage=storage, robust=robust, kern=(ka, kb), err=(ea, eb))
    if a and b:
        out["rel"] = {k: rel(a[k], b[k]) for k in ("C", "V", "mu", "P", "yp")}
    if ta: out["us"] = (round(1e6 * ta, 2), round(1e6 * tb, 2))
    print(out, flush=True)
