"""GPU-box probe: per-kernel timings of the large-d engine for a few geometries (not a test)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpsmf_amd import _capi

def run(d, r, T, storage, coef, wg, graph=True, ypred=True):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((T, d), dtype=np.float32)
    C0 = 0.1 * rng.standard_normal((d, r))
    f = _capi.DeviceFilter(d, r, storage=storage, coef_update=coef, eta_full=coef, n_workgroups=wg, use_graph=graph, store_y_pred=ypred)
    f.upload_series(Y)
    f.set_state(C0, 0.1*np.eye(r), np.eye(r), 0.1*np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
    f.run(0, min(T, 300))
    ms = f.run_timed(0, T)
    sw = f.time_kernel(0, 200); se = f.time_kernel(1, 200)
    g = f.geometry()
    f.close()
    return dict(d=d, r=r, storage=storage, coef=coef, wg=g['n_sweep_wg'], rows=g['rows_per_wg'], us_per_step=1e3*ms/T, sweep_us=sw, serial_us=se, graph=graph)

if __name__ == "__main__":
    T = 2000
    for (d, r) in ((100000, 32), (10000, 20)):
        for storage in ("f32", "f64"):
            for coef in (True, False):
                for wg in (256, 512, 1024):
                    print(json.dumps(run(d, r, T, storage, coef, wg)), flush=True)
    print(json.dumps(run(100000, 32, T, "f32", True, 512, graph=False)), flush=True)
