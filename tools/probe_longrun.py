"""GPU-box probe: the small-rank kernel's side-by-side inversions (psmf_blk_filter6d, a W recursion carried by sweeps) over 20 000
timesteps against filter3s (PSMF_FILTER6_DUAL=0) on the same series, float64 storage: agreement and the asymmetry of P at the end."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rpsmf_amd import _capi as c
from oracle import psmf_oracle as O
d, T = 1500, 20000
for r in (12, 16, 5):
    for robust in (False, True):
        Y = O.synthetic_series(d, r, T, 3 + r, noise="t" if robust else "normal", dtype=np.float64)
        rng = np.random.default_rng(r)
        C0 = 0.1 * rng.standard_normal((d, r))
        V0, P0, Q = 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r)
        out = {}
        for dual in ("1", "0"):
            os.environ["PSMF_FILTER6_DUAL"] = dual
            f = c.DeviceFilter(d, r, robust=robust, storage="f64", engine="block")
            f.upload_series(Y)
            f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8)
            k = f.geometry()["filter_kernel"]
            f.run(0, T)
            out[dual] = (k, f.get_state())
            f.close()
        rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
        s1, s0 = out["1"][1], out["0"][1]
        print(r, robust, out["1"][0], "vs", out["0"][0], {k: rel(s1[k], s0[k]) for k in ("C", "V", "P", "mu")}, "asym P", float(np.max(np.abs(s1["P"] - s1["P"].T)) / np.max(np.abs(s1["P"]))))
