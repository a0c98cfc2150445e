"""GPU-box probe: is the cos-phase full filter at d = 2e4 sensitive to f32 storage (any kernel), or is filter4 wrong?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import psmf_oracle as O
from rpsmf_amd import _capi as c
d, r, T = 20000, 20, 800
q = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
Y = O.synthetic_series(d, r, T, 35853, dtype=np.float32)
rng = np.random.default_rng(35853 + 7)
C0 = (0.1 * rng.standard_normal((d, r))).astype(np.float32).astype(np.float64)
theta0 = (1e-3 * np.arange(1, r + 1)) if (len(sys.argv) > 2 and sys.argv[2] == "true") else 0.05 + 0.1 * np.random.default_rng(77).random(r)
V0, P0, Q = 0.1 * np.eye(r), np.eye(r), q * np.eye(r)
cps = (50, 100, 200, 400, 800)
st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=1.8, theta=theta0.copy(), gradsum=np.zeros(r))
st, Yp, tr = O.run_epoch(st, Y.astype(np.float64), O.Mode(), O.CosPhaseDyn(r), keep=cps)
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
for storage, f4 in (("f64", "1"), ("f32", "1"), ("f64", "0"), ("f32", "0")):
    os.environ["PSMF_FILTER4"] = f4
    f = c.DeviceFilter(d, r, storage=storage, dyn_kind=c.DYN_COS_PHASE)
    f.upload_series(Y)
    f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=1.8, theta=theta0)
    f.zero_gradsum()
    kp = 0; out = []
    for k in cps:
        f.run(kp, k); kp = k
        s = f.get_state()
        out.append(f"k={k}: C {rel(s['C'], tr[k][0].C):.1e} mu {rel(s['mu'], tr[k][0].mu):.1e} P {rel(s['P'], tr[k][0].P):.1e}")
    print(storage, f.geometry()["filter_kernel"], " | ".join(out), "|mu|max", np.abs(s["mu"]).max(), flush=True)
    f.close()
