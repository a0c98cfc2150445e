import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from adversarial_cases import make_case
from oracle import psmf_oracle as O
from rpsmf_amd import _capi
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
D, T = 10000, 1000
for name, r, robust in (("tiny_Q", 32, False), ("tiny_Q", 44, True)):
    cs = make_case(name, D, r, T, robust)
    st = O.State(C=cs["C0"].copy(), V=cs["V0"], mu=np.zeros(r), P=cs["P0"], Q=cs["Q"], rho=1.0, lam=1.8)
    st, Yp, trace = O.run_epoch(st, cs["Y"].astype(np.float64), O.Mode(robust=robust), O.RandomWalkDyn(), keep=cs["checkpoints"], want_grad=False)
    for env in ({}, {"PSMF_STEP_DUAL": "0"}, {"PSMF_STEP_PERSISTENT": "0"}, {"PSMF_STEP_PERSISTENT": "0", "PSMF_STEP_DUAL": "0"}):
        for k in ("PSMF_STEP_DUAL", "PSMF_STEP_PERSISTENT"): os.environ.pop(k, None)
        os.environ.update(env)
        f = _capi.DeviceFilter(D, r, robust=robust, storage="f64", engine="step")
        f.upload_series(cs["Y"]); f.set_state(cs["C0"], cs["V0"], cs["P0"], cs["Q"], np.zeros(r), rho=1.0, lambda0=1.8)
        kern = f.geometry()["filter_kernel"]; kp = 0; worst = {}
        for k in cs["checkpoints"]:
            f.run(kp, k); s = f.get_state(); ref = trace[k][0]
            worst[k] = max(rel(s[n], getattr(ref, n)) for n in ("C", "V", "mu", "P")); kp = k
        f.close()
        print(name, r, robust, env, kern, {k: "%.1e" % v for k, v in worst.items()}, flush=True)
