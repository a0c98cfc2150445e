"""Diagnostic: device dynamics kinds at tiny r / d against the oracle (per checkpoint)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import psmf_oracle as O
from rpsmf_amd import _capi as c, nonlinearities as NL
from conftest import relerr

def run(nl, d, r, T, cps, env=None):
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((T, d)); C0 = rng.standard_normal((d, r))
    theta = 0.1 * rng.random(nl.n_params)
    V0, P0, Q = 5 * np.eye(r), np.eye(r), np.eye(r)
    dyn = O.CallableDyn(nl, nl.n_params)
    st = O.State(C=C0, V=V0, mu=np.zeros(r), P=P0, Q=Q, rho=1.0, lam=0.0, theta=theta.copy(), gradsum=np.zeros(nl.n_params))
    st, Yp, tr = O.run_epoch(st, Y, O.Mode(), dyn, keep=cps)
    f = c.DeviceFilter(d, r, storage="f64", dyn_kind=nl.device_kind, dyn_flags=nl.device_flags, dyn_terms=nl.device_terms)
    f.upload_series(Y); f.set_state(C0, V0, P0, Q, np.zeros(r), rho=1.0, lambda0=0.0, theta=theta); f.zero_gradsum()
    kp = 0
    for k in cps:
        f.run(kp, k); kp = k
        s = f.get_state(); ref = tr[k][0]
        print(type(nl).__name__, "d", d, "r", r, "k", k, {n: "%.1e" % relerr(s[n], getattr(ref, n)) for n in ("C", "V", "mu", "P", "gradsum")}, flush=True)
    f.close()

for r in (1, 2, 3):
    run(NL.FourierBasis(r, 1), 3, r, 60, (1, 2, 10, 48, 49, 60))
run(NL.FourierBasis(1, 1), 40, 1, 60, (1, 10, 48, 60))
run(NL.Sinusoid(1), 3, 1, 60, (1, 10, 48, 60))
run(NL.CosPhase(1), 3, 1, 60, (1, 10, 48, 60))
