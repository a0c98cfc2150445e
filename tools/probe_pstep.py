"""GPU-box probe (not a test): the persistent per-step kernel (psmf_pstep.hip) against the float64 oracle and against the two-launch
per-step engine (PSMF_STEP_PERSISTENT=0), then us per timestep of both at a few shapes.

    python tools/probe_pstep.py [quick]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import psmf_oracle as O
from rpsmf_amd import _capi as c


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def make(d, r, T, seed, robust):
    rng = np.random.default_rng(seed)
    Ct = rng.standard_normal((d, r))
    x = rng.standard_normal(r)
    Y = np.empty((T, d))
    for t in range(T):
        x = x + 0.1 * rng.standard_normal(r)
        Y[t] = Ct @ x + 0.3 * (rng.standard_t(3.0, d) if robust else rng.standard_normal(d))
    C0 = 0.1 * rng.standard_normal((d, r))
    return Y, C0


def run_dev(persistent, d, r, Y, C0, Q, robust, storage, cuts, dyn=c.DYN_RANDOM_WALK, theta=None, recursive=False, **kw):
    os.environ["PSMF_STEP_PERSISTENT"] = "1" if persistent else "0"
    f = c.DeviceFilter(d, r, storage=storage, robust=robust, engine="step", dyn_kind=dyn, recursive=recursive, **kw)
    f.upload_series(Y)
    f.set_state(C0, 0.1 * np.eye(r), np.eye(r), Q, np.zeros(r), rho=1.0, lambda0=1.8, theta=theta)
    outs = []
    kern = f.geometry()["filter_kernel"]
    for a, b in cuts:
        f.run(a, b)
        s = f.get_state()
        s["yp"] = f.y_pred(a, b - a)
        outs.append(s)
    f.close()
    return kern, outs


def check(d, r, T, robust, storage, general_Q=False, seed=1, dyn=c.DYN_RANDOM_WALK, recursive=False, **kw):
    Y, C0 = make(d, r, T, seed, robust)
    if storage == "f32":
        Y = Y.astype(np.float32).astype(np.float64)
        C0 = C0.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(seed + 7)
    A = rng.standard_normal((r, r)) / np.sqrt(r)
    Q = 0.1 * np.eye(r) + (0.05 * (A @ A.T) if general_Q else 0.0)
    cut = max(1, T // 3)
    cuts = ((0, cut), (cut, T))
    theta = 0.01 * (1 + np.arange(r)) / r if dyn == c.DYN_COS_PHASE else None
    kp, po = run_dev(True, d, r, Y, C0, Q, robust, storage, cuts, dyn=dyn, theta=theta, recursive=recursive, **kw)
    kt, to = run_dev(False, d, r, Y, C0, Q, robust, storage, cuts, dyn=dyn, theta=theta, recursive=recursive, **kw)
    res = dict(d=d, r=r, T=T, robust=robust, storage=storage, general_Q=general_Q, dyn=dyn, recursive=recursive, kern=(kp, kt), **kw)
    if dyn == c.DYN_RANDOM_WALK and not kw:
        st = O.State(C=C0, V=0.1 * np.eye(r), mu=np.zeros(r), P=np.eye(r), Q=Q, rho=1.0, lam=1.8)
        for i, (a, b) in enumerate(cuts):
            st, Yp, _ = O.run_epoch(st, Y[a:b], O.Mode(robust=robust), O.RandomWalkDyn(), k0=a, want_grad=False)
            res[f"vs_oracle_{i}"] = {n: relerr(po[i][n], getattr(st, n)) for n in ("C", "V", "mu", "P")}
            res[f"vs_oracle_{i}"]["yp"] = relerr(po[i]["yp"], Yp)
            res[f"two_launch_vs_oracle_{i}"] = max(relerr(to[i][n], getattr(st, n)) for n in ("C", "V", "mu", "P"))
    for i in range(len(cuts)):
        res[f"vs_two_launch_{i}"] = {n: relerr(po[i][n], to[i][n]) for n in ("C", "V", "mu", "P", "yp", "Q")}
        for n in ("rho", "lam", "s", "eta", "N", "phi", "omega", "k"):
            res[f"vs_two_launch_{i}"][n] = abs(po[i][n] - to[i][n]) / max(abs(to[i][n]), 1e-300)
        if dyn == c.DYN_COS_PHASE:
            res[f"vs_two_launch_{i}"]["theta"] = relerr(po[i]["theta"], to[i]["theta"])
            res[f"vs_two_launch_{i}"]["gradsum"] = relerr(po[i]["gradsum"], to[i]["gradsum"])
    print(json.dumps(res), flush=True)


def timing(d, r, T, storage, robust=False):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((T, d), dtype=np.float32)
    C0 = 0.1 * rng.standard_normal((d, r))
    out = dict(d=d, r=r, T=T, storage=storage, robust=robust)
    for persistent in (True, False):
        os.environ["PSMF_STEP_PERSISTENT"] = "1" if persistent else "0"
        f = c.DeviceFilter(d, r, storage=storage, robust=robust, engine="step")
        f.upload_series(Y)
        f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
        f.run(0, min(T, 300))
        f.sync()
        ms = min(f.run_timed(0, T) for _ in range(3))
        out["persistent_us" if persistent else "two_launch_us"] = 1e3 * ms / T
        out["kern_p" if persistent else "kern_t"] = f.geometry()["filter_kernel"]
        f.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    check(300, 5, 20, False, "f64")
    check(4096, 32, 40, False, "f64")
    if not quick:
        check(4096, 32, 40, True, "f64")
        check(2000, 20, 30, True, "f32")
        check(777, 16, 25, False, "f64", general_Q=True)
        check(50, 3, 25, True, "f64")
        check(20000, 9, 30, False, "f32")
        check(100000, 32, 60, False, "f32")
        check(3000, 17, 30, True, "f64", general_Q=True)
        check(1500, 12, 40, False, "f64", dyn=c.DYN_COS_PHASE)
        check(1500, 24, 40, True, "f64", dyn=c.DYN_COS_PHASE, recursive=True, update_every=5)
        check(900, 8, 30, False, "f64", coef_update=False, eta_full=False, pbar_predict=False)
        # 33 <= r <= 48: the hub with LDS-resident matrices
        check(3000, 40, 30, False, "f64")
        check(2500, 48, 30, True, "f64")
        check(2000, 33, 30, True, "f32")
        check(1500, 37, 30, False, "f64", general_Q=True)
        check(1500, 44, 30, True, "f64", general_Q=True)
        check(1200, 36, 40, False, "f64", dyn=c.DYN_COS_PHASE)
        check(1200, 41, 40, True, "f64", dyn=c.DYN_COS_PHASE, recursive=True, update_every=5)
        check(900, 45, 30, False, "f64", coef_update=False, eta_full=False, pbar_predict=False)
        check(40000, 40, 40, False, "f32")
    timing(100000, 32, 2000, "f32")
    if not quick:
        timing(100000, 32, 2000, "f64")
        timing(100000, 32, 2000, "f32", robust=True)
        timing(20000, 32, 2000, "f64")
        timing(10000, 20, 2000, "f32")
        timing(20000, 10, 2000, "f64")
        timing(20000, 40, 2000, "f64")
        timing(20000, 48, 2000, "f64")
        timing(50000, 40, 2000, "f32")
        timing(5000, 36, 2000, "f64", robust=True)
