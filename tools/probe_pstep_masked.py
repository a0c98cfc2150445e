"""GPU-box probe: the masked persistent per-step kernel against the two-launch masked engine (PSMF_STEP_PERSISTENT=0) and the oracle's
masked step, then us per timestep of both.   python tools/probe_pstep_masked.py [quick]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import psmf_oracle as O
from rpsmf_amd import _capi as c


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def run(persistent, d, r, Y, M, C0, robust, storage, cuts):
    os.environ["PSMF_STEP_PERSISTENT"] = "1" if persistent else "0"
    f = c.DeviceFilter(d, r, storage=storage, robust=robust, engine="step", masked=True)
    f.upload_series(Y)
    f.upload_mask(M)
    f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
    kern = f.geometry()["filter_kernel"]
    outs = []
    for a, b in cuts:
        f.run(a, b)
        s = f.get_state()
        s["yp"] = f.y_pred(a, b - a)
        s["sc"] = f.step_scalars(a, b - a)
        outs.append(s)
    f.close()
    return kern, outs


def check(d, r, T, robust, storage, seed=3):
    rng = np.random.default_rng(seed)
    Ct = rng.standard_normal((d, r))
    x = rng.standard_normal(r)
    Y = np.empty((T, d))
    for t in range(T):
        x = x + 0.1 * rng.standard_normal(r)
        Y[t] = Ct @ x + 0.3 * rng.standard_normal(d)
    M = (rng.random((T, d)) > 0.4).astype(np.uint8)
    Y = Y * M
    C0 = 0.1 * rng.standard_normal((d, r))
    if storage == "f32":
        Y = Y.astype(np.float32).astype(np.float64)
        C0 = C0.astype(np.float32).astype(np.float64)
    cut = max(1, T // 3)
    cuts = ((0, cut), (cut, T))
    kp, po = run(True, d, r, Y, M, C0, robust, storage, cuts)
    kt, to = run(False, d, r, Y, M, C0, robust, storage, cuts)
    res = dict(d=d, r=r, T=T, robust=robust, storage=storage, kern=(kp, kt))
    st = O.State(C=C0, V=0.1 * np.eye(r), mu=np.zeros(r), P=np.eye(r), Q=0.1 * np.eye(r), rho=1.0, lam=1.8)
    for i, (a, b) in enumerate(cuts):
        yp = np.empty((b - a, d))
        for k in range(a, b):
            st, info = O.lowrank_step(st, Y[k], k + 1, O.Mode(robust=robust), O.RandomWalkDyn(), mask=M[k].astype(float), want_grad=False)
            yp[k - a] = info.y_pred
        res[f"vs_oracle_{i}"] = {n: relerr(po[i][n], getattr(st, n)) for n in ("C", "V", "mu", "P")}
        res[f"vs_oracle_{i}"]["yp"] = relerr(po[i]["yp"], yp)
        res[f"two_launch_vs_oracle_{i}"] = max(relerr(to[i][n], getattr(st, n)) for n in ("C", "V", "mu", "P"))
        res[f"vs_two_launch_{i}"] = {n: relerr(po[i][n], to[i][n]) for n in ("C", "V", "mu", "P", "yp", "sc", "Q")}
        for n in ("rho", "lam"):
            res[f"vs_two_launch_{i}"][n] = abs(po[i][n] - to[i][n]) / max(abs(to[i][n]), 1e-300)
    print(json.dumps(res), flush=True)


def timing(d, r, T, storage):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((T, d), dtype=np.float32)
    M = (rng.random((T, d)) > 0.4).astype(np.uint8)
    C0 = 0.1 * rng.standard_normal((d, r))
    out = dict(d=d, r=r, T=T, storage=storage)
    for persistent in (True, False):
        os.environ["PSMF_STEP_PERSISTENT"] = "1" if persistent else "0"
        f = c.DeviceFilter(d, r, storage=storage, engine="step", masked=True)
        f.upload_series(Y)
        f.upload_mask(M)
        f.set_state(C0, 0.1 * np.eye(r), np.eye(r), 0.1 * np.eye(r), np.zeros(r), rho=1.0, lambda0=1.8)
        f.run(0, min(T, 200))
        f.sync()
        ms = min(f.run_timed(0, T) for _ in range(3))
        out["persistent_us" if persistent else "launches_us"] = 1e3 * ms / T
        out["kern_p" if persistent else "kern_t"] = f.geometry()["filter_kernel"]
        f.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    check(300, 5, 20, False, "f64")
    check(4096, 32, 30, False, "f64")
    if not quick:
        check(4096, 32, 30, True, "f64")
        check(2000, 20, 30, True, "f32")
        check(40, 24, 25, False, "f64")
        check(20000, 10, 30, True, "f64")
        check(700, 3, 40, False, "f64")
        check(50000, 32, 24, False, "f64")
    timing(100000, 32, 1000, "f32")
    if not quick:
        timing(100000, 32, 1000, "f64")
        timing(20000, 10, 1000, "f64")
        timing(20000, 32, 1000, "f64")
