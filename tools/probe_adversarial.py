"""Adversarial series for the blocked engine's carried Newton-Schulz starts: device vs CPU oracle and the
inversion counters per case.  usage (GPU box): python tools/probe_adversarial.py [d] [T]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from adversarial_cases import CASES, make_case  # noqa: E402
from oracle import psmf_oracle as O  # noqa: E402
from rpsmf_amd import _capi  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
    for name in CASES:
        for r, robust in ((32, False), (20, True)):
            cs = make_case(name, d, r, T, robust)
            st = O.State(C=cs["C0"].copy(), V=cs["V0"], mu=np.zeros(r), P=cs["P0"], Q=cs["Q"], rho=1.0, lam=1.8)
            t0 = time.time()
            st, Yp, trace = O.run_epoch(st, cs["Y"].astype(np.float64), O.Mode(robust=robust), O.RandomWalkDyn(), keep=cs["checkpoints"], want_grad=False)
            t_or = time.time() - t0
            f = _capi.DeviceFilter(d, r, robust=robust, storage="f32")
            f.upload_series(cs["Y"])
            f.set_state(cs["C0"], cs["V0"], cs["P0"], cs["Q"], np.zeros(r), rho=1.0, lambda0=1.8)
            kp = 0
            line = []
            for k in cs["checkpoints"]:
                f.counters(reset=True)
                f.run(kp, k)
                s = f.get_state()
                c = f.counters()
                ref = trace[k][0]
                errs = {n: rel(s[n], getattr(ref, n)) for n in ("C", "V", "mu", "P")}
                line.append(f"k={k}: " + " ".join(f"{n}={e:.1e}" for n, e in errs.items()) +
                            f" | ns={c['ns_steps']} sweep={c['sweep_steps']} it={c['ns_iterations']} failed={c['ns_failed']}")
                kp = k
            ey = rel(f.y_pred(0, T), Yp)
            f.close()
            print(f"[{name} r={r} {'rPSMF' if robust else 'PSMF'}] oracle {t_or:.0f}s  y_pred={ey:.1e}")
            for l in line:
                print("    ", l)
            sys.stdout.flush()


if __name__ == "__main__":
    main()
