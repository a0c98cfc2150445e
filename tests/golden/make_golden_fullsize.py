#!/usr/bin/env python
"""Full-horizon known answers of BASELINE config E (d = 100 000, r = 32, T = 10 000) from the CPU oracle.

What the headline number is quoted on -- 10 000 timesteps per pass, state carried from pass to pass -- is too long
for the oracle to be re-run inside a GPU test (about 0.1 s of host time per timestep at this size), so this script runs it
ONCE in the build container and stores, at a few checkpoints of two consecutive epochs, everything r-sized of the filter
state plus fixed linear sketches of the d-sized quantities:

    V, P, mu, Q[0,0], rho, lambda               (exact, float64)
    S^T C   with a fixed seeded S (d x 64)      (every row of C enters every entry)
    C[rows], rows = 128 fixed row indices       (element-wise)
    y_hat_k[rows] at the checkpoint             (element-wise)
    y_hat_k[track] for EVERY k of both epochs, track = 4 fixed series (nothing of the horizon is unobserved)

The workload is bench.py's: `bench.Series` (ExperimentSynthetic/data.py:6-60 semantics; seeds of the reference Makefile:55,64),
`bench.init_state`, random-walk dynamics, full filter.  Epoch 2 starts from the state epoch 1 ended with, as
PSMFIter.step_reset / rPSMFIter.step_reset do (pypsmf/psmf/psmf.py:75-83; rpsmf.py:106-114: rPSMF puts Q, R, lambda back
to their initial values, C, V, mu, P are carried).

    python tests/golden/make_golden_fullsize.py psmf      # ~16 min of CPU (3 BLAS threads)
    python tests/golden/make_golden_fullsize.py rpsmf

Output: tests/golden/fullsize_E_{psmf,rpsmf}.npz.  Test infrastructure; the oracle is the checker, never the product.
"""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CHECKPOINTS = (300, 1000, 3000, 10_000, 10_300, 11_000, 13_000, 20_000)     # global timestep index over the two epochs
N_SKETCH, N_ROWS, N_TRACK = 64, 128, 4


def sketch_matrix(d):
    return np.random.default_rng(20240607).standard_normal((d, N_SKETCH))


def row_subset(d):
    return np.linspace(0, d - 1, 2 * N_ROWS).astype(np.int64)[::2]


def track_subset(d):
    return np.array([0, d // 3, (2 * d) // 3, d - 1], dtype=np.int64)


def generate(robust, d=100_000, r=32, T=10_000, epochs=2, checkpoints=CHECKPOINTS, log=print):
    import bench
    from oracle import psmf_oracle as O

    seed = 35833 if robust else 35853
    series = bench.Series(d, r, T, seed, 0, d, robust)
    st0 = bench.init_state(d, r, seed)
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=st0["P"].copy(), Q=st0["Q"].copy(),
                 rho=st0["rho"], lam=st0["lam"])
    mode, dyn = O.Mode(robust=robust), O.RandomWalkDyn()
    S, rows, track = sketch_matrix(d), row_subset(d), track_subset(d)
    out = {"d": d, "r": r, "T": T, "epochs": epochs, "robust": int(robust), "seed": seed, "rows": rows, "track": track,
           "checkpoints": np.array([k for k in checkpoints if k <= epochs * T], dtype=np.int64)}
    ytrack = np.empty((epochs * T, N_TRACK))
    t_start = time.perf_counter()
    kg = 0
    for ep in range(epochs):
        if ep > 0 and robust:                   # rpsmf.py:106-114
            st.Q, st.rho, st.lam = st0["Q"].copy(), st0["rho"], st0["lam"]
        for a, Yc in series.chunks(chunk=500):
            Y64 = Yc.astype(np.float64)
            for j in range(Y64.shape[0]):
                k = a + j + 1
                st, info = O.lowrank_step(st, Y64[j], k, mode, dyn, want_grad=False)
                ytrack[kg] = info.y_pred[track]
                kg += 1
                if kg in checkpoints:
                    p = f"k{kg}_"
                    out[p + "V"], out[p + "P"], out[p + "mu"] = st.V.copy(), st.P.copy(), st.mu.copy()
                    out[p + "q"], out[p + "rho"], out[p + "lam"] = float(st.Q[0, 0]), float(st.rho), float(st.lam)
                    out[p + "StC"] = S.T @ st.C
                    out[p + "Crows"] = st.C[rows].copy()
                    out[p + "yhat_rows"] = info.y_pred[rows].copy()
                    out[p + "eta"], out[p + "N"] = info.eta, info.N
                    log(f"[{'rPSMF' if robust else 'PSMF'}] checkpoint {kg}: {time.perf_counter() - t_start:.0f} s", flush=True)
            if a % 1000 == 0:
                log(f"[{'rPSMF' if robust else 'PSMF'}] epoch {ep} k={a + Y64.shape[0]}: {time.perf_counter() - t_start:.0f} s", flush=True)
    out["yhat_track"] = ytrack.astype(np.float32)          # compared at 1e-5: float32 (6e-8) is plenty
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "psmf"
    robust = which == "rpsmf"
    res = generate(robust)
    path = os.path.join(HERE, f"fullsize_E_{which}.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes")
