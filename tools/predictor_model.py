"""CPU model (numpy, float64) of filter3's start predictor along config E's transient (d = 2e4, r = 32, 3 000 timesteps): the residual
||I - M_k Z_0||_F of the plain start, of the shipped start (rank-2 downdate, kappa-rescaled) and of its parts, of the best scalar rescaling,
and of the two-point start (linear inter- / extrapolation between the two inversions' previous inverses).  What it shows: the shipped start
is limited by |1 - kappa_k / kappa_{k-1}| ||Lbar Z|| (Lbar's share of M is rescaled with everything else); the two-point start removes that
term (X: / 6.5, W: / 3) for no matrix product.  Read by docs/MEASUREMENTS.md (round 5).   python tools/predictor_model.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
d, r, T = 20000, 32, 3000
ser = bench.Series(d, r, T, 35853, 0, d, False)
st0 = bench.init_state(d, r, 35853)
Y = np.concatenate([Yc for a, Yc in ser.chunks()], axis=0).astype(np.float64)
C = st0["C"].astype(np.float64).copy(); V = st0["V"].copy(); P = st0["P"].copy(); mu = st0["mu"].copy()
q = float(st0["Q"][0, 0]); rho = float(st0["rho"])
I = np.eye(r)
G = C.T @ C
Zprev = None; kap_prev = None; hprev = wprev = None; Nprev = eeprev = None
rows = []
extra = []
for k in range(T):
    y = Y[k]
    mub = mu
    Pbar = P + q * I
    w = V @ mub; s = float(mub @ w)
    eta = rho + float(np.sum(G * Pbar)) / d
    N = s + eta
    kap = 1.0 / (rho + s)
    Lbar = np.linalg.inv(Pbar)
    M = Lbar + kap * G
    Z = np.linalg.inv(M)
    if Zprev is not None:
        # plain start, kappa-rescaled start, rank-2 downdated + rescaled start
        res_plain = np.linalg.norm(I - M @ Zprev)
        sc = kap_prev / kap
        U = np.stack([hprev, wprev], axis=1)               # G_k - G_{k-1} = U K U^T
        K = np.array([[0.0, 1.0 / Nprev], [1.0 / Nprev, eeprev / Nprev**2]])
        a = Zprev @ U
        Tm = np.linalg.inv(np.linalg.inv(kap_prev * K) + U.T @ a)
        Z0 = sc * (Zprev - a @ Tm @ a.T)
        res_pred = np.linalg.norm(I - M @ Z0)
        Zp = Zprev - a @ Tm @ a.T
        c = kap / kap_prev
        tau_exact = np.trace(Lbar @ Zp) / r
        tau_apx = np.trace(Zprev) / (q * r)            # Lbar ~ I / q, Z' ~ Z_{k-1}
        g_exact = 1.0 / (c + (1 - c) * tau_exact)
        g_apx = 1.0 / (c + (1 - c) * tau_apx)
        res_g_exact = np.linalg.norm(I - M @ (g_exact * Zp))
        res_g_apx = np.linalg.norm(I - M @ (g_apx * Zp))
        # the W inversion: N = M / beta + I / q  (beta = 1 here)
        Nk = M + I / q; Wprev_ = np.linalg.inv(Mprev + I / q)
        aW = Wprev_ @ U
        TW = np.linalg.inv(np.linalg.inv(kap_prev * K) + U.T @ aW)
        Wp = Wprev_ - aW @ TW @ aW.T
        resW_old = np.linalg.norm(I - Nk @ (Wp / c))
        tauW = (2.0 / q) * np.trace(Wprev_) / r
        resW_new = np.linalg.norm(I - Nk @ (Wp / (c + (1 - c) * tauW)))
        beta = 1.0
        t = (1 - c) / (c * beta)
        Z0n = (1 - t) * Zp / c + t * Wp / (c * beta)
        epsY = (1 - c) * (1 + 1 / beta) / c
        W0n = (Wp + epsY * (Wp - beta * Zp)) / c
        resX_new2 = np.linalg.norm(I - M @ Z0n)
        resW_new2 = np.linalg.norm(I - Nk @ W0n)
        extra.append((k, res_g_exact, res_g_apx, resW_old, resW_new, resX_new2, resW_new2))
        # what remains: Lbar changes + the scale applied to Lbar
        Mhat = (kap / kap_prev) * (Mprev + kap_prev * (U @ K @ U.T))
        dM = M - Mhat
        rows.append((k, res_plain, res_pred, np.linalg.norm(dM @ Z0), np.linalg.norm((Lbar - Lbar_prev) @ Z0), abs(1 - kap / kap_prev) * np.linalg.norm(Lbar @ Z0)))
    # filter update
    e = y - C @ mub
    h = C.T @ e; ee = float(e @ e)
    Pp = 0.5 * (Z + Z.T)
    mu = mub + kap * (Pp @ h)
    C = C + np.outer(e, w) / N
    V = V - np.outer(w, w) / N
    G = G + (np.outer(h, w) + np.outer(w, h)) / N + ee * np.outer(w, w) / N**2
    P = Pp
    Zprev, kap_prev, hprev, wprev, Nprev, eeprev, Mprev, Lbar_prev = Z, kap, h, w, N, ee, M, Lbar
R = np.array(rows)
for lo, hi in ((1, 50), (50, 200), (200, 500), (500, 1000), (1000, 2000), (2000, 3000)):
    m = (R[:, 0] >= lo) & (R[:, 0] < hi)
    print(f"steps {lo:5d}-{hi:5d}: plain {np.median(R[m,1]):.2e}  predicted {np.median(R[m,2]):.2e} (90%: {np.quantile(R[m,2],0.9):.2e})  |dM Z0| {np.median(R[m,3]):.2e}  dLbar part {np.median(R[m,4]):.2e}  scale-on-Lbar part {np.median(R[m,5]):.2e}")

E = np.array(extra)
for lo, hi in ((1, 50), (50, 200), (200, 500), (500, 1000), (1000, 2000), (2000, 3000)):
    m = (E[:, 0] >= lo) & (E[:, 0] < hi)
    print(f"steps {lo:5d}-{hi:5d}: X optimal scalar exact {np.median(E[m,1]):.2e} / approx {np.median(E[m,2]):.2e} (90%: {np.quantile(E[m,2],0.9):.2e}) | W old {np.median(E[m,3]):.2e} new {np.median(E[m,4]):.2e} (90%: {np.quantile(E[m,4],0.9):.2e})")

print("two-point (Z', W') interpolation / extrapolation of the resolvent:")
for lo, hi in ((1, 50), (50, 200), (200, 500), (500, 1000), (1000, 2000), (2000, 3000)):
    m = (E[:, 0] >= lo) & (E[:, 0] < hi)
    Rm = (R[:, 0] >= lo) & (R[:, 0] < hi)
    print(f"steps {lo:5d}-{hi:5d}: X old {np.median(R[Rm,2]):.2e} -> {np.median(E[m,5]):.2e} (90%: {np.quantile(E[m,5],0.9):.2e}) | W old {np.median(E[m,3]):.2e} -> {np.median(E[m,6]):.2e} (90%: {np.quantile(E[m,6],0.9):.2e})")
