"""Exact time-blocking of the PSMF / rPSMF recursion (host model of the blocked device engine).

Within a block of B consecutive steps every innovation e_j lies in the span of
Z = [C_{k0} | y_{k0+1} ... y_{k0+B}]  (d x (r+B)), so with K = Z^T Z ((r+B) x (r+B), the only
d-sized contraction, ONE per block) the B steps run in coefficient space:

    C_j = Z A_j,  A_0 = [I_r; 0]          a_j = u_{r+j} - A_{j-1} mu_bar_j      (e_j = Z a_j)
    y_hat_j = Z b_j,  b_j = A_{j-1} mu_bar_j
    h_j = C_{j-1}^T e_j = A_{j-1}^T K a_j     ee_j = a_j^T K a_j     G_{j-1} = A_{j-1}^T K A_{j-1}
    A_j = A_{j-1} + a_j w_j^T / N_j           (the r x r recursion is unchanged)

and C_{k0+B} = Z A_B, Y_hat_block = Z [b_1 .. b_B] are two more d-sized products per block.
Valid for unmasked data with uniform diagonal R (PSMFIter / rPSMFIter as shipped) and any f.
SURVEY section 7 ("exact time-blocking"); same result as the step-by-step recursion up to
float64 round-off (tests/test_blocked_host.py).
"""

import numpy as np

__all__ = ["blocked_epoch_host"]


def blocked_epoch_host(C0, Y, V, P, Q, mu, rho, B=32, robust=False, lambda0=0.0, alpha=1.0, beta=1.0,
                       storage=np.float64, gram_allreduce=None):
    """Full filter, random-walk dynamics.  Y: (T, d).  `storage`: dtype C is rounded to at block ends
    (the device stores C in f32 by default).  `gram_allreduce`: optional callable applied to each
    block's K (row-sharded multi-GPU: the one collective per block).  Returns
    (C, V, P, mu, rho, lam, Y_pred)."""
    C = np.array(C0, dtype=storage).astype(np.float64)
    V, P, Q, mu = (np.array(a, dtype=np.float64) for a in (V, P, Q, mu))
    T, d_local = Y.shape
    d = float(gram_allreduce.d_global) if gram_allreduce is not None and hasattr(gram_allreduce, "d_global") else float(d_local)
    r = C.shape[1]
    lam = float(lambda0)
    Yp = np.empty((T, d_local))
    for k0 in range(0, T, B):
        nb = min(B, T - k0)
        Z = np.hstack([C, Y[k0:k0 + nb].T.astype(np.float64)])           # d x (r + nb)
        K = Z.T @ Z
        if gram_allreduce is not None:
            K = gram_allreduce(K)
        A = np.zeros((r + nb, r))
        A[:r] = np.eye(r)
        KA = K @ A
        G = A.T @ KA
        Bc = np.zeros((r + nb, nb))                                        # columns b_j
        for j in range(nb):
            mu_bar = mu
            P_bar = P + Q
            b = A @ mu_bar
            a = -b
            a[r + j] += 1.0
            Ka = K[:, r + j] - KA @ mu_bar
            h = A.T @ Ka
            ee = float(a @ Ka)
            w = V @ mu_bar
            s = float(mu_bar @ w)
            eta = rho + float(np.sum(G * P_bar)) / d
            N = s + eta
            kappa = 1.0 / (rho + s)
            P_plus = np.linalg.inv(np.linalg.inv(P_bar) + kappa * G)
            P_plus = 0.5 * (P_plus + P_plus.T)
            bb = kappa * h
            mu = mu_bar + P_plus @ bb
            V = V - np.outer(w, w) / N
            if robust:
                phi = (lam + ee / N) / (lam + d)
                omega = (lam + kappa * ee - float(bb @ P_plus @ bb)) / (lam + d)
                V = alpha * phi * V
                P_plus = beta * omega * P_plus
                Q = omega * Q
                rho = omega * rho
                lam = lam + d
            P = P_plus
            G = G + (np.outer(h, w) + np.outer(w, h)) / N + ee * np.outer(w, w) / N**2
            A = A + np.outer(a, w) / N
            KA = KA + np.outer(Ka, w) / N
            Bc[:, j] = b
        C = (Z @ A).astype(storage).astype(np.float64)
        Yp[k0:k0 + nb] = (Z @ Bc).T
    return C, V, P, mu, rho, lam, Yp
