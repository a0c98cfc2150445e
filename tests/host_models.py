"""Host (numpy) models of what ONE rank of the sharded device engines computes -- TEST INFRASTRUCTURE ONLY.

They restate the device algorithms (not the reference's: that is oracle/) so that the N > 1 logic -- which partial sums
are exchanged, when, and that the replicated float64 r x r state stays bit-identical -- can be exercised without GPUs
(gloo, world_size 2) and compared with the unsharded CPU oracle:

  sharded_epoch_host            per-step engine: row sweep on the local rows, all-reduce of the r + 1 partial sums
                                (h = C^T e, ee = e^T e) per timestep, tracked Gram (rpsmf_amd/csrc/psmf_kernels.hip)
  blocked_epoch_host            blocked engine, one block after the other: all-reduce of K = Z^T Z per block
                                (psmf_capi.hip enqueue_block)
  blocked_pipelined_epoch_host  blocked engine as bench.py runs it: all-reduce of the first block's K, then of one
                                cross-Gram XG = [Z_b | Y_{b+1}]^T Y_{b+1} per block, from which the next block's K is
                                assembled with the tracked Gram (psmf_capi.hip enqueue_blocks_pipelined,
                                psmf_block.hip assemble_K)

Exact time-blocking: within a block of B consecutive steps every innovation e_j lies in the span of
Z = [C_{k0} | y_{k0+1} ... y_{k0+B}]  (d x (r+B)), so with K = Z^T Z the B steps run in coefficient space:
    C_j = Z A_j,  A_0 = [I_r; 0]          a_j = u_{r+j} - A_{j-1} mu_bar_j      (e_j = Z a_j)
    y_hat_j = Z b_j,  b_j = A_{j-1} mu_bar_j
    h_j = A_{j-1}^T K a_j     ee_j = a_j^T K a_j     G_{j-1} = A_{j-1}^T K A_{j-1}
    A_j = A_{j-1} + a_j w_j^T / N_j           (the r x r recursion is unchanged)
and C_{k0+B} = Z A_B, Y_hat_block = Z [b_1 .. b_B] are two more d-sized products per block.
"""

import numpy as np

__all__ = ["sharded_epoch_host", "blocked_epoch_host", "blocked_pipelined_epoch_host", "gloo_allreduce"]


def gloo_allreduce(dist):
    """vec -> sum over ranks through torch.distributed (gloo): fixed rank order, same bits on every rank."""
    def f(vec):
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return vec
        import torch

        shape = np.shape(vec)
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64).reshape(-1))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy().reshape(shape)
    return f


def sharded_epoch_host(C_local, Y_local, V, P, Q, mu, rho, d_global, robust=False, lambda0=0.0, dist=None,
                       allreduce=None):
    """Host model (numpy) of what ONE rank of the sharded device filter computes over a series:
    the row sweep on its own rows, the per-step all-reduce of the r+1 partial sums (h = C^T e,
    ee = e^T e), and the replicated float64 r x r recursion with the algebraically tracked Gram
    matrix (rpsmf_amd/csrc/psmf_kernels.hip).  Full filter, random-walk dynamics, uniform R.

    Used to test the N > 1 logic without GPUs (gloo, world_size 2) and as executable
    documentation of the exchange pattern.  `allreduce(vec) -> vec` defaults to
    `gloo_allreduce(dist)`.  Returns the final (C_local, V, P, mu, rho, lam, Y_pred_local).
    """
    if allreduce is None:
        allreduce = gloo_allreduce(dist)
    C = np.array(C_local, dtype=np.float64)
    V, P, Q, mu = (np.array(a, dtype=np.float64) for a in (V, P, Q, mu))
    r = C.shape[1]
    T = Y_local.shape[0]
    dd = float(d_global)
    lam = float(lambda0)
    G = allreduce((C.T @ C).reshape(-1)).reshape(r, r)        # exact Gram once, then tracked
    Yp = np.empty_like(Y_local, dtype=np.float64)
    I = np.eye(r)
    for t in range(T):
        mu_bar = mu
        P_bar = P + Q
        w = V @ mu_bar
        s = float(mu_bar @ w)
        eta = rho + float(np.sum(G * P_bar)) / dd
        N = s + eta
        kappa = 1.0 / (rho + s)
        # --- row sweep on the local rows (device: psmf_sweep_solve)
        yhat = C @ mu_bar
        e = Y_local[t] - yhat
        Yp[t] = yhat
        part = np.concatenate([C.T @ e, [e @ e]])
        C = C + np.outer(e, w) / N
        # --- the only exchange of the step
        red = allreduce(part)
        h, ee = red[:r], float(red[r])
        # --- replicated r x r stage (device: solve block + psmf_serial)
        P_plus = np.linalg.inv(np.linalg.inv(P_bar) + kappa * G)
        P_plus = 0.5 * (P_plus + P_plus.T)
        b = kappa * h
        mu = mu_bar + P_plus @ b
        V = V - np.outer(w, w) / N
        if robust:
            phi = (lam + ee / N) / (lam + dd)
            omega = (lam + kappa * ee - float(b @ P_plus @ b)) / (lam + dd)
            V = phi * V
            P_plus = omega * P_plus
            Q = omega * Q
            rho = omega * rho
            lam = lam + dd
        P = P_plus
        G = G + (np.outer(h, w) + np.outer(w, h)) / N + ee * np.outer(w, w) / N**2
    return C, V, P, mu, rho, lam, Yp


def blocked_epoch_host(C0, Y, V, P, Q, mu, rho, B=32, robust=False, lambda0=0.0, alpha=1.0, beta=1.0,
                       storage=np.float64, gram_allreduce=None, d_global=None):
    """Full filter, random-walk dynamics.  Y: (T, d).  `storage`: dtype C is rounded to at block ends
    (the device stores C in f32 by default).  `gram_allreduce`: optional callable applied to each
    block's K (row-sharded multi-GPU: the one collective per block).  Returns
    (C, V, P, mu, rho, lam, Y_pred)."""
    C = np.array(C0, dtype=storage).astype(np.float64)
    V, P, Q, mu = (np.array(a, dtype=np.float64) for a in (V, P, Q, mu))
    T, d_local = Y.shape
    d = float(d_local if d_global is None else d_global)
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


def _block_steps(K, nb, r, G, V, P, Q, mu, rho, lam, d, robust, alpha, beta):
    """The nb steps of one block in coefficient space (psmf_block.hip psmf_blk_filter): returns the updated r x r state,
    the tracked Gram, A_nb and the columns b_j."""
    A = np.zeros((r + nb, r))
    A[:r] = np.eye(r)
    KA = K @ A
    Bc = np.zeros((r + nb, nb))
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
    return G, V, P, Q, mu, rho, lam, A, Bc


def blocked_pipelined_epoch_host(C0, Y, V, P, Q, mu, rho, B=32, robust=False, lambda0=0.0, alpha=1.0, beta=1.0,
                                 storage=np.float64, allreduce=None, d_global=None):
    """The pipelined blocked engine on one rank's rows.  Exchanges: K of the first block (once), then per block the
    cross-Gram XG = [[Z_b^T Y_{b+1}], [Y_{b+1}^T Y_{b+1}]] computed from the C the block STARTED with; the next block's
    K is assembled as [[G_B, A_B^T X], [., Y'^T Y']] with the algebraically tracked G_B (replicated arithmetic).
    Returns (C, V, P, mu, rho, lam, Y_pred)."""
    if allreduce is None:
        allreduce = lambda x: x
    C = np.array(C0, dtype=storage).astype(np.float64)
    V, P, Q, mu = (np.array(a, dtype=np.float64) for a in (V, P, Q, mu))
    T, d_local = Y.shape
    d = float(d_local if d_global is None else d_global)
    r = C.shape[1]
    lam = float(lambda0)
    Yp = np.empty((T, d_local))
    starts = list(range(0, T, B))
    K = None
    for bi, k0 in enumerate(starts):
        nb = min(B, T - k0)
        Z = np.hstack([C, Y[k0:k0 + nb].T.astype(np.float64)])
        if bi == 0:
            K = allreduce(Z.T @ Z)
            G = K[:r, :r].copy()
        # cross-Gram for the next block, from this block's Z (C as of the block start)
        XG = None
        if bi + 1 < len(starts):
            k1 = starts[bi + 1]
            nb1 = min(B, T - k1)
            Y1 = Y[k1:k1 + nb1].T.astype(np.float64)
            XG = allreduce(np.vstack([Z.T @ Y1, Y1.T @ Y1]))
        G, V, P, Q, mu, rho, lam, A, Bc = _block_steps(K, nb, r, G, V, P, Q, mu, rho, lam, d, robust, alpha, beta)
        C = (Z @ A).astype(storage).astype(np.float64)
        Yp[k0:k0 + nb] = (Z @ Bc).T
        if XG is not None:
            X, YY = XG[:r + nb], XG[r + nb:]
            K = np.zeros((r + nb1, r + nb1))
            K[:r, :r] = G                       # tracked, NOT recomputed from the rounded C
            K[:r, r:] = A.T @ X
            K[r:, :r] = K[:r, r:].T
            K[r:, r:] = YY
    return C, V, P, mu, rho, lam, Yp
