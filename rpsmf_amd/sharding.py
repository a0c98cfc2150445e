"""Row sharding of the d x r dictionary over GPUs (one process per GPU).

The filter shards naturally along the rows of C and y_k: each rank sweeps its own rows, the
ranks exchange the r+1 partial sums (h = C^T e, ee = e^T e) once per timestep with one
RCCL all-reduce, and every rank repeats the identical r x r float64 arithmetic, so the
replicated state stays bit-identical across ranks.
"""

import numpy as np

__all__ = ["shard_rows", "shard_bounds", "allreduce_sum_host"]


def shard_bounds(d, world):
    """Contiguous, near-equal row blocks: bounds[i]..bounds[i+1] for rank i."""
    base, extra = divmod(int(d), int(world))
    sizes = [base + (1 if i < extra else 0) for i in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def shard_rows(d, world, rank):
    """(row0, d_local) of `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if world > d:
        raise ValueError("more ranks than rows")
    b = shard_bounds(d, world)
    return int(b[rank]), int(b[rank + 1] - b[rank])


def allreduce_sum_host(vec, dist=None):
    """Sum a small float64 vector over ranks with torch.distributed (gloo) if initialised;
    used by the CPU (numpy backend) sharded path and by tests.  Rank order is fixed by the
    backend, every rank receives the same bits."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec
    import torch

    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def sharded_epoch_host(C_local, Y_local, V, P, Q, mu, rho, d_global, robust=False, lambda0=0.0, dist=None,
                       allreduce=None):
    """Host model (numpy) of what ONE rank of the sharded device filter computes over a series:
    the row sweep on its own rows, the per-step all-reduce of the r+1 partial sums (h = C^T e,
    ee = e^T e), and the replicated float64 r x r recursion with the algebraically tracked Gram
    matrix (rpsmf_amd/csrc/psmf_kernels.hip).  Full filter, random-walk dynamics, uniform R.

    Used to test the N > 1 logic without GPUs (gloo, world_size 2) and as executable
    documentation of the exchange pattern.  `allreduce(vec) -> vec` defaults to
    `allreduce_sum_host(vec, dist)`.  Returns the final (C_local, V, P, mu, rho, lam, Y_pred_local).
    """
    if allreduce is None:
        allreduce = lambda v: allreduce_sum_host(v, dist)
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
