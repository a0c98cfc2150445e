"""CPU oracle for the masked PSMF / rPSMF filter of ExperimentImpute  --  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this.  Parity status: PINNED against the reference's stored known answers
(``ExperimentImpute/output/LondonAir_PM25_*_{PSMF,rPSMF}.json`` etc., replayed in
``tests/test_oracle_kat.py`` from fixtures made by ``tests/golden/make_golden.py``) and
against the reference functions run in the build container.

Follows (paths relative to /root/reference):
  ProbabilisticSequentialMatrixFactorizer   ExperimentImpute/PSMF.py:40-95
  robust_PSMF                               ExperimentImpute/rPSMF.py:40-148
  RMSEM / compute_number_inside_bars        ExperimentImpute/common.py:79-94
  stochasticGradientStateSpaceMF (MLE-SMF)  ExperimentImpute/MLESMF.py:40-92
  temporalRegularizedMF (TMF)               ExperimentImpute/TMF.py:30-73
(the two baseline filters of the imputation tables that share the masked Kalman contractions: SURVEY 8(f)-4;
pinned by running the reference functions in the build container, fixture tests/golden/impute_baselines.npz)

The reference forms Mk = diag(M[:, t]), a d x d Woodbury inverse and d x d traces every
column; here every such object is reduced to the r x r quantities of SURVEY App. A
(weights kappa_i = m_i / (rho_i + s)), which is algebraically identical.
"""

from __future__ import annotations

import numpy as np

__all__ = ["impute_filter", "mle_smf_filter", "tmf_filter", "rmse_on_mask", "coverage"]


def rmse_on_mask(Y1, Y2, Mmiss):
    """common.py:79-84: RMSE over the entries where Mmiss == 1."""
    n_miss = np.sum(Mmiss)
    diff = (Y1 - Y2) * Mmiss
    return float(np.sqrt(np.sum(diff * diff) / n_miss))


def coverage(Mmiss, Yorg, YrecL, YrecH):
    """common.py:87-94: fraction of held-out entries strictly inside (YrecL, YrecH)."""
    sel = Mmiss == 1
    inside = (Yorg < YrecH) & (YrecL < Yorg) & sel
    return float(np.sum(inside) / np.sum(Mmiss))


def impute_filter(Y, C, X, M, Mmiss, V, Q0, rho0, P, sig, Iter, YorgInt, Einit,
                  robust=False, lambda0=0.0, return_state=False):
    """Masked filter over the n columns of Y (d, n), ``Iter`` passes.

    Y       (d, n) data, 0 where missing.     M      (d, n) 0/1, 1 = observed.
    C       (d, r) initial dictionary.        X      (r, n) initial coefficients; mutated in
                                                      place exactly like the reference
                                                      (PSMF.py:74).
    V, P    (r, r).  Q0 (r, r).  rho0 scalar or (d,) = diag(R).
    Returns (Epred, Efull, InsideBars[, state dict]) with Epred / Efull of shape (1, Iter+1).
    """
    d, n = Y.shape
    r = C.shape[1]
    C = np.array(C, dtype=float)
    V = np.array(V, dtype=float)
    P = np.array(P, dtype=float)
    Q0 = np.array(Q0, dtype=float)
    rho0 = np.full(d, float(rho0)) if np.ndim(rho0) == 0 else np.asarray(rho0, dtype=float)
    Mf = np.asarray(M, dtype=float)

    Epred = np.zeros((1, Iter + 1))
    Efull = np.zeros((1, Iter + 1))
    Epred[0, 0] = Einit
    Efull[0, 0] = Einit
    Yrec = np.zeros((d, n))
    YrecL = np.zeros((d, n))
    YrecH = np.zeros((d, n))
    I_r = np.eye(r)

    Q, rho, lam = Q0, rho0, float(lambda0)
    for it in range(Iter):
        if robust:
            Q, rho, lam = Q0, rho0, float(lambda0)  # rPSMF.py:77-79 (V, P, C carry over)
        for t in range(n):
            m = Mf[:, t]
            xp = X[:, n - 1].copy() if t == 0 else X[:, t - 1].copy()  # PSMF.py:65
            PP = P + Q
            yhat = C @ xp
            Yrec[:, t] = yhat  # unmasked prediction (PSMF.py:68)
            e = m * (Y[:, t] - yhat)
            w = V @ xp
            s = float(xp @ w)
            kap = m / (rho + s)
            Cm = C * m[:, None]
            G = Cm.T @ C
            G_R = (C * kap[:, None]).T @ C
            b = C.T @ (kap * e)
            P_plus = np.linalg.solve(I_r + PP @ G_R, PP)
            P_plus = 0.5 * (P_plus + P_plus.T)
            X[:, t] = xp + P_plus @ b
            eta = (float(m @ rho) + float(np.sum(G * PP))) / d  # PSMF.py:77 (divide by d)
            N = s + eta
            C = C + np.outer(e, w) / N
            V_new = V - np.outer(w, w) / N
            if robust:
                q = float(kap @ (e * e))
                ee = float(e @ e)
                omega = (lam + q - float(b @ P_plus @ b)) / (lam + d)  # rPSMF.py:105
                P = omega * P_plus
                phi = (lam + ee / N) / (lam + d)  # rPSMF.py:112-114 (e is 0 on missing rows)
                V = phi * V_new
                band = sig * np.sqrt(s * m + eta)  # rPSMF.py:121-123
                Q = omega * Q
                rho = omega * rho
                lam = lam + d
            else:
                P = P_plus
                V = V_new
                band = sig * np.sqrt(N)  # PSMF.py:83-84
            YrecL[:, t] = yhat - band
            YrecH[:, t] = yhat + band
        Yrec2 = C @ X
        Epred[0, it + 1] = rmse_on_mask(Yrec, YorgInt, Mmiss)
        Efull[0, it + 1] = rmse_on_mask(Yrec2, YorgInt, Mmiss)
    inside = coverage(Mmiss, YorgInt, YrecL, YrecH)
    if return_state:
        return Epred, Efull, inside, dict(C=C, X=X, V=V, P=P, Yrec=Yrec, YrecL=YrecL, YrecH=YrecH)
    return Epred, Efull, inside


def mle_smf_filter(Y, C, X, M, Mmiss, Q, rho, P, sig, Iter, YorgInt, Einit, return_state=False):
    """MLE-SMF (MLESMF.py:40-92): the masked Kalman update of x and P with weights m_i / rho_i (its Woodbury inverse
    uses R itself, MLESMF.py:70, not R + s I), and a stochastic-gradient step on C,
    C += gam / eta_k * (m o (y - C x_p)) x_p^T with gam = 1e-6 / (pass + 1)^0.7 (MLESMF.py:59-60,79); bands
    y_hat -+ sig sqrt(eta_k) (MLESMF.py:81-82).  ``lam`` of the reference signature is unused there."""
    d, n = Y.shape
    r = C.shape[1]
    C = np.array(C, dtype=float)
    P = np.array(P, dtype=float)
    Q = np.array(Q, dtype=float)
    rho = np.full(d, float(rho)) if np.ndim(rho) == 0 else np.asarray(rho, dtype=float)
    Mf = np.asarray(M, dtype=float)
    Epred = np.zeros((1, Iter + 1)); Efull = np.zeros((1, Iter + 1))
    Epred[0, 0] = Einit; Efull[0, 0] = Einit
    Yrec = np.zeros((d, n)); YrecL = np.zeros((d, n)); YrecH = np.zeros((d, n))
    I_r = np.eye(r)
    for it in range(Iter):
        gam = 1e-6 / ((it + 1) ** 0.7)
        for t in range(n):
            m = Mf[:, t]
            xp = X[:, n - 1].copy() if t == 0 else X[:, t - 1].copy()
            PP = P + Q
            yhat = C @ xp
            Yrec[:, t] = yhat
            e = m * (Y[:, t] - yhat)
            kap = m / rho
            G = (C * m[:, None]).T @ C
            G_R = (C * kap[:, None]).T @ C
            b = C.T @ (kap * e)
            P_plus = np.linalg.solve(I_r + PP @ G_R, PP)
            P_plus = 0.5 * (P_plus + P_plus.T)
            X[:, t] = xp + P_plus @ b
            P = P_plus
            eta = (float(m @ rho) + float(np.sum(G * PP))) / d
            C = C + (gam / eta) * np.outer(e, xp)
            band = sig * np.sqrt(eta)
            YrecL[:, t] = yhat - band
            YrecH[:, t] = yhat + band
        Epred[0, it + 1] = rmse_on_mask(Yrec, YorgInt, Mmiss)
        Efull[0, it + 1] = rmse_on_mask(C @ X, YorgInt, Mmiss)
    inside = coverage(Mmiss, YorgInt, YrecL, YrecH)
    if return_state:
        return Epred, Efull, inside, dict(C=C, X=X, P=P, Yrec=Yrec, YrecL=YrecL, YrecH=YrecH)
    return Epred, Efull, inside


def tmf_filter(Y, C, X, M, Mmiss, Iter, YorgInt, Einit, nu=2.0, return_state=False):
    """TMF (TMF.py:30-73): x_t = (C_M^T C_M + nu I)^-1 (nu x_p + C_M^T y)  =  x_p + (nu I + G)^-1 C_M^T e, then the plain
    gradient step C += gam (m o (y - C x_p)) x_p^T, gam = 1e-6 / (pass + 1)^0.7, nu = 2 (TMF.py:46-48,60-63)."""
    d, n = Y.shape
    r = C.shape[1]
    C = np.array(C, dtype=float)
    Mf = np.asarray(M, dtype=float)
    Epred = np.zeros((1, Iter + 1)); Efull = np.zeros((1, Iter + 1))
    Epred[0, 0] = Einit; Efull[0, 0] = Einit
    Yrec = np.zeros((d, n))
    I_r = np.eye(r)
    for it in range(Iter):
        gam = 1e-6 / ((it + 1) ** 0.7)
        for t in range(n):
            m = Mf[:, t]
            xp = X[:, n - 1].copy() if t == 0 else X[:, t - 1].copy()
            yhat = C @ xp
            Yrec[:, t] = yhat
            e = m * (Y[:, t] - yhat)
            G = (C * m[:, None]).T @ C
            X[:, t] = xp + np.linalg.solve(nu * I_r + G, C.T @ e)
            C = C + gam * np.outer(e, xp)
        Epred[0, it + 1] = rmse_on_mask(Yrec, YorgInt, Mmiss)
        Efull[0, it + 1] = rmse_on_mask(C @ X, YorgInt, Mmiss)
    if return_state:
        return Epred, Efull, dict(C=C, X=X, Yrec=Yrec)
    return Epred, Efull
