"""Masked PSMF / rPSMF of the imputation experiment, on the device.

`ProbabilisticSequentialMatrixFactorizer` and `robust_PSMF` keep the signatures and return values
of ExperimentImpute/PSMF.py:40-95 and ExperimentImpute/rPSMF.py:40-148 (they are drop-in
replacements inside those scripts' `main()`); `impute_batch` runs many independent replicas
(seeds: own mask, C0, X0) in one launch -- one workgroup per replica -- which is how the
experiment's `for i in range(repeats)` loop maps onto a GPU.

The filter input is Y = YorgInt * M (data with the removed entries zeroed), exactly what the
experiment constructs (PSMF.py:141-148); the device reads YorgInt and the 0/1 mask.
"""

import ctypes as C

import numpy as np

from . import _capi

__all__ = ["ProbabilisticSequentialMatrixFactorizer", "robust_PSMF", "stochasticGradientStateSpaceMF", "temporalRegularizedMF",
           "impute_batch"]

METHODS = {"psmf": 0, "rpsmf": 1, "mle_smf": 2, "tmf": 3}


def _uniform_rho(R, d):
    if np.ndim(R) == 0:
        return float(R)
    R = np.asarray(R, dtype=float)
    dg = R if R.ndim == 1 else np.diagonal(R)
    if R.ndim == 2 and np.count_nonzero(R) != np.count_nonzero(dg):
        raise NotImplementedError("the device path assumes a diagonal R (as the reference's comments do)")
    if not np.all(dg == dg[0]):
        raise NotImplementedError("the device path needs R = rho * I")
    return float(dg[0])


def kernel_name(d, r):
    """Which device code psmf_impute_run uses for a (d, r) shape (psmf_impute_kernel_id of include/psmf_hip.h)."""
    cfg = _capi.PsmfImputeConfig(abi_version=_capi.ABI_VERSION, d=int(d), n=2, r=int(r), batch=1, method=0, n_iter=1)
    code = _capi.load_library().psmf_impute_kernel_id(C.byref(cfg))
    if code >= 300:
        return f"psmf_impute_kernel3<{code - 300}>"
    return {1: "psmf_impute_kernel", 2: "psmf_impute_kernel2", 4: "masked per-step engine"}.get(code, f"error {code}")


def replica_slices(batch, parts):
    """Contiguous, near-equal slices of `batch` replicas for `parts` devices / ranks (the repeats of ExperimentImpute are
    independent filters -- "replicas only", no exchange): [(start, stop), ...], empty slices when parts > batch."""
    base, extra = divmod(int(batch), int(parts))
    out, a = [], 0
    for i in range(int(parts)):
        b = a + base + (1 if i < extra else 0)
        out.append((a, b))
        a = b
    return out


def _impute_batch_devices(devices, YorgInt, M, Mmiss, C0, X0, *args, **kw):
    """impute_batch with the replicas dealt over several devices of THIS process: one host thread per device (psmf_impute_run is
    synchronous and releases the GIL), results merged in replica order.  One process per GPU works the same way with
    replica_slices(batch, world)[rank] and device=LOCAL_RANK."""
    import threading

    M, Mmiss, C0, X0 = (np.asarray(a) for a in (M, Mmiss, C0, X0))
    B = M.shape[0]
    sl = [(dev, a, b) for dev, (a, b) in zip(devices, replica_slices(B, len(devices))) if b > a]
    res, errs = [None] * len(sl), []

    def work(i, dev, a, b):
        try:
            res[i] = impute_batch(YorgInt, M[a:b], Mmiss[a:b], C0[a:b], X0[a:b], *args, device=int(dev), **kw)
        except BaseException as e:      # noqa: BLE001 -- re-raised by the caller's thread
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i, dev, a, b)) for i, (dev, a, b) in enumerate(sl)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    out = {}
    for k in res[0]:
        if isinstance(res[0][k], np.ndarray):
            out[k] = np.concatenate([r_[k] for r_ in res], axis=0)
    out["elapsed_ms"] = max(r_["elapsed_ms"] for r_ in res)      # the slices ran side by side
    out["kernel"] = res[0]["kernel"]
    out["devices"] = [(int(dev), a, b) for dev, a, b in sl]
    return out


def impute_batch(YorgInt, M, Mmiss, C0, X0, V, Q, R, P, sig, Iter, robust=False, lambda0=0.0, device=0,
                 want_bands=False, method=None):
    """Run `batch` replicas.  Reference layouts: YorgInt (d, n); M, Mmiss (batch, d, n);
    C0 (batch, d, r); X0 (batch, r, n).  Returns a dict with Epred, Efull (batch, Iter),
    inside (batch,), C (batch, d, r), X (batch, r, n), elapsed_ms and, if requested,
    Yrec / YrecL / YrecH (batch, d, n).  `device`: a HIP device ordinal, or a sequence of them -- the replicas are then dealt
    over those devices in contiguous slices (replica_slices) and run side by side."""
    if not np.isscalar(device):
        if np.asarray(M).ndim == 2:
            raise ValueError("a device list needs a batch of replicas")
        return _impute_batch_devices(list(device), YorgInt, M, Mmiss, C0, X0, V, Q, R, P, sig, Iter, robust=robust, lambda0=lambda0,
                                     want_bands=want_bands, method=method)
    lib = _capi.load_library()
    meth = METHODS[method] if method is not None else int(bool(robust))     # "mle_smf" / "tmf": the baseline filters
    YorgInt = np.asarray(YorgInt, dtype=np.float64)
    d, n = YorgInt.shape
    M = np.asarray(M)
    Mmiss = np.asarray(Mmiss)
    C0 = np.asarray(C0, dtype=np.float64)
    X0 = np.asarray(X0, dtype=np.float64)
    if M.ndim == 2:
        M, Mmiss, C0, X0 = M[None], Mmiss[None], C0[None], X0[None]
    B = M.shape[0]
    r = C0.shape[2]
    if M.shape != (B, d, n) or Mmiss.shape != (B, d, n) or C0.shape != (B, d, r) or X0.shape != (B, r, n):
        raise ValueError("inconsistent shapes")
    # time-major device layout: column t of the reference's (d, n) arrays is row t
    Yt = np.ascontiguousarray(YorgInt.T)
    Mt = np.ascontiguousarray(np.transpose(M != 0, (0, 2, 1)).astype(np.uint8))
    Mmt = np.ascontiguousarray(np.transpose(Mmiss != 0, (0, 2, 1)).astype(np.uint8))
    Cb = np.array(C0, dtype=np.float64, order="C", copy=True)   # the device writes the final C here; the caller's C0 is NOT mutated (PSMF.py:80 rebinds C)
    Xb = np.ascontiguousarray(np.transpose(X0, (0, 2, 1)))
    Vm, Pm, Qm = (np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(r, r) for a in (V, P, Q))
    Epred = np.zeros((B, Iter))
    Efull = np.zeros((B, Iter))
    inside = np.zeros(B)
    bands = [np.zeros((B, n, d)) for _ in range(3)] if want_bands else [None, None, None]
    cfg = _capi.PsmfImputeConfig(abi_version=_capi.ABI_VERSION, d=d, n=n, r=r, batch=B, method=meth,
                                 n_iter=int(Iter), device=int(device), want_bands=int(want_bands),
                                 sig=float(sig), lambda0=float(lambda0))
    ms = C.c_float()
    status = np.zeros(B, dtype=np.int32)
    dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    rc = lib.psmf_impute_run(C.byref(cfg), dp(Yt), up(Mt), up(Mmt), dp(Cb), dp(Xb), dp(Vm), dp(Pm), dp(Qm),
                             _uniform_rho(R, d), dp(Epred), dp(Efull), dp(inside), dp(bands[0]), dp(bands[1]),
                             dp(bands[2]), status.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms))
    if rc != _capi.OK:
        msg = lib.psmf_last_error(None).decode()
        if rc == _capi.ERR_NUMERIC:
            raise np.linalg.LinAlgError(msg)
        if rc == _capi.ERR_ARG:
            raise ValueError(msg)
        raise _capi.PsmfError(f"psmf_impute_run failed ({rc}): {msg}")
    # a replica whose r x r system broke down has NaN results and status != 0; the others are complete (the reference's
    # repeats loop records NaN for such a repeat and carries on, ExperimentImpute/rPSMF.py:236-243)
    out = dict(Epred=Epred, Efull=Efull, inside=inside, C=Cb, X=np.transpose(Xb, (0, 2, 1)), elapsed_ms=ms.value, status=status,
               kernel=kernel_name(d, r))
    if want_bands:
        out.update(Yrec=np.transpose(bands[0], (0, 2, 1)), YrecL=np.transpose(bands[1], (0, 2, 1)),
                   YrecH=np.transpose(bands[2], (0, 2, 1)))
    return out


def _single(Y, C, X, d, n, r, M, Mmiss, V, Q, R, P, sig, Iter, YorgInt, Einit, robust, lambda0, method=None):
    Y = np.asarray(Y, dtype=float)
    YorgInt = np.asarray(YorgInt, dtype=float)
    if Y.shape != (d, n) or C.shape != (d, r) or X.shape != (r, n):
        raise ValueError("shape mismatch with d, n, r")
    if not np.array_equal(Y, YorgInt * (np.asarray(M) != 0)):
        raise ValueError("the device path requires Y == YorgInt * M (as the experiment constructs it)")
    res = impute_batch(YorgInt, M, Mmiss, C, X, V, Q, R, P, sig, Iter, robust=robust, lambda0=lambda0, method=method)
    X[...] = res["X"][0]  # the reference updates the caller's X in place (PSMF.py:74)
    if res["status"][0] != 0:
        # the reference's functions return NaN errors when the recursion diverges (its main() then records NaN for the repeat,
        # rPSMF.py:236-243): same here, with NaN coverage
        res["inside"][0] = np.nan
    Epred = np.zeros((1, Iter + 1))
    Efull = np.zeros((1, Iter + 1))
    Epred[0, 0] = Efull[0, 0] = Einit
    Epred[0, 1:] = res["Epred"][0]
    Efull[0, 1:] = res["Efull"][0]
    RunTime = np.zeros((1, Iter + 1))
    RunTime[0, 1:] = 1e-3 * res["elapsed_ms"] * np.arange(1, Iter + 1) / Iter  # passes are not timed separately
    return Epred, Efull, RunTime, float(res["inside"][0])


def ProbabilisticSequentialMatrixFactorizer(Y, C, X, d, n, r, M, Mmiss, lam, V, Q, R, P, sig, Iter, YorgInt,
                                            Einit):
    """ExperimentImpute/PSMF.py:40-95 on the device.  `lam` is unused, as in the reference."""
    return _single(Y, C, X, d, n, r, M, Mmiss, V, Q, R, P, sig, Iter, YorgInt, Einit, False, 0.0)


def robust_PSMF(Y, C, X, d, n, r, M, Mmiss, V, Q0, R0, P, lambda0, sig, Iter, YorigInt, Einit):
    """ExperimentImpute/rPSMF.py:40-148 on the device."""
    return _single(Y, C, X, d, n, r, M, Mmiss, V, Q0, R0, P, sig, Iter, YorigInt, Einit, True, lambda0)


def stochasticGradientStateSpaceMF(Y, C, X, d, n, r, M, Mmiss, lam, Q, R, P, sig, Iter, YorgInt, Einit):
    """MLE-SMF, ExperimentImpute/MLESMF.py:40-92, on the device (same arguments and return tuple; `lam` is unused there)."""
    return _single(Y, C, X, d, n, r, M, Mmiss, np.eye(r), Q, R, P, sig, Iter, YorgInt, Einit, False, 0.0, method="mle_smf")


def temporalRegularizedMF(Y, C, X, d, n, r, M, Mmiss, lam, R, Iter, YorgInt, Einit):
    """TMF, ExperimentImpute/TMF.py:30-73, on the device.  Returns (Epred, Efull, RunTime) like the reference
    (`lam` and `R` are unused there)."""
    I = np.eye(r)
    ep, ef, rt, _ = _single(Y, C, X, d, n, r, M, Mmiss, I, I, 1.0, I, 0.0, Iter, YorgInt, Einit, False, 0.0, method="tmf")
    return ep, ef, rt
