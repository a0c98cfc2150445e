"""ctypes binding of include/psmf_hip.h (libpsmf_hip.so).  numpy + ctypes only.

There is no CPU fallback behind this module: if the shared library is missing, or no HIP
device is visible, the device path raises.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpsmf_hip.so")

ABI_VERSION = 3
RMAX = 64
F32, F64 = 0, 1
DYN_RANDOM_WALK, DYN_COS_PHASE, DYN_SCALED_WALK, DYN_SINUSOID, DYN_FOURIER, DYN_HOST = 0, 1, 2, 3, 4, 5
UNIQUE_ID_BYTES = 128

OK, ERR_ARG, ERR_HIP, ERR_RCCL, ERR_NUMERIC, ERR_STATE, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6


class PsmfError(RuntimeError):
    pass


class PsmfConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "d", "r", "row0", "d_local", "robust", "coef_update", "eta_full", "pbar_predict",
        "fixed_lambda", "dyn_kind", "n_theta", "storage", "store_y_pred", "recursive", "update_every",
        "gram_refresh", "device", "use_graph", "n_workgroups", "engine", "dyn_flags", "dyn_terms", "nonuniform_R", "masked")] + [(n, C.c_double) for n in (
        "alpha", "beta", "adam_lr", "adam_lr_end", "adam_lr_steps", "adam_b1", "adam_b2")]


class PsmfImputeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "d", "n", "r", "batch", "method", "n_iter", "device", "want_bands")] + [
        ("sig", C.c_double), ("lambda0", C.c_double)]


_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int64)   # psmf_allreduce_fn

# name -> (restype, argtypes); every symbol include/psmf_hip.h declares
SIGNATURES = {
    "psmf_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(PsmfConfig)]),
    "psmf_destroy": (None, [C.c_void_p]),
    "psmf_last_error": (C.c_char_p, [C.c_void_p]),
    "psmf_build_id": (C.c_char_p, []),
    "psmf_device_count": (C.c_int, []),
    "psmf_set_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _dp]),
    "psmf_zero_gradsum": (C.c_int, [C.c_void_p]),
    "psmf_get_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "psmf_set_adam": (C.c_int, [C.c_void_p, _dp, _dp]),
    "psmf_set_schedules": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64]),
    "psmf_set_q_matrix_schedule": (C.c_int, [C.c_void_p, _dp, C.c_int64]),
    "psmf_set_row_noise": (C.c_int, [C.c_void_p, _dp, C.c_double]),
    "psmf_set_noise_rotation": (C.c_int, [C.c_void_p, _dp, _dp]),
    "psmf_step_host": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp]),
    "psmf_project": (C.c_int, [C.c_void_p, _dp, C.c_int64, _dp]),
    "psmf_predict_sq_error": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp, _dp]),
    "psmf_upload_series": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64]),
    "psmf_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "psmf_sync": (C.c_int, [C.c_void_p]),
    "psmf_download_y_pred": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64]),
    "psmf_download_mu": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int64]),
    "psmf_predict": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp]),
    "psmf_sq_error": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp]),
    "psmf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "psmf_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "psmf_comm_init_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "psmf_run_timed": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_float)]),
    "psmf_time_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "psmf_geometry": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "psmf_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "psmf_comm_abort": (C.c_int, [C.c_void_p]),
    "psmf_filter_kernel": (C.c_int, [C.c_void_p]),
    "psmf_filter_kernel_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int]),
    "psmf_measure_copy_bandwidth": (C.c_int, [C.c_int, C.c_size_t, C.c_int, _dp]),
    "psmf_impute_run": (C.c_int, [C.POINTER(PsmfImputeConfig), _dp, _u8p, _u8p, _dp, _dp, _dp, _dp, _dp,
                                  C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "psmf_impute_kernel_id": (C.c_int, [C.POINTER(PsmfImputeConfig)]),
    "psmf_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "psmf_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "psmf_upload_mask": (C.c_int, [C.c_void_p, _u8p, C.c_int64, C.c_int64]),
    "psmf_set_step_size": (C.c_int, [C.c_void_p, C.c_double]),
    "psmf_masked_metrics": (C.c_int, [C.c_void_p, _u8p, C.c_int64, C.c_int64, C.c_double, _dp]),
    "psmf_download_step_scalars": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int64]),
}

_lib = None


def load_library():
    """dlopen libpsmf_hip.so and type every entry point.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PsmfError(
            f"{LIB_PATH} is missing: build it with `python -m rpsmf_amd.build` "
            "(hipcc --offload-arch=gfx950).  The device path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dyn_n_theta(kind, r, flags=0, terms=0):
    """length of theta for a device dynamics kind (psmf_dyn_kind in include/psmf_hip.h)"""
    if kind == DYN_COS_PHASE:
        return r
    if kind == DYN_SCALED_WALK:
        return r * r + (r if flags & 1 else 0)
    if kind == DYN_SINUSOID:
        return (r * r if flags & 1 else 0) + r + (r if flags & 2 else 0)
    if kind == DYN_FOURIER:
        return terms * (2 * r * r + 4 * r)
    return 0


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


class DeviceFilter:
    """One device-resident filter (or row shard).  Thin, argument-checking wrapper of the C ABI."""

    def __init__(self, d, r, *, robust=False, coef_update=True, eta_full=True, pbar_predict=True,
                 fixed_lambda=False, dyn_kind=DYN_RANDOM_WALK, storage="f32", store_y_pred=True,
                 recursive=False, update_every=1, gram_refresh=0, device=0, use_graph=True,
                 n_workgroups=0, engine="auto", alpha=1.0, beta=1.0, adam_lr=1e-3, adam_lr_end=0.0, adam_lr_steps=0.0,
                 adam_b1=0.9, adam_b2=0.999, row0=0, d_local=None, dyn_flags=0, dyn_terms=0, nonuniform_R=False, masked=False):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.d, self.r = int(d), int(r)
        self.row0 = int(row0)
        self.d_local = int(d if d_local is None else d_local)
        self.dyn_kind = int(dyn_kind)
        self.n_theta = dyn_n_theta(self.dyn_kind, self.r, int(dyn_flags), int(dyn_terms))
        if storage == "auto":
            # f32 where the blocked engine runs (C is rounded once per block of 64 - r timesteps: errors ~1e-6);
            # f64 where the per-step engine runs (one rounding of C per timestep would breach the 1e-5 bar around
            # k = 300)
            blocked = (engine in ("auto", "block", 0, 2) and self.r <= 32 and self.dyn_kind != DYN_HOST and not nonuniform_R
                       and not masked and os.environ.get("PSMF_ENGINE") != "1")
            storage = "f32" if blocked else "f64"
        self.storage = F64 if storage in ("f64", F64, np.float64) else F32
        self.store_y_pred = bool(store_y_pred)
        cfg = PsmfConfig(
            abi_version=ABI_VERSION, d=self.d, r=self.r, row0=self.row0, d_local=self.d_local,
            robust=int(robust), coef_update=int(coef_update), eta_full=int(eta_full),
            pbar_predict=int(pbar_predict), fixed_lambda=int(fixed_lambda), dyn_kind=int(dyn_kind),
            n_theta=self.n_theta, storage=self.storage, store_y_pred=int(store_y_pred),
            recursive=int(recursive), update_every=int(update_every), gram_refresh=int(gram_refresh),      # recursive: 1 / True in-loop Adam, 2 in-loop SGD
            device=int(device), use_graph=int(use_graph), n_workgroups=int(n_workgroups),
            engine={"auto": 0, "step": 1, "block": 2}.get(engine, engine), dyn_flags=int(dyn_flags), dyn_terms=int(dyn_terms), nonuniform_R=int(bool(nonuniform_R)), masked=int(masked),
            alpha=float(alpha), beta=float(beta), adam_lr=float(adam_lr), adam_lr_end=float(adam_lr_end),
            adam_lr_steps=float(adam_lr_steps), adam_b1=float(adam_b1), adam_b2=float(adam_b2))
        rc = self._lib.psmf_create(C.byref(self._h), C.byref(cfg))
        if rc != OK:
            msg = self._lib.psmf_last_error(None).decode()
            self._h = C.c_void_p()
            raise (ValueError if rc == ERR_ARG else PsmfError)(f"psmf_create failed ({rc}): {msg}")
        self.T = 0

    # -- plumbing
    def _check(self, rc):
        if rc == OK:
            return
        msg = self._lib.psmf_last_error(self._h).decode()
        if rc == ERR_NUMERIC:
            raise np.linalg.LinAlgError(msg)
        if rc == ERR_ARG:
            raise ValueError(msg)
        raise PsmfError(f"libpsmf_hip error {rc}: {msg}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.psmf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state
    def set_state(self, C_=None, V=None, P=None, Q=None, mu=None, rho=None, lambda0=None, theta=None):
        r, dl = self.r, self.d_local
        C_ = _f64(C_, (dl, r))
        V, P, Q = _f64(V, (r, r)), _f64(P, (r, r)), _f64(Q, (r, r))
        mu = _f64(mu, (r,))
        theta = _f64(theta, (self.n_theta,)) if self.n_theta else None
        nan = float("nan")
        self._check(self._lib.psmf_set_state(
            self._h, _ptr(C_), _ptr(V), _ptr(P), _ptr(Q), _ptr(mu),
            nan if rho is None else float(rho), nan if lambda0 is None else float(lambda0), _ptr(theta)))

    def zero_gradsum(self):
        self._check(self._lib.psmf_zero_gradsum(self._h))

    def set_adam(self, m, v):
        m, v = _f64(m, (self.n_theta,)), _f64(v, (self.n_theta,))
        self._check(self._lib.psmf_set_adam(self._h, _ptr(m), _ptr(v)))

    def get_state(self, want_C=True):
        r, dl = self.r, self.d_local
        out = dict(
            C=np.empty((dl, r)) if want_C else None, V=np.empty((r, r)), P=np.empty((r, r)),
            Q=np.empty((r, r)), mu=np.empty(r), theta=np.zeros(self.n_theta), gradsum=np.zeros(self.n_theta))
        sc = np.empty(8)
        self._check(self._lib.psmf_get_state(
            self._h, _ptr(out["C"]), _ptr(out["V"]), _ptr(out["P"]), _ptr(out["Q"]), _ptr(out["mu"]),
            _ptr(out["theta"]) if self.n_theta else None, _ptr(out["gradsum"]) if self.n_theta else None,
            _ptr(sc)))
        out.update(rho=sc[0], lam=sc[1], s=sc[2], eta=sc[3], N=sc[4], phi=sc[5], omega=sc[6], k=int(sc[7]))
        return out

    # -- series / run
    def upload_series(self, Y, t0=0, T_total=None):
        Y = np.asarray(Y)
        if Y.ndim != 2 or Y.shape[1] != self.d_local:
            raise ValueError(f"series must be (T, {self.d_local}) time-major, got {Y.shape}")
        if Y.dtype == np.float32:
            dt = F32
        else:
            Y = Y.astype(np.float64, copy=False)
            dt = F64
        Y = np.ascontiguousarray(Y)
        nt = Y.shape[0]
        T_total = t0 + nt if T_total is None else int(T_total)
        self._check(self._lib.psmf_upload_series(self._h, Y.ctypes.data_as(C.c_void_p), dt, t0, nt, T_total))
        self.T = max(self.T, T_total)

    def upload_mask(self, M, t0=0):
        """Observation mask of the steps t0+1 .. t0+nt (masked handles): (nt, d_local), nonzero = observed."""
        M = np.ascontiguousarray(np.asarray(M) != 0, dtype=np.uint8)
        if M.ndim != 2 or M.shape[1] != self.d_local:
            raise ValueError(f"mask must be (T, {self.d_local}) time-major, got {M.shape}")
        self._check(self._lib.psmf_upload_mask(self._h, M.ctypes.data_as(_u8p), int(t0), M.shape[0]))

    def set_step_size(self, gam):
        """masked = 2 / 3 (MLE-SMF / TMF): step size of the gradient update of C for the runs that follow."""
        self._check(self._lib.psmf_set_step_size(self._h, float(gam)))

    def masked_metrics(self, Mmiss, sig, t0=0):
        """(sum (y_hat - y)^2, sum (C x_t - y)^2, entries inside their band, count) over the held-out entries Mmiss (nt, d_local)
        of this handle's rows, reduced on the device (ExperimentImpute/common.py:79-94)."""
        Mm = np.ascontiguousarray(np.asarray(Mmiss) != 0, dtype=np.uint8)
        if Mm.ndim != 2 or Mm.shape[1] != self.d_local:
            raise ValueError(f"held-out mask must be (T, {self.d_local}) time-major, got {Mm.shape}")
        out = np.empty(4)
        self._check(self._lib.psmf_masked_metrics(self._h, Mm.ctypes.data_as(_u8p), int(t0), Mm.shape[0], float(sig), _ptr(out)))
        return out

    def step_scalars(self, t0, nt):
        """(s_t, eta_t) of the steps t0+1 .. t0+nt of a masked handle -> (nt, 2)"""
        out = np.empty((nt, 2))
        self._check(self._lib.psmf_download_step_scalars(self._h, _ptr(out), int(t0), int(nt)))
        return out

    def set_row_noise(self, rho_rows, rho_mean=None):
        """diag(R) of this handle's rows (nonuniform_R handles); rho_mean = sum(diag R) over ALL rows / d (default: these rows')"""
        rho_rows = _f64(rho_rows, (self.d_local,))
        if rho_mean is None:
            rho_mean = float(rho_rows.sum()) / self.d
        self._check(self._lib.psmf_set_row_noise(self._h, _ptr(rho_rows), float(rho_mean)))

    def set_noise_rotation(self, U, lam):
        """A non-diagonal R = U diag(lam) U^T (eigenvectors in the columns of U): the handle keeps the series and C rotated and
        runs the non-uniform-diagonal step (psmf_set_noise_rotation); one shard, before set_state / upload_series."""
        U, lam = _f64(U, (self.d, self.d)), _f64(lam, (self.d,))
        self._check(self._lib.psmf_set_noise_rotation(self._h, _ptr(U), _ptr(lam)))

    def set_schedules(self, rho_k=None, q_k=None):
        """R_k = rho_k[k] I, Q_k = q_k[k] Q for the 1-based step k (entry 0 unused); None = constant."""
        rho_k, q_k = _f64(rho_k), _f64(q_k)
        n = max(0 if rho_k is None else rho_k.size, 0 if q_k is None else q_k.size)
        for a in (rho_k, q_k):
            if a is not None and a.size != n:
                raise ValueError("schedules must have the same length")
        self._check(self._lib.psmf_set_schedules(self._h, _ptr(rho_k), _ptr(q_k), n))

    def set_q_matrix_schedule(self, Q_k=None):
        """Q_k [n, r, r] as a matrix of its own per 1-based step k (matrix 0 unused): a Q[k] that is not a multiple of Q[1]
        (psmf.py:115).  Per-step engine only (engine="step"); None drops the schedule."""
        if Q_k is None:
            self._check(self._lib.psmf_set_q_matrix_schedule(self._h, None, 0))
            return
        Q_k = _f64(Q_k)
        if Q_k.ndim != 3 or Q_k.shape[1:] != (self.r, self.r):
            raise ValueError(f"Q_k: expected [n, {self.r}, {self.r}], got {Q_k.shape}")
        self._check(self._lib.psmf_set_q_matrix_schedule(self._h, _ptr(Q_k), Q_k.shape[0]))

    def step_host(self, k, mu_bar, P_bar, want_PQ=True):
        """One timestep k -> k + 1 with host-evaluated mu_bar [r], P_bar [r, r] (dyn_kind = DYN_HOST).
        Returns (mu, gf, P, Q) of the finished step."""
        r = self.r
        mu_bar, P_bar = _f64(mu_bar, (r,)), _f64(P_bar, (r, r))
        mu, gf = np.empty(r), np.empty(r)
        P = np.empty((r, r)) if want_PQ else None
        Q = np.empty((r, r)) if want_PQ else None
        self._check(self._lib.psmf_step_host(self._h, int(k), _ptr(mu_bar), _ptr(P_bar), _ptr(mu), _ptr(gf), _ptr(P), _ptr(Q)))
        return mu, gf, P, Q

    def project(self, mu):
        """C @ mu[q] for each of the n given r-vectors -> (n, d_local)"""
        mu = _f64(mu)
        mu = mu.reshape(-1, self.r)
        out = np.empty((mu.shape[0], self.d_local))
        self._check(self._lib.psmf_project(self._h, _ptr(mu), mu.shape[0], _ptr(out)))
        return out

    def predict_sq_error(self, T, Y_true):
        """sum of (C mu_pred_q - Y_true[q])^2 over the roll-out window; Y_true: (n_pred, d_local)"""
        Y_true = _f64(Y_true)
        if Y_true.ndim != 2 or Y_true.shape[1] != self.d_local:
            raise ValueError(f"held-out observations must be (n_pred, {self.d_local})")
        v = C.c_double()
        self._check(self._lib.psmf_predict_sq_error(self._h, int(T), Y_true.shape[0], _ptr(Y_true), C.byref(v)))
        return v.value

    def run(self, k_begin, k_end, sync=True):
        self._check(self._lib.psmf_run(self._h, int(k_begin), int(k_end)))
        if sync:
            self.sync()

    def sync(self):
        self._check(self._lib.psmf_sync(self._h))

    def run_timed(self, k_begin, k_end):
        ms = C.c_float()
        self._check(self._lib.psmf_run_timed(self._h, int(k_begin), int(k_end), C.byref(ms)))
        return ms.value

    def time_kernel(self, which, iters=200):
        us = C.c_float()
        self._check(self._lib.psmf_time_kernel(self._h, int(which), int(iters), C.byref(us)))
        return us.value

    def geometry(self):
        g = (C.c_int32 * 7)()
        self._check(self._lib.psmf_geometry(self._h, g))
        kern = {0: "psmf_sweep_solve", 1: "psmf_blk_filter", 2: "psmf_blk_filter2", 3: "psmf_blk_filter3", 4: "psmf_blk_filter3s",
                5: "psmf_blk_filter4", 6: "psmf_blk_filter4s", 7: "psmf_blk_filter5", 8: "psmf_blk_filter6", 9: "psmf_blk_filter6d", 10: "psmf_blk_filter7", 11: "psmf_pstep_k"}.get(self._lib.psmf_filter_kernel(self._h), "?")
        return dict(n_sweep_wg=g[0], rows_per_wg=g[1], row_stride=g[2], lanes_per_row=g[3], graph_chunk=g[4],
                    engine={1: "step", 2: "block"}.get(g[5], g[5]), block_steps=g[6], filter_kernel=kern)

    def counters(self, reset=False):
        c = (C.c_int64 * 8)()
        self._check(self._lib.psmf_counters(self._h, c, int(bool(reset))))
        # filter_launches counts BLOCKS (one launch each unless the blocks of a run are chained into one launch);
        # filter_kernel_launches the launches of the filter kernel, filter_kernel_us_mean their mean duration
        return dict(ns_steps=c[0], sweep_steps=c[1], ns_iterations=c[2], ns_failed=c[3], filter_launches=c[7],
                    filter_us_mean=0.01 * c[4] / max(1, c[7]), filter_gap_us_mean=0.01 * c[5] / max(1, c[7] - 1),
                    filter_kernel_launches=c[6], filter_kernel_us_mean=0.01 * c[4] / max(1, c[6]))

    def filter_kernel_time(self, reset=False):
        """(launches, total_ms) of the chained filter-kernel launches since the last reset (HIP events on their stream)."""
        n, ms = C.c_int64(0), C.c_double(0.0)
        self._check(self._lib.psmf_filter_kernel_time(self._h, C.byref(n), C.byref(ms), int(bool(reset))))
        return int(n.value), float(ms.value)

    def y_pred(self, t0, nt, dtype=np.float64):
        out = np.empty((nt, self.d_local), dtype=dtype)
        dt = F32 if out.dtype == np.float32 else F64
        self._check(self._lib.psmf_download_y_pred(self._h, out.ctypes.data_as(C.c_void_p), dt, int(t0), int(nt)))
        return out

    def mu_history(self, k0, nk):
        out = np.empty((nk, self.r))
        self._check(self._lib.psmf_download_mu(self._h, _ptr(out), int(k0), int(nk)))
        return out

    def predict(self, T, n_pred):
        out = np.empty((n_pred, self.d_local))
        self._check(self._lib.psmf_predict(self._h, int(T), int(n_pred), _ptr(out)))
        return out

    def sq_error(self, t0, nt):
        v = C.c_double()
        self._check(self._lib.psmf_sq_error(self._h, int(t0), int(nt), C.byref(v)))
        return v.value

    # -- multi-GPU
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = lib.psmf_comm_unique_id(buf)
        if rc != OK:
            raise PsmfError("psmf_comm_unique_id failed: " + lib.psmf_last_error(None).decode())
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._check(self._lib.psmf_comm_init(self._h, int(nranks), int(rank), buf))

    def comm_abort(self):
        """drop the RCCL communicator without waiting for its peers (failure path of a multi-rank start)"""
        self._check(self._lib.psmf_comm_abort(self._h))

    def comm_info(self):
        """What the exchanges of this handle run on: transport ("rccl" / "host" / None), ranks, rank and device as the
        communicator itself reports them (RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice)."""
        v = (C.c_int32 * 4)()
        self._check(self._lib.psmf_comm_info(self._h, v))
        return dict(transport={0: None, 1: "rccl", 2: "host"}[v[0]], ranks=int(v[1]), rank=int(v[2]), device=int(v[3]))

    def comm_init_host(self, nranks, rank, allreduce):
        """Host-mediated communicator: `allreduce(vec) -> vec` (numpy float64, same length) is called wherever the sharded
        engine needs its sum over the ranks; it must return the same bits on every rank."""
        def _cb(_ctx, buf, count):
            try:
                v = np.ctypeslib.as_array(buf, shape=(int(count),))
                v[:] = allreduce(v.copy())
                return 0
            except Exception:          # no exception may cross the C boundary
                import traceback

                traceback.print_exc()
                return 1

        self._allreduce_cb = ALLREDUCE_FN(_cb)      # keep the trampoline alive as long as the handle
        self._check(self._lib.psmf_comm_init_host(self._h, int(nranks), int(rank), C.cast(self._allreduce_cb, C.c_void_p), None))


def measure_copy_bandwidth(device=0, nbytes=1 << 30, iters=20):
    """GB/s (read + written) of a streaming copy kernel on `device`."""
    lib = load_library()
    v = C.c_double()
    rc = lib.psmf_measure_copy_bandwidth(int(device), int(nbytes), int(iters), C.byref(v))
    if rc != OK:
        raise PsmfError("psmf_measure_copy_bandwidth failed: " + lib.psmf_last_error(None).decode())
    return v.value


def device_pci_bus_id(device=0):
    buf = C.create_string_buffer(32)
    rc = load_library().psmf_device_pci_bus_id(int(device), buf, 32)
    if rc != OK:
        raise PsmfError("psmf_device_pci_bus_id failed: " + load_library().psmf_last_error(None).decode())
    return buf.value.decode()


def device_count():
    return load_library().psmf_device_count()
