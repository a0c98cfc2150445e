/*
 * psmf_hip.h -- C ABI of the MI355X (gfx950) PSMF / rPSMF filter library  (libpsmf_hip.so)
 *
 * The reference (alan-turing-institute/rPSMF) is pure Python and has no FFI layer; the
 * functions below are what a ctypes binding for its hot path binds instead of executing the
 * Python/numpy bodies cited next to each entry point (paths relative to the reference root).
 * Plain pointers and sizes only; no exceptions cross the boundary; every function returns
 * PSMF_OK (0) or a negative psmf_status, and psmf_last_error() gives the message.
 *
 * Conventions
 *   - the caller owns all host buffers; the library owns all device memory;
 *   - host matrices are row-major float64 unless a dtype argument says otherwise;
 *   - one handle = one filter (or one row-shard of a filter on one GPU); a handle is not
 *     thread-safe, independent handles may be used from different threads;
 *   - calls are asynchronous on the handle's HIP stream unless they return host data.
 */
#ifndef PSMF_HIP_H
#define PSMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSMF_ABI_VERSION 3
#define PSMF_RMAX 64 /* largest supported rank r */
#define PSMF_ROTATION_DMAX 32768 /* largest d of psmf_set_noise_rotation (the d x d eigenvector matrix stays resident) */

typedef enum {
  PSMF_OK = 0,
  PSMF_ERR_ARG = -1,         /* bad argument / shape / unsupported configuration           */
  PSMF_ERR_HIP = -2,         /* HIP runtime error                                          */
  PSMF_ERR_RCCL = -3,        /* RCCL error                                                 */
  PSMF_ERR_NUMERIC = -4,     /* singular r x r system (numpy.linalg.LinAlgError upstream)  */
  PSMF_ERR_STATE = -5,       /* call sequence error (e.g. run before set_state)            */
  PSMF_ERR_NO_DEVICE = -6    /* no HIP device visible                                      */
} psmf_status;

typedef enum { PSMF_F32 = 0, PSMF_F64 = 1 } psmf_dtype;

/* State transition f(theta, x, t); theta packs its blocks in the reference's order (BaseNonLinearity.dims).
 * PSMF_DYN_RANDOM_WALK  f = x                                   pypsmf/psmf/nonlinearities.py:42-56
 * PSMF_DYN_COS_PHASE    f = cos(2 pi theta t + x)               ExperimentSynthetic/synthetic_psmf.py:105-106   theta: [r]
 * PSMF_DYN_SCALED_WALK  f = A x (+ b)                           nonlinearities.py:59-78     theta: A [r*r], b [r] if dyn_flags & 1
 * PSMF_DYN_SINUSOID     f = [A] sin(2 pi b t + [c o] x)         nonlinearities.py:81-114    theta: A [r*r] if dyn_flags & 1 (scaled),
 *                                                                                           b [r], c [r] if dyn_flags & 2 (phased)
 * PSMF_DYN_FOURIER      f = sum_n A_n sin(2 pi b_n t + c_n o x) + D_n cos(2 pi e_n t + f_n o x)   nonlinearities.py:117-150
 *                                                               theta: A_1, D_1, .., A_N, D_N [r*r each], then b_n, c_n, e_n, f_n [r each]
 *                                                               per n; N = dyn_terms <= 4
 * PSMF_DYN_HOST         any callable: the host evaluates mu_bar = f(theta, mu, k) and P_bar = F P F^T + Q (r-sized) and
 *                       advances the device one timestep at a time with psmf_step_host (the d-sized work stays on the device)
 * Kinds 0-4 are evaluated inside the device time loop of either engine, with analytic Jacobians df/dx, df/dtheta (the reference
 * uses autograd, psmf.py:41-44): the blocked engine (r <= 32) or the serial stage of the per-step engine's launched form (r > 32,
 * a non-uniform R, engine = 1; the persistent per-step kernel takes kinds 0, 1).  Kind 5 is the per-step engine's. */
typedef enum { PSMF_DYN_RANDOM_WALK = 0, PSMF_DYN_COS_PHASE = 1, PSMF_DYN_SCALED_WALK = 2, PSMF_DYN_SINUSOID = 3,
               PSMF_DYN_FOURIER = 4, PSMF_DYN_HOST = 5 } psmf_dyn_kind;

typedef struct psmf_filter* psmf_handle;

/* Mode table: which hook configuration of PSMFIter / rPSMFIter runs fused. */
typedef struct {
  int32_t abi_version;   /* PSMF_ABI_VERSION                                                  */
  int32_t d;             /* global number of rows of C (= len(y_k))                           */
  int32_t r;             /* rank, 1..PSMF_RMAX                                                */
  int32_t row0;          /* first global row held by this handle (row sharding)               */
  int32_t d_local;       /* rows held by this handle (== d when not sharded)                  */
  int32_t robust;        /* 0: PSMFIter  psmf.py:90-180;  1: rPSMFIter  rpsmf.py:116-184      */
  int32_t coef_update;   /* 1: Kalman update of (mu, P) psmf.py:140-165; 0: mu_k = mu_bar,
                            P_k = P_bar  (synthetic_psmf.py:89-98)                            */
  int32_t eta_full;      /* 1: eta = tr(R + C Pbar C^T)/d  psmf.py:121-125; 0: tr(R)/d        */
  int32_t pbar_predict;  /* 1: Pbar = F P F^T + Q  psmf.py:107-115; 0: Pbar = P_{k-1}         */
  int32_t fixed_lambda;  /* rpsmf.py:36-40                                                    */
  int32_t dyn_kind;      /* psmf_dyn_kind                                                     */
  int32_t n_theta;       /* length of theta: 0 random walk / host, r cos-phase, see psmf_dyn_kind       */
  int32_t storage;       /* psmf_dtype of C, y, y_pred in HBM (arithmetic on r x r is f64)    */
  int32_t store_y_pred;  /* keep y_hat_k = C_{k-1} mu_bar_k for every step (psmf.py:93)       */
  int32_t recursive;     /* 1: PSMFRecursive -- Adam step on theta inside the time loop every
                            `update_every` steps (psmf.py:287-304); 2: the same with plain SGD
                            (psmf.py:244-248; learning rate = adam_lr / its schedule)           */
  int32_t update_every;
  int32_t gram_refresh;  /* recompute G = C^T C exactly every this many steps (0 = only at
                            set_state); between refreshes G is updated algebraically           */
  int32_t device;        /* HIP device ordinal                                                */
  int32_t use_graph;     /* 1: replay the per-step launches from a hipGraph                   */
  int32_t n_workgroups;  /* row-sweep workgroups, 0 = auto                                    */
  int32_t engine;        /* 0 = auto, 1 = per-step engine (one row sweep per timestep),
                            2 = exact time-blocked engine (one Gram Z^T Z per block of min(64 - r, 48) steps,
                            the steps in coefficient space; needs r <= 32)                      */
  int32_t dyn_flags;     /* see psmf_dyn_kind                                                  */
  int32_t dyn_terms;     /* PSMF_DYN_FOURIER: N                                                */
  int32_t nonuniform_R;  /* 1: R = rho * diag(rho_rows) with per-row values uploaded by psmf_set_row_noise
                            (psmf.py:140-153 takes any diagonal R); per-step engine, weighted Gram every step */
  int32_t masked;        /* 1: every step has a 0/1 observation mask over the rows (psmf_upload_mask): the masked filter of
                            ExperimentImpute/PSMF.py:59-84, rPSMF.py:75-135 -- e = m o (y - y_hat), rows with m_i = 0 not
                            updated, Gram / innovation covariance over the observed rows only, eta and lambda with d (not the
                            observed count), y_hat stored unmasked.  Per-step engine, one masked Gram pass per step; needs the
                            full filter (coef_update, eta_full, pbar_predict = 1), uniform R, store_y_pred = 1.
                            2 / 3: the baseline filters that share these contractions -- 2 = MLE-SMF (MLESMF.py:57-88: weights m_i / rho,
                            C += gam / eta (m o e) x_p^T, bands -+ sig sqrt(eta)), 3 = TMF (TMF.py:47-66: Pbar = I / nu with Q = I / nu and
                            P = 0 from the caller, kappa = 1, C += gam (m o e) x_p^T); gam from psmf_set_step_size */
  double alpha, beta;    /* rPSMF scaling factors (rpsmf.py:45-51), 1.0 unless use_scaling     */
  double adam_lr, adam_lr_end, adam_lr_steps; /* lr (Constant) or lr_start/lr_end/steps
                            (ExponentialLearningRate, learning_rate.py:20-27; steps = 0 ->
                            constant), used when recursive = 1                                 */
  double adam_b1, adam_b2;
} psmf_config;

/* ---- lifetime -------------------------------------------------------------------------- */
/* replaces PSMFIter.__init__ / rPSMFIter.__init__  (psmf.py:18-46, rpsmf.py:12-51) */
int psmf_create(psmf_handle* out, const psmf_config* cfg);
void psmf_destroy(psmf_handle h);
const char* psmf_last_error(psmf_handle h);
/* SHA-256 (hex) of the sources and compiler flags this library was built from (rpsmf_amd/build.py compares it with the sources on
 * disk: a stale binary is rebuilt, never silently used).  (New: the reference is interpreted Python.) */
const char* psmf_build_id(void); /* h may be NULL: error of the last failed create */
int psmf_device_count(void);

/* ---- state ----------------------------------------------------------------------------- */
/* Any pointer may be NULL = leave that part unchanged.  C: d_local x r;  V, P, Q: r x r;
 * mu: r; theta: n_theta.  rho = diag(R) (uniform), lambda0 = Student-t dof (rPSMF).
 * Pass a NaN for rho / lambda0 to leave them unchanged.
 * replaces step_reset (psmf.py:75-83, rpsmf.py:106-114) and the constructors' state. */
int psmf_set_state(psmf_handle h, const double* C, const double* V, const double* P,
                   const double* Q, const double* mu, double rho, double lambda0,
                   const double* theta);
int psmf_zero_gradsum(psmf_handle h);
/* scalars[8] = { rho_k, lambda_k, s, eta, N, phi, omega, step_counter } of the last step. */
int psmf_get_state(psmf_handle h, double* C, double* V, double* P, double* Q, double* mu,
                   double* theta, double* gradsum, double* scalars);
/* Adam moments for the recursive mode (psmf.py:190-204): m, v of length n_theta. */
int psmf_set_adam(psmf_handle h, const double* m, const double* v);

/* Per-step schedules of PSMFIter's R[k], Q[k] (psmf.py:115,123,141): R_k = rho_k[k] I and Q_k = q_k[k] * Q for the 1-based
 * step k = 1 .. n - 1 (entry 0 unused; steps beyond n - 1 are refused by psmf_run).  NULL = constant (rho / Q of
 * psmf_set_state).  Not with robust = 1 (rPSMF runs on its own omega-scaled Q_{k-1}, R_{k-1}, rpsmf.py:123,128,141). */
int psmf_set_schedules(psmf_handle h, const double* rho_k, const double* q_k, int64_t n);
/* PSMFIter's Q[k] when it is NOT a scalar multiple of Q[1] (psmf.py:115 reads a matrix per step): n matrices of r x r doubles,
 * row-major, matrix k for the 1-based step k (matrix 0 unused; steps beyond n - 1 are refused by psmf_run).  P_bar_k =
 * F P F^T + Q_k is then formed from the uploaded matrix in the serial stage of the per-step engine (launched form; the handle
 * must have been created with engine = 1).  NULL or n = 0 drops the schedule.  Not with robust = 1, masked handles or
 * PSMF_DYN_HOST. */
int psmf_set_q_matrix_schedule(psmf_handle h, const double* Q_k, int64_t n);

/* Non-uniform diagonal R (cfg.nonuniform_R = 1): rho_rows[d_local] = diag(R) of this handle's rows, rho_mean = sum of diag(R)
 * over ALL rows / d (tr(R) / d of psmf.py:121-125; the same value on every shard).  The `rho` of psmf_set_state is then the
 * scalar in front (1 to start with; rPSMF multiplies it by omega_k, rpsmf.py:169). */
int psmf_set_row_noise(psmf_handle h, const double* rho_rows, double rho_mean);

/* A NON-DIAGONAL R (the dense d x d branch of psmf.py:150-152, rpsmf.py:150-152), cfg.nonuniform_R = 1, one shard (d_local = d):
 * R = U diag(lam) U^T with U (d x d, row-major, eigenvectors in its COLUMNS, orthonormal) and lam[d] >= 0 from the caller's
 * symmetric eigen-solver (LAPACK dsyevd / numpy.linalg.eigh; O(d^3), once).  The recursion is equivariant under the orthogonal
 * change of observation coordinates y -> U^T y, C -> U^T C (eta and the residual norms are invariant), so the handle keeps the
 * series and the dictionary ROTATED, runs the non-uniform-diagonal step with rho_rows = lam, and rotates at its boundary:
 * psmf_set_state / psmf_upload_series on the way in, psmf_get_state's C, psmf_download_y_pred, psmf_predict, psmf_project on the
 * way out (float64 GEMMs against the resident U on the matrix cores) -- callers see original coordinates everywhere; no d x d
 * inverse is formed and a step stays O(d r^2).  Call it before the first psmf_set_state / psmf_upload_series.  U stays resident:
 * 8 d^2 bytes, d <= PSMF_ROTATION_DMAX (PSMF_ERR_ARG beyond).  Orthonormality of U is checked over the whole matrix in O(d^2)
 * (U^T U z = z for two sign vectors z): PSMF_ERR_ARG if it fails. */
int psmf_set_noise_rotation(psmf_handle h, const double* U, const double* lam);

/* ---- series ---------------------------------------------------------------------------- */
/* Y: nt x d_local time-major block holding y_{t0+1} .. y_{t0+nt}; T_total sizes the device
 * buffers on first use.  replaces the dict y[k] of (d,1) arrays passed to step (psmf.py:85-88) */
int psmf_upload_series(psmf_handle h, const void* Y, int dtype, int64_t t0, int64_t nt,
                       int64_t T_total);

/* masked = 1: M: nt x d_local uint8, time-major like Y, 1 = observed, for the steps t0+1 .. t0+nt (after psmf_upload_series, which
 * sizes the buffer).  Where M = 0 the value of Y is never read.  replaces Mk = diag(M[:, t]) of ExperimentImpute/PSMF.py:62. */
int psmf_upload_mask(psmf_handle h, const uint8_t* M, int64_t t0, int64_t nt);

/* ---- the hot loop ---------------------------------------------------------------------- */
/* for k in k_begin+1 .. k_end: inner(k, y_k)  entirely on the device
 * replaces PSMFIter.step / inner and its ten hooks (psmf.py:85-180), rPSMFIter overrides
 * (rpsmf.py:116-184), PSMFRecursive.inner (psmf.py:287-304). Asynchronous. */
int psmf_run(psmf_handle h, int64_t k_begin, int64_t k_end);
int psmf_sync(psmf_handle h); /* waits, then reports a device-side numeric failure if any */

/* Host-stepped dynamics (dyn_kind = PSMF_DYN_HOST): ONE timestep, k -> k + 1 (k = number of finished steps), with
 * mu_bar = f(theta, mu_k, k + 1) [r] and P_bar = F P_k F^T + Q [r x r] formed by the caller (psmf.py:104-115); the device
 * does everything d-sized of inner() (psmf.py:90-102).  Returns what the caller needs for the next step and for the
 * theta gradient: mu_{k+1} [r], g_f = d(incremental likelihood)/df [r] (gradsum += J_theta^T g_f, psmf.py:167-177),
 * P_{k+1}, Q_{k+1} [r x r] (rPSMF scales Q by omega).  Any output may be NULL.  Synchronous. */
int psmf_step_host(psmf_handle h, int64_t k, const double* mu_bar, const double* P_bar, double* mu_out, double* gf_out,
                   double* P_out, double* Q_out);
/* out[q] = C mu[q] for n given r-vectors (n x d_local float64): the y_hat of a roll-out the caller computed (psmf.py:182-188). */
int psmf_project(psmf_handle h, const double* mu, int64_t n, double* out);

/* y_hat for steps t0+1..t0+nt (needs store_y_pred) -> out (nt x d_local), dtype f32/f64 */
int psmf_download_y_pred(psmf_handle h, void* out, int dtype, int64_t t0, int64_t nt);
/* posterior means mu_{k0} .. mu_{k0+nk-1} of the last run (row k = state after step k; the row of the
 * run's first step index holds the mean it started from) -> out (nk x r float64).  What TrackingMixin
 * reads as `_mu[k]` (tracking.py:140-142) when the experiment keeps `_mu` un-pruned. */
int psmf_download_mu(psmf_handle h, double* out, int64_t k0, int64_t nk);
/* psmf.py:182-188: mu rolled forward n_pred steps from the current state, y_hat = C mu_pred;
 * out: n_pred x d_local float64.  T = index of the last filtered step. */
int psmf_predict(psmf_handle h, int64_t T, int64_t n_pred, double* out);
/* sum_t sum_i (y_hat - y)^2 over steps t0+1..t0+nt of the uploaded series (local rows);
 * tracking.py:63-76 error norms without copying y_pred back. */
int psmf_sq_error(psmf_handle h, int64_t t0, int64_t nt, double* out);
/* the same for the roll-out window: sum over q < n_pred and the local rows of (C mu_pred_q - Y_true[q])^2, Y_true the held-out
 * observations y_{T+1} .. y_{T+n_pred} (n_pred x d_local float64, host): tracking.py:74-76 (`_E_pred`). */
int psmf_predict_sq_error(psmf_handle h, int64_t T, int64_t n_pred, const double* Y_true, double* out);

/* masked = 2 / 3: the step size gam of the stochastic-gradient update of C for the runs that follow (the experiments use
 * 1e-6 / (pass + 1)^0.7, MLESMF.py:59-60, TMF.py:46-48). */
int psmf_set_step_size(psmf_handle h, double gam);
/* masked != 0, after psmf_run over the steps t0+1 .. t0+nt: the evaluation sums of ExperimentImpute over the held-out entries
 * Mmiss (nt x d_local uint8, 1 = artificially removed) of this handle's rows, reduced on the device:
 *   out4[0] = sum (y_hat - y)^2          RMSEM(Yrec, YorgInt, Mmiss)^2 * out4[3]            PSMF.py:88, common.py:79-84
 *   out4[1] = sum (c_i . x_t - y)^2      RMSEM(C @ X, ...) with the present C, x_t = posterior mean of step t   PSMF.py:86-89
 *   out4[2] = entries strictly inside y_hat -+ sig sqrt(N_t) (PSMF.py:83-84) resp. sig sqrt(s_t m_i + eta_t) (rPSMF.py:121-123)
 *   out4[3] = number of held-out entries                                                     common.py:87-94
 * (row-sharded filter: the caller adds the four sums over the shards).  y = the uploaded series (YorgInt). */
int psmf_masked_metrics(psmf_handle h, const uint8_t* Mmiss, int64_t t0, int64_t nt, double sig, double* out4);
/* masked = 1: (s_t, eta_t) of the steps t0+1 .. t0+nt -> out (nt x 2 float64): what the bands YrecL / YrecH are formed from. */
int psmf_download_step_scalars(psmf_handle h, double* out, int64_t t0, int64_t nt);

/* ---- multi-GPU (row shards, one process per GPU, RCCL over xGMI) ------------------------- */
#define PSMF_UNIQUE_ID_BYTES 128
int psmf_comm_unique_id(void* id_out /* PSMF_UNIQUE_ID_BYTES */);
int psmf_comm_init(psmf_handle h, int nranks, int rank, const void* unique_id);
/* Give up the handle's RCCL communicator WITHOUT waiting for its peers (ncclCommAbort; psmf_destroy calls ncclCommDestroy, which
 * may block when a peer never joined): for the failure path of a multi-rank start.  The handle is a single shard again afterwards.
 * (New; the reference is single-process.) */
int psmf_comm_abort(psmf_handle h);

/* Host-mediated communicator instead of RCCL: wherever a sharded engine needs its sum-all-reduce (per-step engine: r + 1
 * float64 per timestep; blocked engine: one 64 x 64 Gram per run and one 128 x 64 cross-Gram per block) the library copies
 * the message to the host and calls fn(ctx, buf, count), which must replace buf[0..count) by its sum over all ranks -- same
 * rank order on every rank, so that the replicated r x r state stays bit-identical -- and return 0.  For transports
 * the caller owns (MPI, gloo) and for exercising the sharded arithmetic with several shards on ONE GPU (tests).  Each call
 * synchronises the stream it sits on: correct, not fast.  (New: the reference is single-process, pypsmf/psmf/psmf.py:85-88.) */
typedef int (*psmf_allreduce_fn)(void* ctx, double* buf, int64_t count);
int psmf_comm_init_host(psmf_handle h, int nranks, int rank, psmf_allreduce_fn fn, void* ctx);

/* What the handle's communicator is: out4[0] = 0 none, 1 RCCL, 2 host-mediated; out4[1] = number of ranks, out4[2] = this rank,
 * out4[3] = HIP device -- for an RCCL communicator as RCCL itself reports them (ncclCommCount, ncclCommUserRank, ncclCommCuDevice),
 * so that a benchmark line can state how many ranks the exchange really spanned.  (New; the reference is single-process.) */
int psmf_comm_info(psmf_handle h, int32_t* out4);
/* PCI bus id ("0000:c1:00.0") of HIP device `device` into buf (len >= 16): which physical GPU a rank drives. */
int psmf_device_pci_bus_id(int device, char* buf, int len);

/* ---- measurement ------------------------------------------------------------------------ */
/* psmf_run bracketed by HIP events on the handle's stream; *ms = elapsed milliseconds. */
int psmf_run_timed(psmf_handle h, int64_t k_begin, int64_t k_end, float* ms);
/* average duration (microseconds) of `iters` back-to-back launches of one kernel of the step
 * on the handle's stream, HIP-event timed: which = 0 row sweep (+ concurrent r x r solve),
 * 1 = serial r x r stage.  Blocked engine: 0 = coefficient-space filter of one full block,
 * 1 = cross-Gram of the next block (+ reduction), 2 = apply.  State is saved and restored around the measurement. */
int psmf_time_kernel(psmf_handle h, int which, int iters, float* avg_us);
/* geometry actually used: out[0] = sweep workgroups, out[1] = rows per workgroup,
 * out[2] = padded row length (elements), out[3] = lanes per row, out[4] = graph chunk steps,
 * out[5] = engine in use (1 per-step, 2 blocked), out[6] = steps per block (blocked engine) */
int psmf_geometry(psmf_handle h, int32_t* out7);
/* Which kernel advances the r x r / coefficient-space state with the handle's present configuration (mode flags, dynamics, the Q
 * last uploaded, schedules, switches): 0 = per-step engine (psmf_sweep_solve + psmf_serial), 1 = psmf_blk_filter (general blocked
 * kernel), 2 = psmf_blk_filter2, 3 = psmf_blk_filter3, 4 = psmf_blk_filter3s, 5 = psmf_blk_filter4, 6 = psmf_blk_filter4s, 7 = psmf_blk_filter5,
 * 8 = psmf_blk_filter6 (every configuration at r <= 16 but the simplified hooks), 9 = psmf_blk_filter6d (its instantiation with the two
 * inversions side by side: random walk, Q = q I), 10 = psmf_blk_filter7 (the same design on 2 x 2 tiles: what is left at 17 <= r <= 32),
 * 11 = psmf_pstep_k, the per-step engine as ONE persistent launch per run (C on chip, device-flag hand-offs; psmf_pstep.hip).
 * The same function decides what is launched (select_filter_kernel, psmf_capi.hip).
 * (New: diagnostics for tests and bench.py -- the reference has one code path, pypsmf/psmf/psmf.py:90-102.) */
int psmf_filter_kernel(psmf_handle h);
/* diagnostics of the blocked engine's r x r inversions since the last reset: out[0] = timesteps inverted by
 * Newton-Schulz refinement, out[1] = by the direct symmetric sweep, out[2] = Newton-Schulz iterations in
 * total, out[3] = failed Newton-Schulz attempts, out[4] / out[5] = summed in-kernel durations / gaps between
 * consecutive filter kernels in 10 ns ticks, out[7] = filter launches; reset != 0 clears them.  (New: the reference has no
 * counterpart; its np.linalg.inv calls are pypsmf/psmf/psmf.py:147-149.) */
int psmf_counters(psmf_handle h, int64_t* out8, int reset);
/* Blocked engine with chained blocks (one launch of the coefficient-space filter kernel per psmf_run): number of such
 * launches completed since the last reset and the sum of their durations, measured with HIP events recorded around each
 * launch on the stream it runs on (at most 256 runs between two psmf_sync calls are timed).  Waits like psmf_sync.
 * (New: measurement aid for bench.py; the reference times whole runs with time.time(), ExperimentImpute/PSMF.py:59,91.) */
int psmf_filter_kernel_time(psmf_handle h, int64_t* launches, double* total_ms, int reset);

/* Measured HBM bandwidth of a plain streaming copy (`bytes` read + `bytes` written per launch, `iters` launches, HIP events):
 * the "measured copy-kernel peak" SURVEY 8(d) asks for beside the nominal 8 TB/s.  *gbps = (read + written bytes) / time. */
int psmf_measure_copy_bandwidth(int device, size_t bytes, int iters, double* gbps);

/* ================= masked, batched small-d filter (ExperimentImpute) ==================== */
typedef struct {
  int32_t abi_version;
  int32_t d, n, r;       /* Y is d x n, C d x r, X r x n (reference layout)                  */
  int32_t batch;         /* independent replicas (seeds): own mask, C0, X0                    */
  int32_t method;        /* 0: PSMF.py:40-95   1: rPSMF.py:40-148   2: MLE-SMF, MLESMF.py:40-92 (V unused)
                          * 3: TMF, TMF.py:30-73 (V, P, Q, rho unused; nu = 2; no bands, inside = 0)        */
  int32_t n_iter;        /* passes over the n columns (Iter)                                  */
  int32_t device;
  int32_t want_bands;    /* also return Yrec, YrecL, YrecH                                    */
  double sig;            /* band half-width in standard deviations                             */
  double lambda0;
} psmf_impute_config;

/* Any d and 1 <= r <= PSMF_RMAX.  Shapes whose replica state fits one workgroup's LDS (d <= 512, r <= 16) run one workgroup per
 * replica, all replicas in one launch (psmf_impute_kernel3 / psmf_impute_kernel2); larger shapes run the replicas one after the
 * other on the masked per-step engine of the large-d handle (all four methods).  psmf_impute_kernel_id() tells which.
 * All arrays time-major (column t of the reference's d x n matrices is row t here):
 *   YorgInt  n x d  float64  data with native missing values set to 0        (shared)
 *   M        batch x n x d  uint8   1 = observed                              (per replica)
 *   Mmiss    batch x n x d  uint8   1 = artificially removed (evaluation set)
 *   C        batch x d x r  float64 in: C0   out: final C
 *   X        batch x n x r  float64 in: X0   out: final X (the reference mutates X in place)
 *   V, P, Q  r x r float64 (shared initial values);  rho = uniform diag(R)
 *   Epred, Efull  batch x n_iter  (RMSE after each pass, PSMF.py:88-89)
 *   inside        batch           (coverage, common.py:87-94)
 *   Yrec, YrecL, YrecH  batch x n x d float64 or NULL
 *   status        batch int32 or NULL: per-replica outcome, PSMF_OK or PSMF_ERR_NUMERIC (r x r system lost positive definiteness,
 *                 or non-finite errors).  A failed replica gets NaN in Epred / Efull / inside, the others are unaffected and the
 *                 call returns PSMF_OK -- the reference records NaN for a diverged repeat and carries on (rPSMF.py:236-243).
 *                 With status = NULL a failed replica fails the call (PSMF_ERR_NUMERIC).
 * replaces ProbabilisticSequentialMatrixFactorizer / robust_PSMF (and, method 2 / 3, the baseline filters
 * stochasticGradientStateSpaceMF / temporalRegularizedMF that share their masked contractions) and the RMSEM /
 * compute_number_inside_bars calls made on their outputs. */
int psmf_impute_run(const psmf_impute_config* cfg, const double* YorgInt, const uint8_t* M,
                    const uint8_t* Mmiss, double* C, double* X, const double* V,
                    const double* P, const double* Q, double rho, double* Epred, double* Efull,
                    double* inside, double* Yrec, double* YrecL, double* YrecH, int32_t* status, float* elapsed_ms);
/* Which column loop psmf_impute_run uses for cfg->d, cfg->r (with the present environment switches): 1 = round 1's loop,
 * 2 = psmf_impute_kernel2, 300 + NG = psmf_impute_kernel3<NG> (d <= 80, r <= 14; NG = 4-row groups: 3, 5, 8, 12, 20),
 * 4 = masked per-step engine of the large-d handle.  Negative = error.  (Diagnostics; the reference has one code path.) */
int psmf_impute_kernel_id(const psmf_impute_config* cfg);

#ifdef __cplusplus
}
#endif
#endif /* PSMF_HIP_H */
