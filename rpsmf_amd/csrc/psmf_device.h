// Device-side data structures shared by the kernels and the C-ABI host code.
// gfx950 only: wave = 64 lanes, workgroups of 256 threads (4 waves, one per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psmf {

constexpr int RM = 64;    // PSMF_RMAX
constexpr int WG = 256;   // threads per workgroup everywhere

// All O(r^2) state lives here, in float64, replicated on every GPU of a sharded filter.
// r x r matrices are stored compactly (row stride r); all are symmetric.
struct DevState {
  double V[RM * RM];      // dictionary column covariance V_{k-1}
  double P[RM * RM];      // coefficient covariance P_{k-1}
  double Pplus[RM * RM];  // (I + kappa Pbar G)^-1 Pbar for the CURRENT step (written by the solve block)
  double Pbar[RM * RM];   // predictive covariance of the current step
  double Q[RM * RM];      // running Q (scaled by omega in rPSMF)
  double G[RM * RM];      // Gram matrix C^T C of the current C (tracked algebraically)
  double GR[RM * RM];     // non-uniform diagonal R: weighted Gram sum_i c_i c_i^T / (rho_i + s) of the CURRENT step (psmf_gram_partial)
  // carried across the blocks of the blocked engine (valid while ns_valid != 0):
  double Lbar[RM * RM];       // r x r: Pbar^-1 of the next step (blocked engine: r <= 32; per-step engine with solve_dual: any r <= RM)
  double XpX[RM * RM / 4];    // last P+ (Newton-Schulz start of half X; blocked engine only: r <= 32)
  double XpY[RM * RM];        // last W  (Newton-Schulz start of half Y; per-step engine with solve_dual: W of the step, any r <= RM)
  double mu[RM];          // posterior mean mu_{k-1}
  double mu_bar[RM];      // predictive mean of the current step
  double w[RM];           // V mu_bar
  double wN[RM];          // w / N   (rank-1 update direction used by the row sweep)
  double gf[RM];          // d(incremental likelihood)/d f of the last finished step (host-stepped dynamics: the host forms J_theta^T gf)
  double red[2 * (RM + 1) + 6];   // h[0..r), ee at [r] (+ b, q of a non-uniform R behind them): all-reduced partial sums (multi-GPU path)
  double rho, lam;        // running diag(R) (uniform) and Student-t dof
  double s, eta, N, kappa;  // scalars of the current step
  double phi, omega, ee;  // scalars of the last finished step
  double s_done, eta_done, N_done;  // s, eta, N of the last finished step
  double sgd_gamma;       // masked_method 2 / 3: the pass's step size gam = 1e-6 / (pass + 1)^0.7 (MLESMF.py:59-60, TMF.py:46-48), set by the host
  long long kq;           // masked handle: series index of the step whose masked Gram the next psmf_serial_mgram launch computes (= k + 1 during a run)
  long long k;            // number of finished steps = 0-based series index of the current step
  int err;                // != 0: numeric failure (singular system) at step err
  int ns_valid;           // != 0: Lbar / XpX / XpY describe the current state; 3: so does the f3_* dump (cleared by every host state upload);
                          // 7: per-step engine with the inversions side by side (StepParams.solve_dual): Lbar = Pbar^-1 of the NEXT step, XpY = W
  // filter3 (psmf_blk3.hip): the r x r state between the blocks of a run, exactly as the waves hold it in registers
  // ([register][lane]: coalesced 512-byte rows), valid while ns_valid == 3.  The row-major V / P / G / Lbar / XpX / XpY
  // above are written by the LAST block of a run only (BlockParams.last).
  double f3_G[16 * 64], f3_W[16 * 64], f3_Xc[4][8 * 64], f3_V[16 * 64];
  float f3_Xa[4][16 * 64];
  double f3_sc[8];        // [0] 1 / q that f3_W was formed with, [1] 1 / omega and [2] beta omega of the last step
  long long dbg[8];       // filter3 in-situ breakdown (10 ns ticks): [0] hand-off, [1] K load / assembly, [2] state init, [3] step loop, [4] block end
  long long cnt[8];       // diagnostics of the blocked filter: [0] steps inverted by Newton-Schulz, [1] by the sweep,
                          // [2] Newton-Schulz iterations in total, [3] failed Newton-Schulz attempts (psmf_counters)
  unsigned ticket;        // row workgroups of the current sweep that have stored their partial row (StepParams.tail_reduce); the last one resets it
  unsigned ticket_pad;
};

struct StepParams {
  DevState* st;
  void* C;            // d_local x rp, storage type
  const void* Y;      // series buffer, time-major, storage type
  void* YP;           // y_pred buffer or nullptr
  double* partials;   // n_sweep_wg x ps
  double* mu_hist;    // (T_cap + 1) x r posterior means, row k = mu_k (row k_begin = the mean the run started from); or nullptr
  long long series_t0;  // global index of the first step held in Y
  int d, d_local, r, rp, nv;
  int n_sweep_wg, rows_per_wg, ps;
  int robust, coef_update, eta_full, pbar_predict, fixed_lambda;
  int dyn_kind, n_theta, store_yp, recursive, update_every, track_g;
  // parameters of the dynamics, their summed gradient and Adam moments: n_theta doubles each, global memory (psmf_dyn.hip)
  double* theta;
  double* gradsum;
  double* adam_m;
  double* adam_v;
  int dyn_flags;        // SCALED_WALK: bit 0 = bias; SINUSOID: bit 0 = scaled (matrix A), bit 1 = phased (gains c)
  int dyn_terms;        // FOURIER: N
  // per-step schedules of PSMFIter's Q[k], R[k] (psmf.py:115,123,141): R_k = rho_sched[k] I, Q_k = q_sched[k] * Q; index = 1-based
  // step; nullptr = constant
  const double* rho_sched;
  const double* q_sched;
  // ... or Q_k as a matrix of its own per step (psmf_set_q_matrix_schedule: r x r doubles per step, same indexing; per-step engine,
  // launched form); nullptr = Q of the state, times q_sched
  const double* q_mat;
  // non-uniform diagonal R (per-step engine): R = st->rho * diag(rho_rows), st->rho starting at 1 (rPSMF scales it by omega);
  // rho_mean = sum(rho_rows over ALL shards) / d, so that tr(R) / d = st->rho * rho_mean.  nullptr / 1.0: uniform R = st->rho I
  const double* rho_rows;
  double rho_mean;
  // masked filter (cfg.masked, psmf_masked.hip): T_cap x d_local observation mask, time-major like Y (1 = observed); nullptr = all observed
  const uint8_t* mask;
  const double* mg;     // masked: r*r + 1 doubles -- masked Gram G_m and observed count of the CURRENT step, summed over workgroups and ranks
  const double* mg_tr;  // masked: partial sums of <G_m, Pbar> (mg_ntr of them, one per workgroup of the Gram's reduction; summed in order)
  int mg_ntr;
  double* sc_hist;      // masked: T_cap x 2, (s, eta) of every step (the bands of the pass metrics are formed from them)
  int mask_rows;        // masked: rows of the mask buffer (T_cap)
  // per-step engine, random walk with Q = q I (r <= 32): the two r x r inversions of a step side by side on two waves
  // of the solve block -- P+ = M^-1, M = Lbar + kappa G, and W = (M / beta + I / q)^-1, from which the serial stage forms
  // Lbar' = Pbar'^-1 = (I / q - W / q^2) / omega for the next step (the Woodbury form of filter3): ONE sweep on
  // the path of a step instead of two.  DevState.Lbar / XpY carry Lbar / W while ns_valid == 7.
  int solve_dual;
  // masked handle: which filter of ExperimentImpute the masked steps are.  0: PSMF / rPSMF (PSMF.py:59-84, rPSMF.py:75-135).
  // 2: MLE-SMF (MLESMF.py:57-88): weights m_i / rho (no s), C += gam / eta (m o e) x_p^T, V unused, bands -+ sig sqrt(eta).
  // 3: TMF (TMF.py:47-66): x = x_p + (nu I + G_m)^-1 C^T e, i.e. the same solve with Pbar = I / nu, kappa = 1; C += gam (m o e) x_p^T.
  int masked_method;
  int solve_lds;        // 1: the LDS-and-barrier sweeps of round 1 for every r (PSMF_STEP_WAVE_SOLVE=0); default: wave-local sweeps for r <= 32
  int external_reduce;  // 1: partial sums were reduced into st->red (multi-GPU; tail_reduce)
  int tail_reduce;      // 1: the last row workgroup of a sweep sums the partial rows into st->red (psmf_kernels.hip, tail_reduce_partials)
  int use_ns;           // 1: Newton-Schulz refinement of the r x r inverses (f64 MFMA), sweep as fallback
  int ns_predict;       // 1: filter3 starts the iteration from the rank-2 downdated, kappa-rescaled previous inverse
  int ns_skip_n;        // filter3: timesteps that go straight to the direct sweep after a failed Newton-Schulz start
  double alpha, beta, lr, lr_end, lr_steps, b1, b2;
  double ns_tol2;       // Newton-Schulz: squared Frobenius residual accepted BEFORE the last update (the update squares it)
  double ns_far2;       // Newton-Schulz: squared residual beyond which the start is given up for the direct sweep (filter3)
};

// sum over the 64 lanes of a wave (butterfly; same value in every lane)
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);   // v_rcp_f64, ~1e-8 relative
  x = x * (2.0 - d * x);
  x = x * (2.0 - d * x);
  return x;
}

}  // namespace psmf
