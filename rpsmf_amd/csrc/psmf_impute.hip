// Masked, batched small-d engine: the filter loops of ExperimentImpute
//   ProbabilisticSequentialMatrixFactorizer   ExperimentImpute/PSMF.py:40-95
//   robust_PSMF                               ExperimentImpute/rPSMF.py:40-148
// and the two baseline filters of the imputation tables that share their masked contractions
//   stochasticGradientStateSpaceMF (MLE-SMF)  ExperimentImpute/MLESMF.py:40-92   weights m_i / rho_i, C += gam / eta (m o e) x_p^T
//   temporalRegularizedMF (TMF)               ExperimentImpute/TMF.py:30-73      x_t = x_p + (nu I + G)^-1 C^T e, C += gam (m o e) x_p^T
// plus the RMSEM / compute_number_inside_bars reductions made on their outputs
// (ExperimentImpute/common.py:79-94), for a whole batch of independent replicas (seeds).
//
// One 256-thread workgroup per replica; the replica's entire state (C d x r, V, P, Q, x) lives
// in LDS / registers in float64 and the kernel runs all n_iter * n columns without returning to
// the host.  The observation weights change every column (mask), so the r x r Gram
// sum_i m_i c_i c_i^T is recomputed per column (no algebraic tracking); d <= a few hundred here.
// Per column: masked residual, augmented Gram [C | e]^T diag(m) [C | e] (gives G, b = C^T e, e^T e in
// one pass), the two symmetric sweep inversions of the reference's Woodbury form
// (PSMF.py:30-36), the Kalman update of x, the rank-1 updates of C and V, error bands.
// Everything is latency-bound; inputs of column t+1 are prefetched while column t is processed, and the barriers inside the
// column loop order LDS only (solve_barrier<true>): the prefetch and the per-column stores (X, bands) stay in flight across them.
#include "../../include/psmf_hip.h"
#include "psmf_kernels.hip"
#include "psmf_blk3.hip"      // DPP row sums, readlane helpers
#include "psmf_wave16.hip"    // wave_sweep16m: the single-wave sweep with the lane predicates as multipliers (round 3)

#include <cmath>
#include <cstring>
#include <string>

namespace psmf {

constexpr int IR = 16;   // largest rank of the masked engine (experiments use r = 10)

// per-phase cycle accumulation for tools/impute_prof.hip (PSMF_IMPUTE_STAMPS); no-ops in the product
#ifdef PSMF_IMPUTE_STAMPS
#define IMP_T0() unsigned long long it_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, il_, in_; { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(il_) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define IMP_T(n) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(in_) :: "memory"); __builtin_amdgcn_sched_barrier(0); it_[n] += in_ - il_; il_ = in_; }
#define IMP_TOUT() if ((threadIdx.x & 63) == 0 && p.prof) for (int q_ = 0; q_ < 8; ++q_) p.prof[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + q_] = it_[q_];
#else
#define IMP_T0()
#define IMP_T(n)
#define IMP_TOUT()
#endif

struct ImputeParams {
  int d, n, r, n_iter, robust, want_bands;
  int method;              // 0 PSMF, 1 rPSMF (robust = 1), 2 MLE-SMF, 3 TMF
  double sig, lambda0, rho0;
  const double* Yorg;      // n x d (shared)
  const uint8_t* M;        // batch x n x d
  const uint8_t* Mmiss;    // batch x n x d
  double* C;               // batch x d x r
  double* X;               // batch x n x r
  const double* V0;
  const double* P0;
  const double* Q0;
  double* Epred;           // batch x n_iter
  double* Efull;           // batch x n_iter
  double* inside;          // batch
  double* Yrec;            // batch x n x d or null
  double* YrecL;
  double* YrecH;
  int* err;                // batch
  int q_iso;               // Q0 = q I with q > 0: the two r x r inversions of a column run in parallel on two waves
  unsigned long long* prof;   // diagnostics (tools/impute_prof.hip) or nullptr
};

// End of a pass: sum over the held-out entries of (C x_t - y_t)^2 with the pass's final C (ExperimentImpute/PSMF.py:86-89, the RMSE of C @ X).
// One COLUMN per thread: 256 independent chains of loads in flight.  (A row per thread leaves d threads walking the n
// columns one memory latency at a time -- 0.75 us per column, an eighth of the whole run at the ExperimentImpute shape.)
// X was written by other threads of this workgroup: read around this CU's L1.  sC: the dictionary in LDS, row stride ldc.
__device__ __forceinline__ double held_out_sse(const double* sC, const int ldc, const double* Xg, const double* Yorg,
                                               const uint8_t* Mm, const int d, const int n, const int r, const int tid) {
  double sse = 0.0;
  for (int t = tid; t < n; t += WG) {
    double xr[IR];
#pragma unroll
    for (int l = 0; l < IR; ++l) {           // unconditional loads (index clamped, value masked): all r in flight at once
      const double v = __builtin_nontemporal_load(&Xg[(size_t)t * r + min(l, r - 1)]);
      xr[l] = l < r ? v : 0.0;
    }
    const size_t base = (size_t)t * d;
#pragma unroll 4
    for (int i = 0; i < d; ++i) {
      const uint8_t m = Mm[base + i];
      const double y = Yorg[base + i];
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int l = 0; l < IR; l += 2) {
        d0 = fma(sC[i * ldc + min(l, r - 1)], xr[l], d0);
        d1 = fma(sC[i * ldc + min(l + 1, r - 1)], xr[l + 1], d1);
      }
      const double dl = (d0 + d1) - y;
      sse += m ? dl * dl : 0.0;
    }
  }
  return sse;
}



// ------------------------------------------------------------------------------------------------------------
// Version 2 of the column loop (the default): FOUR workgroup barriers per column instead of ~15.
//   P1  row owners (waves 1.., so that wave 0 stays free): masked residual rows; wave 3: w = V x          | barrier 1
//   P2  waves 1-3: augmented masked Gram [C | e]^T diag(m) [C | e] on the float64 matrix cores, every wave its share
//       of the 4-row groups (v_mfma_f64_16x16x4_f64; r = 16: a second tile for C^T e); wave 0: sum(m), sum(e^2), s,
//       P + Q, kappa                                                                                       | barrier 2
//   P3a wave 0: sums the three partial tiles -- which leaves G in the MFMA output layout (lane = column, 4 rows per
//       lane) -- <G, P + Q>, eta, N, phi                                                                  | barrier 3
//   P3b wave 0 alone, NO barrier: the two symmetric sweep inversions of the reference's Woodbury form (PSMF.py:30-36)
//       with the 16 x 16 matrix in its registers -- the rank-2 update of a pivot round is one float64 MFMA, the pivot
//       block travels by v_readlane (wave_sweep16; since the end of round 3 its multiplier form, wave_sweep16m of
//       psmf_wave16.hip) -- then x_t = x_p + kappa P+ C^T e (four MFMAs), omega, P, Q
//   P4  meanwhile waves 1-3: rank-1 updates of C and V (they need N, phi only); then every row owner: bands, metrics | barrier 4
// P, Q, rho, lambda live in wave 0's registers for the whole run; x is double-buffered in LDS.
// Measured on the config-D shape (d = 19, r = 10): tools/impute_prof.hip.
// ------------------------------------------------------------------------------------------------------------
// Symmetric sweep of the leading r2 x r2 block (r2 even; identity padding beyond r) of the 16 x 16 matrix held by ONE wave
// (lane: column lr = l & 15, rows lk + 4 q, lk = l >> 4):  A <- -A^-1, by 2 x 2 SPD block pivots as sweep_all
// (psmf_kernels.hip) -- half as many dependent rounds as single pivots -- with no LDS memory and no barrier: the rank-2
// update of a pivot round is ONE v_mfma_f64_16x16x4_f64, the pivot block travels by v_readlane.  (A lone wave issues one
// instruction per 4+ cycles, so a round costs what it has instructions: the first version moved the pivot rows and columns
// with 20 lane shuffles and updated with 8 FMAs per lane, 100 instructions per round against 50.)  Symmetric in, symmetric
// out up to round-off (the two halves of a pair are different FMA chains on the matrix cores).
// Rows k, k + 1 of the matrix are the register A[k >> 2] of the lanes lk = k & 3, (k + 1) & 3 -- which is exactly where the
// A operand of the MFMA wants the two columns u, w (by symmetry) in k-slots k & 3, (k + 1) & 3; the B operand is
// -Ki [u; w]^T in the same lanes (pivot columns: +Ki, their C input zeroed), formed from u_j, w_j that one
// v_permlane16_swap pair brings into both rows.  D = keep o A - u t1^T - w t2^T, then the pivot rows are overwritten in place.
__device__ __forceinline__ void wave_sweep16(double (&A)[4], const int r2, const int lk, const int lr, bool& bad) {
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    if (k < r2) {                                  // uniform
      const int b0 = (k & 3) << 4, b1 = b0 + 16, kq = k >> 2;
      const double rk = A[kq];
      const double ka = readlane_f64(rk, b0 | k), kb = readlane_f64(rk, b0 | (k + 1)), ke = readlane_f64(rk, b1 | (k + 1));
      const double det = ka * ke - kb * kb;
      bad |= !(ka > 0.0) | !(det > 0.0);
      const double dinv = fast_rcp(det);
      const double kp = ke * dinv, kq2 = -kb * dinv, ks = ka * dinv;       // Ki = [[kp, kq2], [kq2, ks]]
      // u_j (even row of the pair) and w_j (odd row) in both rows of each pair
      const unsigned lo = __double2loint(rk), hi = __double2hiint(rk);
      const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      const double uj = __hiloint2double(h2[0], l2[0]), wj = __hiloint2double(h2[1], l2[1]);
      const bool c0 = (lr == k), c1 = (lr == k + 1), piv = c0 | c1;
      const bool in_piv = (lk >> 1) == ((k >> 1) & 1);      // the lanes that hold the two pivot rows
      const bool is_u = (lk & 1) == 0;                      // ... row k (else row k + 1)
      const double u1 = c0 ? 1.0 : (c1 ? 0.0 : uj), w1 = c0 ? 0.0 : (c1 ? 1.0 : wj);
      const double cu = is_u ? kp : kq2, cw = is_u ? kq2 : ks;
      const double sv = cu * u1 + cw * w1;                  // t1_j / t2_j; at the pivot columns the entries of Ki
      const double aop = in_piv ? rk : 0.0;
      const double bop = in_piv ? (piv ? sv : -sv) : 0.0;
      const double keep = piv ? 0.0 : 1.0;
      f64x4 acc = {keep * A[0], keep * A[1], keep * A[2], keep * A[3]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) A[q] = acc[q];
      A[kq] = in_piv ? (piv ? -sv : sv) : acc[kq];          // the pivot rows: t1, t2; pivot block: -Ki
    }
  }
}


// WV: the wave index as a compile-time constant -- one column loop per wave, each holding only its own role's registers and code
// (wave 0: the r x r work; wave 1: W beside it, Gram share; waves 2, 3: Gram share, rank-1 updates; wave 3: w = V x).  With the
// wave index as a run-time value every wave carried the union of the roles through the loop (the same change took 7 % off the
// blocked engine's filter kernel and 30 % off the simplified-hooks kernel).
// The column loops of psmf_impute_kernel2 / psmf_impute_kernel3 are FOUR programs, one per wave (impute2_wave<WV>, impute3_wave<WV, NG>),
// that meet at workgroup barriers placed inside role-dependent code: correct only while every program executes the same NUMBER of
// barriers on every path (n_iter, robust, q_iso, method, row-group branches) -- an edit that adds or drops one in a single role would
// deadlock or, worse, pair up the wrong phases silently.  Every barrier of those programs goes through these two wrappers, which count;
// at the end the four counts are compared and a mismatch is reported as a numeric failure of the replica (flag value 9) -- one scalar
// add per barrier.  tests/test_hip_impute_small.py runs every role / method / shape combination through it.
__device__ __forceinline__ void imp_barrier_lds(int& n) { ++n; solve_barrier<true>(); }
__device__ __forceinline__ void imp_barrier_full(int& n) { ++n; __syncthreads(); }
__device__ __forceinline__ void imp_barrier_check(const int n, int* errflag) {
  __shared__ int s_nbar[4];
  if ((threadIdx.x & 63) == 0) s_nbar[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0 && (s_nbar[0] != s_nbar[1] || s_nbar[0] != s_nbar[2] || s_nbar[0] != s_nbar[3])) *errflag = 9;
  __syncthreads();
}

template <int WV>
__device__ __forceinline__ void impute2_wave(const ImputeParams& p) {
  int nbar = 0;          // barriers this wave has executed (imp_barrier_check at the end)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int d = p.d, n = p.n, r = p.r, tid = threadIdx.x, rep = blockIdx.x;
  constexpr int wv = WV;
  const int lane = tid & 63, lk = lane >> 4, lr = lane & 15;
  // ---- LDS carve (doubles) ----
  // Rows of C, V and the r-vectors are padded to IR = 16 entries with ZEROS (and C, e, m to a multiple of 4 rows): every
  // read below is an unconditional 16-wide row -- no index clamps, no selects (with runtime-r indexing this phase was
  // ~800 instructions per column, a quarter of them selects and clamps)
  const int d4 = (d + 3) & ~3;
  double* sC = sm;                            // d4 x IR
  double* sV = sC + (size_t)d4 * IR;          // IR x IR
  double* sx = sV + IR * IR;                  // 2 x IR: prior mean of the current / next column
  double* sw = sx + 2 * IR;                   // V x
  double* se = sw + IR;                       // d4: masked residual
  double* smk = se + d4;                      // d4: mask as 0/1 double
  double* sgp = smk + d4;                     // 3 waves x 2 tiles x 256: Gram partials in MFMA output layout
  double* ssc = sgp + 4 * 2 * 256;            // 8 scalars: 0 s, 1 eta, 2 N, 3 phi, 4 1 / omega_{t-1}, 5 1 / q_{t-1} (the q W was formed with), 6 1 / q_t, 7 kappa
  double* sW = ssc + 8;                       // 256: W_{t-1} = (M_{t-1} + I / q_{t-1})^-1 in MFMA output layout (parallel inversions)
  double* sred = sW + 256;                    // 16: end-of-pass reductions
  int* errflag = reinterpret_cast<int*>(sred + 16);

  const double* Yorg = p.Yorg;
  const uint8_t* Mk = p.M + (size_t)rep * n * d;
  const uint8_t* Mm = p.Mmiss + (size_t)rep * n * d;
  double* Cg = p.C + (size_t)rep * d * r;
  double* Xg = p.X + (size_t)rep * n * r;

  for (int idx = tid; idx < d4 * IR; idx += WG) { const int i = idx >> 4, l = idx & 15; sC[idx] = (i < d && l < r) ? Cg[i * r + l] : 0.0; }
  for (int idx = tid; idx < IR * IR; idx += WG) { const int i = idx >> 4, l = idx & 15; sV[idx] = (i < r && l < r) ? p.V0[i * r + l] : 0.0; }
  if (tid < 2 * IR) sx[tid] = (tid < r) ? Xg[(size_t)(n - 1) * r + tid] : 0.0;   // t = 0 wraps to the last column (PSMF.py:65)
  if (tid < IR) sw[tid] = 0.0;
  for (int idx = tid; idx < d4; idx += WG) { se[idx] = 0.0; smk[idx] = 0.0; }
  if (tid == 0) *errflag = 0;
  const double dd = (double)d, idd = 1.0 / dd;
  const bool sgd = p.method >= 2;     // MLE-SMF / TMF: gradient step on C along x_p, no V
  const bool tmf = p.method == 3;
  const bool one_tile = r < 16;       // the augmented column e fits the 16 x 16 tile
  // Q = q I (every experiment): the two inversions of a column are made independent, as in the blocked engine (psmf_block.hip):
  //   P+_t = M_t^-1,  M_t = Lbar_t + kappa_t G_t                                   (wave 0)
  //   Lbar_{t+1} = (P_t + q_{t+1} I)^-1 = (1 / omega_t) [ I / q_t - W_t / q_t^2 ],  W_t = (M_t + I / q_t)^-1      (wave 1)
  // (Woodbury on P_t + Q_{t+1} = omega_t (M_t^-1 + q_t I)); otherwise wave 0 inverts P + Q and then (P + Q)^-1 + kappa G in turn.
  const bool par = p.q_iso && !tmf;
  const int r2 = r + (r & 1);         // sweep size: even, identity-padded
  // rows of this thread: waves 1, 2, 3 first (wave 0 owns the r x r work), wave 0 only when d > 192
  const int ro = (tid + 192) & 255;
  // wave 0: P, Q in the MFMA output layout (element (lk + 4 q, lr)), rho, lambda
  double Pm[4], Qm[4];
  bool inq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = lk + 4 * q;
    inq[q] = i < r && lr < r;
    const int a = inq[q] ? i * r + lr : 0, b = inq[q] ? lr * r + i : 0;
    Pm[q] = inq[q] ? 0.5 * (p.P0[a] + p.P0[b]) : 0.0;
    Qm[q] = inq[q] ? 0.5 * (p.Q0[a] + p.Q0[b]) : 0.0;
  }
  double rho = p.rho0, lam = p.lambda0;
  double qv = p.Q0[0], iqv = 1.0;     // running q of Q = q I and its reciprocal (parallel inversions)
  Sw16K swk;                          // (waves 0, 1: the lane constants of wave_sweep16m)
  if (WV < 2) sw16k_init(swk, lk, lr);
  bool bad = false;
  unsigned long long nmiss_l = 0;
  int cur = 0;
  imp_barrier_full(nbar);
  IMP_T0();
  for (int it = 0; it < p.n_iter; ++it) {
    const double gam = 1e-6 / pow((double)(it + 1), 0.7);     // MLESMF.py:59-60, TMF.py:46-48
    if (p.robust) {                 // rPSMF.py:77-79: Q, R, lambda restart every pass; V, P, C carry over
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lk + 4 * q;
        const int a = inq[q] ? i * r + lr : 0, b = inq[q] ? lr * r + i : 0;
        Qm[q] = inq[q] ? 0.5 * (p.Q0[a] + p.Q0[b]) : 0.0;
      }
      rho = p.rho0;
      lam = p.lambda0;
      qv = p.Q0[0];
    }
    if (par && (it == 0 || p.robust)) {
      // Lbar_0 = (P + q I)^-1 by one sweep, handed over as the W that reproduces it: W = q I - q^2 Lbar (omega = 1)
      if (wv == 0) {
        double A[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = inq[q] ? Pm[q] + ((lk + 4 * q) == lr ? qv : 0.0) : (((lk + 4 * q) == lr) ? 1.0 : 0.0);
        wave_sweep16m(A, r2, swk, bad);          // -(P + q I)^-1
#pragma unroll
        for (int q = 0; q < 4; ++q) sW[q * 64 + lane] = inq[q] ? ((lk + 4 * q) == lr ? qv : 0.0) + qv * qv * A[q] : 0.0;
        iqv = 1.0 / qv;
        if (lane == 0) { ssc[4] = 1.0; ssc[5] = iqv; ssc[6] = iqv; }
      }
      imp_barrier_full(nbar);
    }
    double sse_pred = 0.0;
    unsigned long long inside_l = 0;
    nmiss_l = 0;
    // prefetch column 0.  The loads are UNCONDITIONAL (row index clamped, value masked where it is used): a load under a
    // runtime predicate is branched around and waited for on the spot -- a full memory latency per column
    double ny[2];
    uint8_t nm[2], nmm[2];
    int rowc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rowc[u] = min(ro + u * WG, d - 1);
      ny[u] = Yorg[rowc[u]];
      nm[u] = Mk[rowc[u]];
      nmm[u] = Mm[rowc[u]];
    }
    for (int t = 0; t < n; ++t) {
      const double* sxc = sx + cur * IR;
      double* sxn = sx + (cur ^ 1) * IR;
      double yv[2];
      uint8_t mv[2], mmv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { yv[u] = ny[u]; mv[u] = nm[u]; mmv[u] = nmm[u]; }
      {
        const size_t cbase = (size_t)min(t + 1, n - 1) * d;      // (the last column is simply loaded twice)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          ny[u] = Yorg[cbase + rowc[u]];
          nm[u] = Mk[cbase + rowc[u]];
          nmm[u] = Mm[cbase + rowc[u]];
        }
      }
      // ---- P1: residual rows (row owners), w = V x (wave 3).  Nothing else: sum(m), sum(e^2), s are formed by the idle wave 0
      //      in P2 from what this phase leaves in LDS ----
      double yh[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = ro + u * WG;
        yh[u] = 0.0;
        if (i < d) {
          // every LDS read of the row is issued before the first use (a loop over the runtime r waits for each pair in turn)
          double cr[IR], xr[IR];
#pragma unroll
          for (int l = 0; l < IR; ++l) { cr[l] = sC[i * IR + l]; xr[l] = sxc[l]; }
          double d0 = 0.0, d1 = 0.0;
#pragma unroll
          for (int l = 0; l < IR; l += 2) { d0 = fma(cr[l], xr[l], d0); d1 = fma(cr[l + 1], xr[l + 1], d1); }
          const double dot = d0 + d1;
          const double mi = mv[u] ? 1.0 : 0.0;
          const double yi = mv[u] ? yv[u] : 0.0;     // Y is 0 where unobserved (PSMF.py:147-148)
          se[i] = mi * (yi - dot);
          smk[i] = mi;
          yh[u] = dot;
        }
      }
      if (wv == 3 && lane < IR) {        // w = V x (rows >= r of V are zero)
        double vr[IR], xr[IR];
#pragma unroll
        for (int l = 0; l < IR; ++l) { vr[l] = sV[lane * IR + l]; xr[l] = sxc[l]; }
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int l = 0; l < IR; l += 2) { a0 = fma(vr[l], xr[l], a0); a1 = fma(vr[l + 1], xr[l + 1], a1); }
        sw[lane] = a0 + a1;
      }
      IMP_T(0);
      imp_barrier_lds(nbar);                                          // ---- barrier 1
      IMP_T(1);
      // ---- P2: augmented masked Gram on the matrix cores, wave w: 4-row groups w, w + 4, ... ----
      double G[4], Bq[4], PP[4], kappa = 0.0, N = 0.0, eta = 0.0, s = 0.0, ee = 0.0, phi = 1.0, msum = 0.0, ild = 0.0;
      double Lb[4] = {0.0, 0.0, 0.0, 0.0}, iqt_w1 = 0.0, kap_w1 = 0.0;
      if (wv == 0) {
        // wave 0 meanwhile: everything of P3a that does not need the Gram
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double qd = par ? ((lk + 4 * q) == lr ? qv : 0.0) : Qm[q];
          PP[q] = inq[q] ? (tmf ? ((lk + 4 * q) == lr ? 0.5 : 0.0) : Pm[q] + qd) : 0.0;     // TMF: P + Q := I / nu, nu = 2 (TMF.py:47,60)
        }
        double ms = 0.0, es = 0.0;
        for (int i = lane; i < d4; i += 64) { const double ev = se[i]; ms += smk[i]; es = fma(ev, ev, es); }   // (padding rows are zero)
        msum = wave_sum_f64_dpp(ms);
        ee = wave_sum_f64_dpp(es);
        s = wave_sum_f64_dpp(lane < IR ? sxc[lane & 15] * sw[lane & 15] : 0.0);                              // s = x^T V x
        if (lane == 0) ssc[0] = s;
        // weights of the observed rows: PSMF / rPSMF 1 / (rho + s) (PSMF.py:71-72), MLE-SMF 1 / rho (MLESMF.py:70), TMF 1
        kappa = tmf ? 1.0 : fast_rcp(sgd ? rho : rho + s);
        ild = fast_rcp(lam + dd);
        if (lane == 0) ssc[7] = kappa;
      } else {
        f64x4 acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        const int ngrp = (d + 3) >> 2;
        for (int g = wv - 1; g < ngrp; g += 3) {
          const int k = 4 * g + lk;                                     // < d4: padding rows hold zeros
          const double cval = sC[k * IR + lr], ek = se[k], mk = smk[k];
          const bool kin = true;
          const double eaug = (one_tile && lr == r) ? ek : 0.0;         // augmented row / column r: e (already masked)
          const double a = mk * cval + eaug;                           // (m is 0 / 1: exact; cval = 0 where eaug != 0)
          const double b1 = cval + eaug;
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
          if (!one_tile) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (lr == 0 && kin) ? ek : 0.0, acc2, 0, 0, 0);
        }
        double* o = sgp + (size_t)(wv - 1) * 512;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q * 64 + lane] = acc1[q];
        if (!one_tile) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[256 + q * 64 + lane] = acc2[q];
        }
      }
      IMP_T(2);
      imp_barrier_lds(nbar);                                          // ---- barrier 2
      IMP_T(3);
      // ---- P3a (wave 0; wave 1 too when it inverts beside it): G, b, Lbar, <G, P + Q>, eta, N, phi ----
      if (wv == 0 || (par && wv == 1)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double g0 = (sgp[q * 64 + lane] + sgp[512 + q * 64 + lane]) + sgp[1024 + q * 64 + lane];
          double g1 = 0.0;
          if (!one_tile) g1 = (sgp[256 + q * 64 + lane] + sgp[768 + q * 64 + lane]) + sgp[1280 + q * 64 + lane];
          G[q] = inq[q] ? g0 : 0.0;
          Bq[q] = one_tile ? g0 : g1;      // b_i = (C^T e)_i sits in column r (one tile) / column 0 (second tile) of rows i
        }
        if (par) {
          const double iom = ssc[4], iq = ssc[5];
          iqt_w1 = ssc[6];                 // read HERE, before barrier 3: wave 0 rewrites these slots at the end of its P3b
          kap_w1 = ssc[7];
          const double c1 = iom * iq, c2 = c1 * iq;
#pragma unroll
          for (int q = 0; q < 4; ++q) Lb[q] = inq[q] ? ((lk + 4 * q) == lr ? c1 : 0.0) - c2 * sW[q * 64 + lane] : 0.0;
        }
      }
      if (wv == 0) {
        double tr = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) tr += G[q] * PP[q];
        const double trGP = wave_sum_f64_dpp(tr);
        eta = (rho * msum + trGP) * idd;     // divide by d, not by #observed (PSMF.py:77)
        N = s + eta;
        if (p.robust) phi = (lam + ee * fast_rcp(N)) * ild;           // rPSMF.py:112-114 (e = 0 on unobserved rows)
        if (lane == 0) { ssc[1] = eta; ssc[2] = N; ssc[3] = phi; }
      }
      IMP_T(4);
      imp_barrier_lds(nbar);                                          // ---- barrier 3
      IMP_T(5);
      if (par && wv == 1) {
        // ---- P3b, wave 1: W_t = (M_t + I / q_t)^-1 for the next column's Lbar, beside wave 0's inversion of M_t ----
        const double kap = kap_w1, iqt = iqt_w1;
        double A[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = inq[q] ? Lb[q] + kap * G[q] + ((lk + 4 * q) == lr ? iqt : 0.0) : (((lk + 4 * q) == lr) ? 1.0 : 0.0);
        wave_sweep16m(A, r2, swk, bad);
#pragma unroll
        for (int q = 0; q < 4; ++q) sW[q * 64 + lane] = inq[q] ? -A[q] : 0.0;      // read after barrier 4 + barrier 2 of the next column
      } else if (wv == 0) {
        // ---- P3b: P+ = ((P + Q)^-1 + kappa G)^-1, x_t, omega, P, Q: wave 0 alone, no barrier ----
        double A[4];
        if (par) {
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = inq[q] ? Lb[q] + kappa * G[q] : (((lk + 4 * q) == lr) ? 1.0 : 0.0);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = inq[q] ? PP[q] : (((lk + 4 * q) == lr) ? 1.0 : 0.0);
          wave_sweep16m(A, r2, swk, bad);                // -(P + Q)^-1
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = inq[q] ? kappa * G[q] - A[q] : (((lk + 4 * q) == lr) ? 1.0 : 0.0);
        }
        wave_sweep16m(A, r2, swk, bad);                // -P+
        // z = P+ b on the matrix cores: A[q] (symmetric) is the A operand of k-block q as it stands; b_i sits in Bq[q] of the
        // lanes lr == lb (column r of the augmented tile / column 0 of the second one): as the B operand it makes column lb
        // of the product z -- no shuffle of b to the columns, no row sums (4 x 12 DPP instructions)
        const int lb = one_tile ? r : 0;
        f64x4 zacc0 = {0.0, 0.0, 0.0, 0.0}, zacc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double bop = (lr == lb && (lk + 4 * q) < r) ? Bq[q] : 0.0;
          const double aop = inq[q] ? A[q] : 0.0;
          if (q & 1) zacc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, zacc1, 0, 0, 0);
          else zacc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, zacc0, 0, 0, 0);
        }
        double z[4], part = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          z[q] = -(zacc0[q] + zacc1[q]);                              // (P+ C^T e)_i, i = lk + 4 q, on the lanes lr == lb
          part += ((lk + 4 * q) < r) ? Bq[q] * z[q] : 0.0;            // b_i z_i on the lanes that hold b_i (lr == lb)
        }
        const double bPb = (readlane_f64(part, lb) + readlane_f64(part, 16 + lb)) + (readlane_f64(part, 32 + lb) + readlane_f64(part, 48 + lb));
        if (lr == lb) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = lk + 4 * q;
            if (i < r) {
              const double xn = sxc[i] + kappa * z[q];
              sxn[i] = xn;                                // (entries >= r of both buffers stay zero)
              Xg[(size_t)t * r + i] = xn;                 // the reference overwrites X[:, t] in place
            }
          }
        }
        double omega = 1.0;
        if (p.robust) omega = (lam + kappa * ee - kappa * kappa * bPb) * ild;   // rPSMF.py:105
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Pm[q] = inq[q] ? omega * -A[q] : 0.0;
          if (p.robust) Qm[q] *= omega;
        }
        if (par) {          // for the next column: 1 / omega_t, 1 / q_t (the q W_t is formed with), 1 / q_{t+1}
          const double iom = p.robust ? fast_rcp(omega) : 1.0;
          if (lane == 0) { ssc[4] = iom; ssc[5] = iqv; ssc[6] = iqv * iom; }
          iqv *= iom;
        }
        if (p.robust) { rho *= omega; lam += dd; qv *= omega; }
      } else {
        // ---- P4a (the other waves): rank-1 updates of C and V with N, phi of this column ----
        const double Nn = ssc[2], ph = ssc[3], et = ssc[1];
        const double wsc = fast_rcp(Nn);
        const double csc = tmf ? gam : gam * fast_rcp(et);        // MLESMF.py:79, TMF.py:63
        const int t0 = par ? 128 : 64;          // first thread of the updating waves
        for (int idx = tid - t0; idx < d * IR; idx += WG - t0) {       // (padding columns: x, w are zero there)
          const int i = idx >> 4, l = idx & 15;
          sC[idx] += sgd ? se[i] * sxc[l] * csc : se[i] * sw[l] * wsc;
        }
        if (!sgd)
          for (int idx = tid - t0; idx < r * IR; idx += WG - t0) {
            const int i = idx >> 4, c = idx & 15;
            sV[idx] = ph * (sV[idx] - sw[i] * sw[c] * wsc);
          }
      }
      // ---- P4b: bands, metrics of the rows this thread owns ----
      {
        const double ss = ssc[0], et = ssc[1], Nn = ssc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = ro + u * WG;
          if (i < d) {
            const double band = p.sig * sqrt(p.robust ? (ss * (mv[u] ? 1.0 : 0.0) + et) : (sgd ? et : Nn));   // rPSMF.py:121-123 / PSMF.py:83-84 / MLESMF.py:81-82
            const double lo = yh[u] - band, hi = yh[u] + band;
            if (mmv[u]) {
              const double dl = yh[u] - yv[u];
              sse_pred += dl * dl;
              nmiss_l += 1;
              if (it == p.n_iter - 1 && !tmf && yv[u] < hi && lo < yv[u]) inside_l += 1;
            }
            if (p.want_bands) {
              const size_t off = ((size_t)rep * n + t) * d + i;
              p.Yrec[off] = yh[u];
              p.YrecL[off] = lo;
              p.YrecH[off] = hi;
            }
          }
        }
      }
      cur ^= 1;
      IMP_T(6);
      imp_barrier_lds(nbar);                                          // ---- barrier 4
      IMP_T(7);
    }
    // ---- end of pass: RMSE of the one-step predictions, RMSE of C @ X, coverage ----
    imp_barrier_full(nbar);                 // (drains the X stores of wave 0)
    double nm_d = (double)nmiss_l;
    const double sse_full = held_out_sse(sC, IR, Xg, Yorg, Mm, d, n, r, tid);
    double v0 = wave_sum(sse_pred), v1 = wave_sum(sse_full), v2 = wave_sum(nm_d), v3 = wave_sum((double)inside_l);
    imp_barrier_full(nbar);
    if (lane == 0) { sred[wv * 4 + 0] = v0; sred[wv * 4 + 1] = v1; sred[wv * 4 + 2] = v2; sred[wv * 4 + 3] = v3; }
    imp_barrier_full(nbar);
    if (tid == 0) {
      const double tp = (sred[0] + sred[4]) + (sred[8] + sred[12]);
      const double tf = (sred[1] + sred[5]) + (sred[9] + sred[13]);
      const double tn = (sred[2] + sred[6]) + (sred[10] + sred[14]);
      const double ti = (sred[3] + sred[7]) + (sred[11] + sred[15]);
      p.Epred[(size_t)rep * p.n_iter + it] = sqrt(tp / tn);
      p.Efull[(size_t)rep * p.n_iter + it] = sqrt(tf / tn);
      if (it == p.n_iter - 1) p.inside[rep] = ti / tn;
    }
    imp_barrier_full(nbar);
  }
  for (int idx = tid; idx < d * r; idx += WG) { const int i = idx / r, l = idx - i * r; Cg[idx] = sC[i * IR + l]; }
  if (wv < 2 && bad) *errflag = 1;           // (benign race: every writer stores 1)
  imp_barrier_full(nbar);
  imp_barrier_check(nbar, errflag);
  if (tid == 0) p.err[rep] = *errflag;
  IMP_TOUT();
}

__global__ __launch_bounds__(WG) void psmf_impute_kernel2(ImputeParams p) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wv == 0) impute2_wave<0>(p);
  else if (wv == 1) impute2_wave<1>(p);
  else if (wv == 2) impute2_wave<2>(p);
  else impute2_wave<3>(p);
}

inline size_t impute2_lds_bytes(int d, int r) {
  const size_t d4 = ((size_t)d + 3) & ~(size_t)3;
  const size_t doubles = d4 * IR + IR * IR + 3 * IR + 2 * d4 + 4 * 2 * 256 + 8 + 256 + 16 + 2;
  return (doubles * 8 + 15) & ~(size_t)15;
}


}  // namespace psmf

#include "psmf_impute3.hip"

#ifndef PSMF_IMPUTE_KERNEL_ONLY
namespace {
// Which column loop a shape gets (also what psmf_impute_kernel_id reports): 2 = psmf_impute_kernel2 (round 1's loop, id 1, was removed in round 5),
// 300 + NG = psmf_impute_kernel3<NG> (d <= 80, r <= 14), 4 = the masked per-step engine of the large-d handle (any d, r <= PSMF_RMAX):
// one workgroup per replica needs the replica's C, V, x in LDS (d <= 512, r <= 16).
int impute_select(int d, int r, size_t* lds_out) {
  using namespace psmf;
  if (r > IR || d > 2 * WG) return 4;
  const bool v3 = impute3_ok(d, r) && !(getenv("PSMF_IMPUTE_V3") && atoi(getenv("PSMF_IMPUTE_V3")) == 0);   // small shapes: every wave its own Gram
  const size_t lds = v3 ? impute3_lds_bytes(d, r) : impute2_lds_bytes(d, r);
  if (lds > 160 * 1024) return 4;
  if (lds_out) *lds_out = lds;
  return v3 ? 300 + impute3_groups(d) : 2;
}
}  // namespace

int impute_run_large(const psmf_impute_config* cfg, const double* YorgInt, const uint8_t* M, const uint8_t* Mmiss, double* C, double* X,
                     const double* V, const double* P, const double* Q, double rho, double* Epred, double* Efull, double* inside,
                     double* Yrec, double* YrecL, double* YrecH, int32_t* status, float* elapsed_ms);

extern "C" int psmf_impute_kernel_id(const psmf_impute_config* cfg) {
  if (!cfg || cfg->abi_version != PSMF_ABI_VERSION || cfg->d < 1 || cfg->r < 1 || cfg->r > PSMF_RMAX) return PSMF_ERR_ARG;
  return impute_select(cfg->d, cfg->r, nullptr);
}

extern "C" int psmf_impute_run(const psmf_impute_config* cfg, const double* YorgInt, const uint8_t* M,
                               const uint8_t* Mmiss, double* C, double* X, const double* V, const double* P,
                               const double* Q, double rho, double* Epred, double* Efull, double* inside,
                               double* Yrec, double* YrecL, double* YrecH, int32_t* status, float* elapsed_ms) {
  using namespace psmf;
  auto fail = [&](int code, const std::string& msg) { g_create_error = "psmf_impute_run: " + msg; return code; };
  if (!cfg || !YorgInt || !M || !Mmiss || !C || !X || !V || !P || !Q || !Epred || !Efull || !inside)
    return fail(PSMF_ERR_ARG, "null argument");
  if (cfg->abi_version != PSMF_ABI_VERSION) return fail(PSMF_ERR_ARG, "ABI version mismatch");
  const int d = cfg->d, n = cfg->n, r = cfg->r, B = cfg->batch;
  if (r < 1 || r > PSMF_RMAX) return fail(PSMF_ERR_ARG, "need 1 <= r <= PSMF_RMAX");
  if (d < 1) return fail(PSMF_ERR_ARG, "need d >= 1");
  if (n < 2 || B < 1 || cfg->n_iter < 1) return fail(PSMF_ERR_ARG, "bad n / batch / n_iter");
  if (cfg->method < 0 || cfg->method > 3) return fail(PSMF_ERR_ARG, "method must be 0 (PSMF), 1 (rPSMF), 2 (MLE-SMF) or 3 (TMF)");
  if (cfg->want_bands && (!Yrec || !YrecL || !YrecH)) return fail(PSMF_ERR_ARG, "want_bands needs Yrec, YrecL, YrecH");
  size_t lds = 0;
  const int sel = impute_select(d, r, &lds);
  if (sel == 4)       // beyond one workgroup's LDS: the replicas one after the other on the masked per-step engine (psmf_masked.hip)
    return impute_run_large(cfg, YorgInt, M, Mmiss, C, X, V, P, Q, rho, Epred, Efull, inside, Yrec, YrecL, YrecH, status, elapsed_ms);
  const bool v3 = sel >= 300;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(PSMF_ERR_NO_DEVICE, "no HIP device visible");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(PSMF_ERR_ARG, "bad device ordinal");

#define I_TRY(expr)                                                                          \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) { rc = fail(PSMF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); goto done; } \
  } while (0)

  int rc = PSMF_OK;
  const size_t nd = (size_t)n * d, bnd = (size_t)B * nd;
  double *dY = nullptr, *dC = nullptr, *dX = nullptr, *dV = nullptr, *dP = nullptr, *dQ = nullptr, *dEp = nullptr,
         *dEf = nullptr, *dIn = nullptr, *dYr = nullptr, *dYl = nullptr, *dYh = nullptr;
  uint8_t *dM = nullptr, *dMm = nullptr;
  int* dErr = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<int> herr(B, 0);
  ImputeParams ip;
  I_TRY(hipSetDevice(cfg->device));
  I_TRY(hipMalloc((void**)&dY, nd * 8));
  I_TRY(hipMalloc((void**)&dM, bnd));
  I_TRY(hipMalloc((void**)&dMm, bnd));
  I_TRY(hipMalloc((void**)&dC, (size_t)B * d * r * 8));
  I_TRY(hipMalloc((void**)&dX, (size_t)B * n * r * 8));
  I_TRY(hipMalloc((void**)&dV, r * r * 8));
  I_TRY(hipMalloc((void**)&dP, r * r * 8));
  I_TRY(hipMalloc((void**)&dQ, r * r * 8));
  I_TRY(hipMalloc((void**)&dEp, (size_t)B * cfg->n_iter * 8));
  I_TRY(hipMalloc((void**)&dEf, (size_t)B * cfg->n_iter * 8));
  I_TRY(hipMalloc((void**)&dIn, (size_t)B * 8));
  I_TRY(hipMalloc((void**)&dErr, (size_t)B * 4));
  if (cfg->want_bands) {
    I_TRY(hipMalloc((void**)&dYr, bnd * 8));
    I_TRY(hipMalloc((void**)&dYl, bnd * 8));
    I_TRY(hipMalloc((void**)&dYh, bnd * 8));
  }
  I_TRY(hipMemcpy(dY, YorgInt, nd * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dM, M, bnd, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dMm, Mmiss, bnd, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dC, C, (size_t)B * d * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dX, X, (size_t)B * n * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dV, V, r * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dP, P, r * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dQ, Q, r * r * 8, hipMemcpyHostToDevice));
  ip.d = d; ip.n = n; ip.r = r; ip.n_iter = cfg->n_iter; ip.robust = cfg->method == 1; ip.method = cfg->method; ip.want_bands = cfg->want_bands;
  ip.sig = cfg->sig; ip.lambda0 = cfg->lambda0; ip.rho0 = rho;
  ip.Yorg = dY; ip.M = dM; ip.Mmiss = dMm; ip.C = dC; ip.X = dX; ip.V0 = dV; ip.P0 = dP; ip.Q0 = dQ;
  ip.Epred = dEp; ip.Efull = dEf; ip.inside = dIn; ip.Yrec = dYr; ip.YrecL = dYl; ip.YrecH = dYh; ip.err = dErr; ip.prof = nullptr;
  {
    bool iso = Q[0] > 0.0 && !(getenv("PSMF_IMPUTE_PAR") && atoi(getenv("PSMF_IMPUTE_PAR")) == 0);
    for (int i = 0; i < r && iso; ++i)
      for (int c = 0; c < r; ++c)
        if (Q[i * r + c] != (i == c ? Q[0] : 0.0)) { iso = false; break; }
    ip.q_iso = iso ? 1 : 0;
  }
  if (lds > 48 * 1024)
    I_TRY(hipFuncSetAttribute(v3 ? impute3_kernel(d) : (const void*)psmf_impute_kernel2,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  I_TRY(hipEventCreate(&e0));
  I_TRY(hipEventCreate(&e1));
  I_TRY(hipEventRecord(e0, 0));
  if (v3) { void* args[] = {&ip}; I_TRY(hipLaunchKernel(impute3_kernel(d), dim3(B), dim3(WG), args, lds, 0)); }
  else hipLaunchKernelGGL(psmf_impute_kernel2, dim3(B), dim3(WG), lds, 0, ip);
  I_TRY(hipGetLastError());
  I_TRY(hipEventRecord(e1, 0));
  I_TRY(hipEventSynchronize(e1));
  if (elapsed_ms) I_TRY(hipEventElapsedTime(elapsed_ms, e0, e1));
  I_TRY(hipMemcpy(C, dC, (size_t)B * d * r * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(X, dX, (size_t)B * n * r * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(Epred, dEp, (size_t)B * cfg->n_iter * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(Efull, dEf, (size_t)B * cfg->n_iter * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(inside, dIn, (size_t)B * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(herr.data(), dErr, (size_t)B * 4, hipMemcpyDeviceToHost));
  if (cfg->want_bands) {
    I_TRY(hipMemcpy(Yrec, dYr, bnd * 8, hipMemcpyDeviceToHost));
    I_TRY(hipMemcpy(YrecL, dYl, bnd * 8, hipMemcpyDeviceToHost));
    I_TRY(hipMemcpy(YrecH, dYh, bnd * 8, hipMemcpyDeviceToHost));
  }
  // Per-replica outcome (the reference records NaN for a diverged repeat and carries on, ExperimentImpute/rPSMF.py:236-243): a
  // replica whose r x r system lost positive definiteness -- or whose errors are not finite -- gets status PSMF_ERR_NUMERIC and NaN
  // results; the other replicas are untouched by it (one workgroup each).  Without a status array the call fails as a whole.
  for (int b = 0; b < B; ++b)
    if (herr[b] == 9) { rc = fail(PSMF_ERR_HIP, "internal error: the wave programs of the column loop executed different numbers of barriers (imp_barrier_check)"); goto done; }
  for (int b = 0; b < B; ++b) {
    bool bad = herr[b] != 0;
    for (int it = 0; it < cfg->n_iter && !bad; ++it)
      bad = !std::isfinite(Epred[(size_t)b * cfg->n_iter + it]) || !std::isfinite(Efull[(size_t)b * cfg->n_iter + it]);
    if (status) status[b] = bad ? PSMF_ERR_NUMERIC : PSMF_OK;
    if (!bad) continue;
    if (!status) { rc = fail(PSMF_ERR_NUMERIC, "singular r x r system in replica " + std::to_string(b)); break; }
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    for (int it = 0; it < cfg->n_iter; ++it) Epred[(size_t)b * cfg->n_iter + it] = Efull[(size_t)b * cfg->n_iter + it] = qnan;
    inside[b] = qnan;
  }
done:
  hipFree(dY); hipFree(dM); hipFree(dMm); hipFree(dC); hipFree(dX); hipFree(dV); hipFree(dP); hipFree(dQ);
  hipFree(dEp); hipFree(dEf); hipFree(dIn); hipFree(dErr); hipFree(dYr); hipFree(dYl); hipFree(dYh);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
#undef I_TRY
  return rc;
}
#endif  // PSMF_IMPUTE_KERNEL_ONLY
