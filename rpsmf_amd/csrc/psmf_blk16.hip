// psmf_blk_filter6: the general block filter (psmf_blk_filter, psmf_block.hip -- any dynamics kind, any hook configuration,
// R_k / Q_k schedules, rPSMF, in-loop Adam) for ranks r <= 16, role-specialised.
//
// psmf_blk_filter runs every stage of a timestep on all 256 threads with a workgroup barrier after each: at r = 10
// (ExperimentBeijing: FourierBasis, beijing_psmf.py:97-140) a step was 26 800 cycles, 8 500 of them the two LDS sweep
// inversions (one exchange + barrier per 2 x 2 pivot), 4 200 the r x r products of Pbar = F P F^T + Q through LDS images, 3 000
// the coefficient-space vectors -- all of it latency of barriers and LDS round trips, none of it arithmetic
// (tools/blkgen_prof.hip).  Here the r x r state lives in ONE wave's registers as 16 x 16 tiles in the MFMA output layout and
// the stages that do not depend on each other run side by side on different waves:
//   dynamics forward (all waves, psmf_dyn.hip)                                                               | barriers inside
//   A  wave 0 (matrix wave): Pbar = F P F^T + Q (eight float64 MFMAs, operands straight from the registers: P is symmetric,
//        so a tile in the output layout IS the A operand of its k-blocks), <G, Pbar>, eta -- published -- and the first
//        sweep, -Pbar^-1 (wave_sweep16m: no LDS, no barrier)
//      wave 2 (V wave): w = V mu_bar, s; with wave 0's eta: N, kappa -- published
//      wave 1 (coefficient wave, lane = coefficient row): b = A mu_bar, Ka, a, h = A^T Ka, e'e; then, when w and N are
//        there (LDS flags), the likelihood gradient g_f                                                            | barrier
//   B  wave 0: kappa G - (-Pbar^-1), augmented with kappa h in row / column r2 -> second sweep: P+ AND kappa P+ h (= mu - mu_bar)
//        AND 1 - kappa^2 h'P+h in one go; omega, phi; P, G, Q updates in registers;  wave 2: V
//      wave 1: rank-1 updates of A and K A;  waves 1-3: gradsum += J_theta^T g_f (dyn_backward on 192 threads)      | barrier
//   mu, in-loop Adam (all waves)
// Random walk with Q = q I (b.dual6; the default model at these ranks): the two inversions of a step are made independent as in
// the two-inversion kernels (psmf_block.hip, psmf_blk_filter2's header) -- wave 3 forms W_k = (M_k / beta + I / q_k)^-1 beside wave
// 0's inversion of M_k = Lbar_k + kappa G, and both take Lbar_{k+1} = (I / q_k - W_k / q_k^2) / omega_k from it: one sweep on the path
// instead of two.
// Same recursion and float64 arithmetic as psmf_blk_filter (summation orders differ).  r <= 16 (at r = 15, 16 the tile has no column left
// for the augmentation: kappa P+ h is then a product of four more MFMAs).  PSMF_FILTER6=0 sends these ranks back to psmf_blk_filter.
#pragma once
#include "psmf_blk3.hip"
#include "psmf_wave16.hip"

namespace psmf {

constexpr int F6_RMAX = 16;

// dyn_forward / dyn_backward (psmf_dyn.hip) for the trigonometric kinds at r <= 16, theta in LDS: the same sums in the same
// order, but with 16-wide index maps (thread = (term or row, column): no integer division by the runtime r) and every LDS
// operand of a sum loaded before the first use.  (The generic loops wait for one LDS round trip per term of every sum and divide
// twice per element: 5 600 + 6 000 cycles of a 17 000-cycle FourierBasis step, tools/blk16_prof.hip.)  Needs s_val, s_tp zero
// beyond column r, s_gf zero beyond r, and finite values in the 256 doubles behind theta (the caller clears them).
// What the Jacobian stage of f6_dyn_forward reads from theta, per thread: the elements M_t[i][j] of its (at most two) matrix
// elements and the gains c_t[j], four term slots each.  theta does not change inside a block unless the optimiser runs in the time
// loop (PSMFRecursive), so waves 1-3 load these ONCE per block (f6_jac_cache) instead of twelve LDS operands per element per step.
struct F6Jac {
  double m[2][4], c[4];
  bool on;
};
__device__ __forceinline__ void f6_jac_cache(F6Jac& jc, const StepParams& p, const double* th, const int tid) {
  const int r = p.r, kind = p.dyn_kind, flags = p.dyn_flags, N = p.dyn_terms, j = tid & 15;
  const int nt = dyn_n_terms(kind, N);
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const DynTerm dt = dyn_term(kind, flags, N, r, tt);
    const int mo = tt < nt ? dt.m_off : p.n_theta;                              // (a slot beyond nt: the zero tail behind theta)
    const int co = tt < nt ? (dt.c_off >= 0 ? dt.c_off : 0) : p.n_theta;
    const bool cg = tt >= nt || dt.c_off >= 0;
    jc.c[tt] = cg ? th[co + j] : 1.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = (tid + 192 * k) >> 4;
      jc.m[k][tt] = (i < r && j < r) ? th[mo + i * r + j] : 0.0;
    }
  }
}

// th, g: theta and the gradient sum AS LDS ARRAYS (through StepParams they are generic pointers: flat loads)
__device__ __forceinline__ void f6_dyn_forward(const StepParams& p, const double* th, const F6Jac& jc, const double tk, const double* s_x, double* s_mub, double* s_fd,
                                               double* sF, const int ldf, double* s_val, double* s_tp, double* s_part, const int tid) {
  // tid = 0 .. 191: waves 1-3 (the matrix wave only keeps the barrier count, f6_dyn_barriers)
  const int r = p.r, kind = p.dyn_kind, flags = p.dyn_flags, N = p.dyn_terms;
  const int nt = dyn_n_terms(kind, N);
  const int t = tid >> 4, j = tid & 15;
  const bool act = t < nt && j < r;
  const DynTerm d = dyn_term(kind, flags, N, r, min(t, nt - 1));
  const bool dense = dyn_dense(kind, flags);
  if (dense && nt <= 4) {
    // At most four terms: the (term, column) pairs fit ONE wave -- the trig values, then (behind the one barrier F needs anyway)
    // the matrix-vector products and the sum over the terms without another (the terms of a row sit in the four 16-lane rows of
    // the wave: two lane swaps).  Two barriers instead of three.
    if (tid < 64 && act) {
      const double c = d.c_off >= 0 ? th[d.c_off + j] : 1.0;
      double sn, cs;
      dyn_sincospi(2.0 * th[d.b_off + j] * tk + (c * s_x[j]) * 0.31830988618379067154, sn, cs);
      s_val[t * RM + j] = d.is_cos ? cs : sn;
      s_tp[t * RM + j] = d.is_cos ? -sn : cs;
    }
    __syncthreads();
    {                         // F[i][j], thread = element, twelve rows per pass.  Straight-line over four term slots -- a slot
      // beyond nt points at the zero tail behind theta -- so that the twelve LDS operands of an element are all in flight before
      // the first is used (with a uniform branch per term each waited for its own round trip: 2 350 of the step's 12 100 cycles).
      int mo[4], co[4];
      bool cg[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const DynTerm dt = dyn_term(kind, flags, N, r, tt);
        mo[tt] = tt < nt ? dt.m_off : p.n_theta;       // (dense, trigonometric: every term has a matrix)
        cg[tt] = tt >= nt || dt.c_off >= 0;
        co[tt] = tt < nt ? (dt.c_off >= 0 ? dt.c_off : 0) : p.n_theta;
      }
      if (jc.on) {            // theta fixed for the block: M_t[i][j], c_t[j] from registers
        double tp[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) tp[tt] = s_tp[tt * RM + j] * jc.c[tt];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int idx = tid + 192 * k, i = idx >> 4;
          if (i < r && j < r) {
            double a = 0.0;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) a += jc.m[k][tt] * tp[tt];
            sF[i * ldf + j] = a;
          }
        }
      } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 192 * k, i = idx >> 4;
        if (i < r && j < r) {
          double mv[4], tp[4], cj[4];
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) { mv[tt] = th[mo[tt] + i * r + j]; tp[tt] = s_tp[tt * RM + j]; cj[tt] = th[co[tt] + j]; }
          double a = 0.0;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) a += mv[tt] * (tp[tt] * (cg[tt] ? cj[tt] : 1.0));
          sF[i * ldf + j] = a;
        }
      }
      }
    }
    if (tid < 64) {           // mu_bar: the terms' matrix-vector products (lane = (term, row)), then the sum over the lane rows
      double a = 0.0;
      if (act) {
        if (d.m_off >= 0) {
          const double* row = th + d.m_off + j * r;
          double mv[F6_RMAX], sv[F6_RMAX];
#pragma unroll
          for (int q = 0; q < F6_RMAX; ++q) { mv[q] = row[q]; sv[q] = s_val[t * RM + q]; }     // (columns >= r: finite times zero)
#pragma unroll
          for (int q = 0; q < F6_RMAX; ++q) a += mv[q] * sv[q];
        } else {
          a = s_val[t * RM + j];
        }
      }
      const double mb = xor32_sum_f64(xor16_sum_f64(a));
      if (tid < r) s_mub[tid] = mb;
    }
    __syncthreads();
    return;
  }
  if (act) {
    const double c = d.c_off >= 0 ? th[d.c_off + j] : 1.0;
    double sn, cs;
    dyn_sincospi(2.0 * th[d.b_off + j] * tk + (c * s_x[j]) * 0.31830988618379067154, sn, cs);
    s_val[t * RM + j] = d.is_cos ? cs : sn;
    s_tp[t * RM + j] = d.is_cos ? -sn : cs;
  }
  __syncthreads();
  if (!dense && kind != DYN_FOURIER) {      // one term, no matrix: mu_bar and the diagonal of F straight from the trig values
    if (tid < r) {
      s_mub[tid] = s_val[tid];
      s_fd[tid] = s_tp[tid] * (d.c_off >= 0 ? th[d.c_off + tid] : 1.0);      // (t = 0 for these threads)
    }
    __syncthreads();
    return;
  }
  if (act) {      // the term's share of mu_bar_i, i = j
    double a;
    if (d.m_off >= 0) {
      const double* row = th + d.m_off + j * r;
      double mv[F6_RMAX], sv[F6_RMAX];
#pragma unroll
      for (int q = 0; q < F6_RMAX; ++q) { mv[q] = row[q]; sv[q] = s_val[t * RM + q]; }     // (columns >= r: finite times zero)
      a = 0.0;
#pragma unroll
      for (int q = 0; q < F6_RMAX; ++q) a += mv[q] * sv[q];
    } else {
      a = s_val[t * RM + j];
    }
    s_part[t * RM + j] = a;
  }
  if (dense) {    // F[i][j], thread = element
#pragma unroll
   for (int k = 0; k < 2; ++k) {
    const int idx = tid + 192 * k, i = idx >> 4;
    if (i < r && j < r) {
      double mv[DYN_MAX_TERMS], dv[DYN_MAX_TERMS];
#pragma unroll
      for (int tt = 0; tt < DYN_MAX_TERMS; ++tt) {
        mv[tt] = 0.0; dv[tt] = 0.0;
        if (tt < nt) {
          const DynTerm dt = dyn_term(kind, flags, N, r, tt);
          dv[tt] = s_tp[tt * RM + j] * (dt.c_off >= 0 ? th[dt.c_off + j] : 1.0);
          mv[tt] = dt.m_off >= 0 ? th[dt.m_off + i * r + j] : ((i == j) ? 1.0 : 0.0);
        }
      }
      double a = 0.0;
#pragma unroll
      for (int tt = 0; tt < DYN_MAX_TERMS; ++tt)
        if (tt < nt) a += mv[tt] * dv[tt];
      sF[i * ldf + j] = a;
    }
   }
  }
  __syncthreads();
  if (tid < r) {
    double a = 0.0, fd = 0.0;
    for (int tt = 0; tt < nt; ++tt) {
      const DynTerm dt = dyn_term(kind, flags, N, r, tt);
      a += s_part[tt * RM + tid];
      if (dt.m_off < 0) fd += s_tp[tt * RM + tid] * (dt.c_off >= 0 ? th[dt.c_off + tid] : 1.0);
    }
    s_mub[tid] = a;
    if (!dense) s_fd[tid] = fd;
  }
  __syncthreads();
}

// barriers of dyn_forward (trig16 = false) / f6_dyn_forward for a wave that takes no part in it
__device__ __forceinline__ int f6_dyn_barriers(const StepParams& p, const bool trig16) {
  if (p.dyn_kind == DYN_RANDOM_WALK || p.dyn_kind == DYN_SCALED_WALK) return 1;
  const bool dense = dyn_dense(p.dyn_kind, p.dyn_flags);
  if (trig16 && dense && dyn_n_terms(p.dyn_kind, p.dyn_terms) <= 4) return 2;
  return (!dense && p.dyn_kind != DYN_FOURIER) ? 2 : 3;
}

// gradsum += J_theta^T g_f on NTH = 192 threads (tid3 = 0 .. 191); ends with a barrier
__device__ __forceinline__ void f6_dyn_backward(const StepParams& p, const double* th, double* g, const double tk, const double* s_x,
                                                const double* s_gf, const double* s_val, const double* s_tp, const int tid3) {
  const int r = p.r, kind = p.dyn_kind, flags = p.dyn_flags, N = p.dyn_terms;
  const int nt = dyn_n_terms(kind, N);
  // d/dM_t[i][j] = g_f[i] trig_t(arg_tj): thread = element (i, j), every term
  if (nt <= 4) {        // straight-line over four term slots (a slot beyond nt: the zero tail behind the gradient sums), loads first
    int mo[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) { const DynTerm dt = dyn_term(kind, flags, N, r, tt); mo[tt] = (tt < nt && dt.m_off >= 0) ? dt.m_off : p.n_theta; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid3 + 192 * k, i = idx >> 4, j = idx & 15;
      if (i < r && j < r) {
        const double gi = s_gf[i];
        double gv[4], sv[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) { gv[tt] = g[mo[tt] + i * r + j]; sv[tt] = s_val[tt * RM + j]; }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) g[mo[tt] + i * r + j] = fma(gi, sv[tt], gv[tt]);
      }
    }
  } else {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid3 + 192 * k, i = idx >> 4, j = idx & 15;
    if (i < r && j < r) {
      const double gi = s_gf[i];
#pragma unroll
      for (int tt = 0; tt < DYN_MAX_TERMS; ++tt)
        if (tt < nt) {
          const DynTerm dt = dyn_term(kind, flags, N, r, tt);
          if (dt.m_off >= 0) g[dt.m_off + i * r + j] += gi * s_val[tt * RM + j];
        }
    }
  }
  }
  // d/db_t[j], d/dc_t[j] = (M_t^T g_f)_j trig_t'(arg_tj) {2 pi k, x_j}: thread = (t, j), starting on the SECOND of the three waves
  // (the first one has the rank-1 updates of the coefficient matrices in this phase)
  const int t = ((tid3 + 128) % 192) >> 4, j = tid3 & 15;
  if (t < nt && j < r) {
    const DynTerm d = dyn_term(kind, flags, N, r, t);
    double u;
    if (d.m_off >= 0) {
      double mv[F6_RMAX], gv[F6_RMAX];
#pragma unroll
      for (int i = 0; i < F6_RMAX; ++i) { mv[i] = th[d.m_off + i * r + j]; gv[i] = s_gf[i]; }     // (rows >= r: finite times zero)
      u = 0.0;
#pragma unroll
      for (int i = 0; i < F6_RMAX; ++i) u += mv[i] * gv[i];
    } else {
      u = s_gf[j];
    }
    const double ut = u * s_tp[t * RM + j];
    g[d.b_off + j] += ut * (2.0 * M_PI * tk);
    if (d.c_off >= 0) g[d.c_off + j] += ut * s_x[j];
  }
  __syncthreads();
}

// ROLE: the wave's role as a compile-time constant (0 matrix wave, 1 coefficient wave, 2 V wave, 3 the fourth): one program per role,
// each holding only its own registers.  With the wave index as a run-time value the matrix wave's ~170 registers of state and
// lane constants were live across the all-thread dynamics code of every wave and were moved out and back around it each step.
// DUAL (compile-time as well: the extra selects cost the other configurations 4-5 % as run-time branches): the random walk's two
// inversions side by side, b.dual6.
template <int ROLE, bool DUAL>
__device__ __forceinline__ void f6_program(const BlockParams& b) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const StepParams& p = b.sp;
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x;
  constexpr int wv = ROLE;
  const int lane = tid & 63, lk = lane >> 4, lr = lane & 15;
  const int r2 = r + (r & 1);
  const double dd = (double)p.d;
  // ---- LDS carve: psmf_blk_filter's (blk_filter_lds_bytes), so that assemble_K and the dynamics see the arrays they know ----
  double* sK = sm;                    // RB x RB
  double* sA = sK + RB * RB;          // RB x r, row stride RS
  double* sKA = sA + RB * RS;         // RB x r
  double* s_img = sKA + RB * RS;      // WG: a 16 x 16 tile image of wave 0 (transposition)
  double* s_mub = s_img + WG;         // RM each below
  double* s_f = s_mub + RM;
  double* s_w = s_f + RM;
  double* s_h = s_w + RM;
  double* s_munew = s_h + RM;
  double* s_mu = s_munew + RM;
  double* s_a = s_mu + RM;            // RB
  double* s_Ka = s_a + RB;            // RB
  double* s_sc = s_Ka + RB;           // (8 RB:) 0 s, 1 eta, 2 N, 3 1 / N, 4 kappa, 5 lambda of the step, 6 e'e, 7 rho of the step
  int* s_flag = reinterpret_cast<int*>(s_sc + 16);     // number of steps whose w, s, N, 1 / N, kappa wave 2 has published
  int* s_flagA = s_flag + 1;                           // ... whose eta, lambda, rho wave 0 has published
  double* rowbuf = s_sc + 8 * RB;     // 4 * RM (unused here)
  double* s4 = rowbuf + 4 * RM;       // 4 (+ errflag)
  int* errflag = reinterpret_cast<int*>(s4 + 4);
  double* sF = s4 + 6;                // RM/2 x RS: dense Jacobian
  double* sPm = sF + (RM / 2) * RS;
  double* sW = sPm;                   // dual: W_{k-1} as a tile image (256 of the 272 doubles; s_sc 8: 1 / omega_{k-1}, 9: 1 / q_{k-1}, 10: 1 / q_k)
  double* sT = sPm + (RM / 2) * RS;   // scratch of the dynamics
  double* s_val = sT + (RM / 2) * RS; // DYN_MAX_TERMS x RM
  double* s_tp = s_val + DYN_MAX_TERMS * RM;
  double* s_gf = s_tp + DYN_MAX_TERMS * RM;   // RM
  double* s_u = s_gf + RM;            // RM
  double* s_theta = s_u + RM;         // BLK_TH_CAP   (theta and gradsum of the block, when they fit)
  double* s_grad = s_theta + BLK_TH_CAP;
  const bool dense = dyn_dense(p.dyn_kind, p.dyn_flags);
  const bool th_lds = p.n_theta > 0 && p.n_theta <= BLK_TH_CAP;
  const bool has_bw = p.n_theta > 0 && p.dyn_kind != DYN_RANDOM_WALK;
  constexpr bool dual = DUAL;
  StepParams pd = p;                  // what the dynamics see: theta / gradsum in LDS when they fit
  if (th_lds) { pd.theta = s_theta; pd.gradsum = s_grad; }

  if (!blk_handoff_begin(b)) return;
  if (th_lds)
    for (int idx = tid; idx < p.n_theta; idx += WG) { s_theta[idx] = p.theta[idx]; s_grad[idx] = p.gradsum[idx]; }
  if (!b.assemble) {
    for (int idx = tid; idx < RB * RB; idx += WG) sK[idx] = b.K[idx];
  } else {
    assemble_K<WG>(b, sK, sA, sKA, r, tid);
  }
  if (tid == 0) { *errflag = 0; *s_flag = 0; *s_flagA = 0; }
  if (tid < RM) { s_mub[tid] = 0.0; s_h[tid] = 0.0; s_w[tid] = 0.0; s_f[tid] = 1.0; s_munew[tid] = 0.0; s_gf[tid] = 0.0; }
  for (int idx = tid; idx < (RM / 2) * RS; idx += WG) sF[idx] = 0.0;        // wave 0 reads whole 16 x 16 tiles: zero outside r x r
  for (int idx = tid; idx < DYN_MAX_TERMS * RM; idx += WG) { s_val[idx] = 0.0; s_tp[idx] = 0.0; }
  if (th_lds && tid < 256 && p.n_theta + tid < BLK_TH_CAP) { s_theta[p.n_theta + tid] = 0.0; s_grad[p.n_theta + tid] = 0.0; }
  // the 16-wide dynamics (f6_dyn_forward / f6_dyn_backward): trigonometric kinds with theta in LDS
  const bool trig16 = th_lds && (p.dyn_kind == DYN_COS_PHASE || p.dyn_kind == DYN_SINUSOID || p.dyn_kind == DYN_FOURIER);
  const int nbar_fwd = f6_dyn_barriers(p, trig16);
  if (tid < r) s_mu[tid] = st->mu[tid];
  // ---- wave 0: P, Q, G, wave 2: V as 16 x 16 tiles (element (lk + 4 q, lr)); the lane predicates as multipliers ----
  double Vm[4] = {0.0, 0.0, 0.0, 0.0}, Pm[4] = {0.0, 0.0, 0.0, 0.0}, Qm[4] = {0.0, 0.0, 0.0, 0.0}, Gm[4] = {0.0, 0.0, 0.0, 0.0};
  double finq[4], fdg[4], fpad[4], faugc[4], faugr[4], fxr[4];
  int trx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = lk + 4 * q;
    const bool in = i < r && lr < r;
    finq[q] = in ? 1.0 : 0.0;
    fdg[q] = (in && i == lr) ? 1.0 : 0.0;
    fpad[q] = (!in && i == lr) ? 1.0 : 0.0;
    faugc[q] = (lr == r2 && i < r) ? 1.0 : 0.0;       // column r2: kappa h_i
    faugr[q] = (i == r2 && lr < r) ? 1.0 : 0.0;       // row r2: kappa h_j
    fxr[q] = i < r ? 1.0 : 0.0;
    trx[q] = (lr >> 2) * 64 + (lr & 3) * 16 + i;      // the transposed element in a tile image
    if (wv == 0) {
      const int idx = in ? i * r + lr : 0;
      const double lq = st->Q[idx], lp = st->P[idx];
      Qm[q] = in ? lq : 0.0;
      Pm[q] = in ? lp : 0.0;
    }
    if (wv == 2) {
      const double lv = st->V[in ? i * r + lr : 0];
      Vm[q] = in ? lv : 0.0;
    }
  }
  const int rq_c = r2 >> 2, ln_c = ((r2 & 3) << 4) | r2;       // where element (r2, r2) sits
  Sw16K swk;
  if (wv == 0 || wv == 3) sw16k_init(swk, lk, lr);
  double qv = st->Q[0], iqv = 1.0 / st->Q[0];      // dual: the running q of Q = q I and its reciprocal (wave 0)
  double rho = st->rho, lam = st->lam;
  bool bad = false;
  __syncthreads();
  // A_0 = [I; 0], K A_0 = first r columns of K, G_0 = K[0:r, 0:r] (exact Gram of the stored C); columns r .. RS - 1 zero (the
  // row loops below run over F6_RMAX columns)
  for (int idx = tid; idx < RB * RS; idx += WG) {
    const int m = idx / RS, c = idx - m * RS;
    sA[idx] = (m == c && c < r) ? 1.0 : 0.0;
    sKA[idx] = c < r ? sK[m * RB + c] : 0.0;
  }
  if (wv == 0 || (wv == 3 && dual)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) Gm[q] = finq[q] != 0.0 ? sK[(lk + 4 * q) * RB + lr] : 0.0;
  }
  if (wv == 3 && dual) {
    // Lbar_1 = (P + q I)^-1 by one sweep, handed over as the W that reproduces it: W = q I - q^2 Lbar (omega = 1)
    double A0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = lk + 4 * q;
      const double lp = st->P[finq[q] != 0.0 ? i * r + lr : 0];
      A0[q] = finq[q] * lp + fdg[q] * qv + fpad[q];
    }
    wave_sweep16m(A0, r2, swk, bad);
#pragma unroll
    for (int q = 0; q < 4; ++q) sW[q * 64 + lane] = fdg[q] * qv + finq[q] * qv * qv * A0[q];
    if (lane == 0) { s_sc[8] = 1.0; s_sc[9] = iqv; s_sc[10] = iqv; }
  }
  __syncthreads();

  F6Jac jac;
  jac.on = false;
  if (wv != 0 && trig16 && dense && !p.recursive && dyn_n_terms(p.dyn_kind, p.dyn_terms) <= 4) {
    f6_jac_cache(jac, p, s_theta, tid - 64);
    jac.on = true;
  }
  double s_last = 0.0, eta_last = 0.0, N_last = 0.0, phi = 1.0, omega = 1.0, ee_last = 0.0;
  BLK_T0();
  for (int jb = 0; jb < b.nb; ++jb) {
    const long long kstep = b.k0 + jb + 1;   // 1-based step index
    // ---- mu_bar = f(theta, mu, k), F = df/dx (psmf.py:104-115; psmf_dyn.hip) ----
    // (waves 1-3; the matrix wave keeps its registers and only joins the barriers)
    if (wv == 0) { for (int q = 0; q < nbar_fwd; ++q) __syncthreads(); }
    else if (trig16) f6_dyn_forward(p, s_theta, jac, (double)kstep, s_mu, s_mub, s_f, sF, RS, s_val, s_tp, sT, tid - 64);     // both end with a barrier
    else dyn_forward<WG - 64>(pd, (double)kstep, s_mu, s_mub, s_f, sF, RS, s_val, s_tp, sT, tid - 64);
    BLK_T(0);
    // PSMFIter reads Q[k], R[k] of the step (psmf.py:115,123,141): scalar schedules (never with rPSMF's running Q, R)
    const double qs = p.q_sched ? p.q_sched[kstep - p.series_t0] : 1.0;
    if (p.rho_sched) rho = p.rho_sched[kstep - p.series_t0];
    double A[4] = {0.0, 0.0, 0.0, 0.0}, Pb[4] = {0.0, 0.0, 0.0, 0.0}, wrow[4] = {0.0, 0.0, 0.0, 0.0}, mb[4] = {0.0, 0.0, 0.0, 0.0};
    double s = 0.0, eta = rho, N = 1.0, invN = 1.0, kappa = 0.0;
    if (wv == 0) {
      // ================= phase A, matrix wave =================
#pragma unroll
      for (int q = 0; q < 4; ++q) mb[q] = s_mub[lk + 4 * q];         // mu_bar of this lane's rows (for mu = mu_bar + kappa P+ h)
      double Lb[4] = {0.0, 0.0, 0.0, 0.0};
      if (dual) {
        // Pbar = P + q I (only <G, Pbar> needs it); Lbar_k from W_{k-1}
        const double k1 = s_sc[8] * s_sc[9], k2 = k1 * s_sc[9];
#pragma unroll
        for (int q = 0; q < 4; ++q) { Pb[q] = fma(fdg[q], qv, Pm[q]); Lb[q] = fma(-k2, sW[q * 64 + lane], fma(fdg[q], k1, fpad[q])); }
      } else if (p.pbar_predict) {
        if (dense) {
          // Pbar = F P F^T + Q: T = P F^T, then F T
          double Fr[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) Fr[q] = sF[lr * RS + lk + 4 * q];            // F[lr][lk + 4 q]
          f64x4 t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
          t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pm[0], Fr[0], t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pm[1], Fr[1], t1, 0, 0, 0);
          t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pm[2], Fr[2], t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pm[3], Fr[3], t1, 0, 0, 0);
          f64x4 u0 = {0.0, 0.0, 0.0, 0.0}, u1 = {0.0, 0.0, 0.0, 0.0};
          u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Fr[0], t0[0] + t1[0], u0, 0, 0, 0);
          u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Fr[1], t0[1] + t1[1], u1, 0, 0, 0);
          u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Fr[2], t0[2] + t1[2], u0, 0, 0, 0);
          u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Fr[3], t0[3] + t1[3], u1, 0, 0, 0);
          // symmetrise (F P F^T is symmetric up to round-off) through a tile image: (X + X^T) / 2, bitwise symmetric
#pragma unroll
          for (int q = 0; q < 4; ++q) { Pb[q] = fma(qs, Qm[q], u0[q] + u1[q]); s_img[q * 64 + lane] = Pb[q]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) Pb[q] = 0.5 * (Pb[q] + s_img[trx[q]]);
        } else {
          const double fc = s_f[lr];
#pragma unroll
          for (int q = 0; q < 4; ++q) Pb[q] = fma(s_f[lk + 4 * q] * Pm[q], fc, qs * Qm[q]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) Pb[q] = Pm[q];
      }
      if (p.eta_full) {
        const double tr = fma(Gm[0], Pb[0], Gm[1] * Pb[1]) + fma(Gm[2], Pb[2], Gm[3] * Pb[3]);
        eta += wave_sum_f64_dpp(tr) / dd;
      }
      if (lane == 0) { s_sc[1] = eta; s_sc[5] = lam; s_sc[7] = rho; }
      // (LDS operations of one wave complete in program order: the flag needs no release fence -- which would also wait for
      //  this wave's outstanding global stores)
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(s_flagA, jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      BLK_T(1);
      if (dual) {
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = Lb[q];         // (kappa G and the augmented column are added in phase B)
      } else if (p.coef_update) {
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = Pb[q] + fpad[q];
        wave_sweep16m(A, r2, swk, bad);                 // -Pbar^-1
      }
    } else if (wv == 2) {
      // ================= phase A, V wave: w = V mu_bar, s; N, kappa when wave 0's eta is there =================
#pragma unroll
      for (int q = 0; q < 4; ++q) mb[q] = s_mub[lk + 4 * q];         // mu_bar of this lane's rows = the B operand of V mu_bar
      {
        f64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vm[0], mb[0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vm[1], mb[1], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vm[2], mb[2], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vm[3], mb[3], a1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) wrow[q] = a0[q] + a1[q];          // w_i = (V mu_bar)_i, i = lk + 4 q, in every column
      }
      s = xor32_sum_f64(xor16_sum_f64(fma(mb[0], wrow[0], mb[1] * wrow[1]) + fma(mb[2], wrow[2], mb[3] * wrow[3])));
      if (lr == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s_w[lk + 4 * q] = wrow[q];
      }
      while (__hip_atomic_load(s_flagA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < jb + 1) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      N = s + s_sc[1];
      invN = fast_rcp(N);
      kappa = fast_rcp(s_sc[7] + s);
      if (lane == 0) { s_sc[0] = s; s_sc[2] = N; s_sc[3] = invN; s_sc[4] = kappa; }
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(s_flag, jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (wv == 3) {
      // ================= phase A, W wave (dual): Lbar_k from W_{k-1}, and the scalars wave 0 rewrites in phase B =================
      if (dual) {
        const double k1 = s_sc[8] * s_sc[9], k2 = k1 * s_sc[9];
        kappa = s_sc[10];                                   // (1 / q_k, parked in `kappa` until phase B)
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = fma(-k2, sW[q * 64 + lane], fma(fdg[q], k1, fpad[q]));
      }
    } else if (wv == 1) {
      // ================= phase A, coefficient wave: lane = coefficient row =================
      const int m = lane;
      double pb = 0.0, pk = 0.0;
      {
        double av[F6_RMAX], kv[F6_RMAX], mv[F6_RMAX];
#pragma unroll
        for (int c = 0; c < F6_RMAX; ++c) { av[c] = sA[m * RS + c]; kv[c] = sKA[m * RS + c]; mv[c] = s_mub[c]; }   // (zero beyond r)
#pragma unroll
        for (int c = 0; c < F6_RMAX; ++c) { pb = fma(av[c], mv[c], pb); pk = fma(kv[c], mv[c], pk); }
      }
      const double am = (m == r + jb ? 1.0 : 0.0) - pb;
      const double kam = sK[m * RB + r + jb] - pk;
      s_a[m] = am;
      s_Ka[m] = kam;
      b.Bcoef[(size_t)jb * RB + m] = pb;
      // h = A^T Ka: column j = lr, the sixteen rows 16 lk .. 16 lk + 15 per lane, then across the four lane rows
      double ph = 0.0;
      {
        double av[16], kv[16];
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) { av[mm] = sA[(16 * lk + mm) * RS + lr]; kv[mm] = s_Ka[16 * lk + mm]; }   // (columns >= r: zero)
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) ph = fma(av[mm], kv[mm], ph);
      }
      const double hj = xor32_sum_f64(xor16_sum_f64(ph));
      if (lane < 16) s_h[lane] = hj;
      const double ee1 = wave_sum_f64_dpp(am * kam);
      if (lane == 0) s_sc[6] = ee1;
      // theta gradient at the pre-update state: g_f = d(incremental likelihood)/df (psmf.py:57-64, rpsmf.py:62-71)
      if (has_bw) {
        while (__hip_atomic_load(s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < jb + 1) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (lane < r) {
          const double Nn = s_sc[2], iN = s_sc[3], lm = s_sc[5], wi = s_w[lane];
          double gf;
          if (p.robust) {
            const double D = lm * Nn;
            gf = dd * wi / Nn + 0.5 * (dd + lm) * (-2.0 * hj / D - 2.0 * lm * ee1 * wi / (D * D)) / (1.0 + ee1 / D);
          } else {
            gf = dd * wi * iN - hj * iN - ee1 * wi * iN * iN;
          }
          s_gf[lane] = gf;
        }
      }
    }
    BLK_T(2);
    __syncthreads();                                     // ---- A | B
    BLK_T(3);
    if (wv == 0) {
      // ================= phase B, matrix wave =================
      const double ee = s_sc[6];
      s = s_sc[0]; N = s_sc[2]; invN = s_sc[3]; kappa = s_sc[4];
      double hrow[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { hrow[q] = s_h[lk + 4 * q]; wrow[q] = s_w[lk + 4 * q]; }
      const double hcol = s_h[lr], wcol = s_w[lr];
      double Pp[4], quad = kappa * ee;
      if (p.coef_update) {
        // M = Pbar^-1 + kappa G, augmented with kappa h in row / column r2
        const double khc = kappa * hcol;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          A[q] = fma(kappa, Gm[q], dual ? A[q] : fpad[q] - finq[q] * A[q]) + (faugc[q] * (kappa * hrow[q]) + faugr[q] * khc);
        wave_sweep16m(A, r2, swk, bad);                 // [[-P+, kappa P+ h], [., 1 - kappa^2 h'P+h]]
        BLK_T(4);
#pragma unroll
        for (int q = 0; q < 4; ++q) Pp[q] = -finq[q] * A[q];
        if (r2 < 16) {
          const double a_c = rq_c == 0 ? A[0] : (rq_c == 1 ? A[1] : (rq_c == 2 ? A[2] : A[3]));
          quad += readlane_f64(a_c, ln_c) - 1.0;          // kappa e'e - kappa^2 h'P+h  (psmf.py:155-165)
          if (lr == r2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_munew[lk + 4 * q] = fma(fxr[q], A[q], mb[q]);      // mu = mu_bar + kappa P+ h
          }
        } else {
          // r = 15, 16: no tile column left for the augmentation -- kappa P+ h as a product (the swept tile, symmetric, is the A
          // operand of its k-blocks; kappa h of the lane's rows in every column the B operand), then kappa^2 h'P+h
          double kh[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) kh[q] = kappa * hrow[q];
          f64x4 z0 = {0.0, 0.0, 0.0, 0.0}, z1 = {0.0, 0.0, 0.0, 0.0};
          z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0], kh[0], z0, 0, 0, 0);
          z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[1], kh[1], z1, 0, 0, 0);
          z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[2], kh[2], z0, 0, 0, 0);
          z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[3], kh[3], z1, 0, 0, 0);
          double dz[4], part = 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q) { dz[q] = -fxr[q] * (z0[q] + z1[q]); part = fma(kh[q], dz[q], part); }      // (kappa P+ h)_i, i = lk + 4 q
          quad -= xor32_sum_f64(xor16_sum_f64(part));
          if (lr == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_munew[lk + 4 * q] = mb[q] + dz[q];
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) Pp[q] = Pb[q];
        if (lr == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) s_munew[lk + 4 * q] = mb[q];
        }
      }
      double pscale = 1.0, qscale = 1.0;
      phi = 1.0; omega = 1.0;
      if (p.robust) {
        const double ild = fast_rcp(lam + dd);
        phi = (lam + ee * invN) * ild;
        omega = (lam + quad) * ild;
        if (p.coef_update) { pscale = p.beta * omega; qscale = omega; }
        rho *= omega;
        if (!p.fixed_lambda) lam += dd;
      }
      if (dual) {          // for the next step's Lbar: 1 / omega_k, 1 / q_k (the q W_k is formed with), 1 / q_{k+1}
        const double io = p.robust ? fast_rcp(omega) : 1.0;
        if (lane == 0) { s_sc[8] = io; s_sc[9] = iqv; s_sc[10] = iqv * io; }
        iqv *= io;
        qv *= qscale;
      }
      // P, G, Q of the step (psmf.py:150-170; G: the tracked Gram of C)
      const double wj = wcol * invN, ew = ee * invN;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Pm[q] = pscale * Pp[q];
        Gm[q] += finq[q] * (fma(hrow[q], wj, wrow[q] * (hcol * invN)) + ew * (wrow[q] * wj));
        Qm[q] *= qscale;
      }
      s_last = s; eta_last = eta; N_last = N; ee_last = ee;
      BLK_T(5);
      if (has_bw) __syncthreads();                       // (the barrier that ends dyn_backward on the other waves)
    } else {
      // ================= phase B, wave 2: V of the step (psmf.py:166-170, rpsmf.py: phi) =================
      if (wv == 2) {
        double vscale = 1.0;
        if (p.robust) { const double lm = s_sc[5]; vscale = p.alpha * ((lm + s_sc[6] * invN) * fast_rcp(lm + dd)); }
        const double wcol = s_w[lr];
#pragma unroll
        for (int q = 0; q < 4; ++q) Vm[q] = vscale * fma(-(wrow[q] * wcol), invN, Vm[q]);       // (w_i w_j first: bitwise symmetric)
      }
      // ================= phase B, W wave (dual): W_k = (M_k / beta + I / q_k)^-1, M_k = Lbar_k + kappa G; its own copy of G =================
      if (wv == 3 && dual) {
        const double iqt = kappa, kap = s_sc[4], iN = s_sc[3], ee = s_sc[6];
        const double ib = p.robust ? fast_rcp(p.beta) : 1.0;
        double hrow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { hrow[q] = s_h[lk + 4 * q]; wrow[q] = s_w[lk + 4 * q]; }
        const double hcol = s_h[lr], wcol = s_w[lr];
        // (A holds Lbar_k + the identity padding: scaling the padding by 1 / beta is harmless, it is swept on its own)
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = fma(ib, fma(kap, Gm[q], A[q]), fdg[q] * iqt);
        wave_sweep16m(A, r2, swk, bad);
#pragma unroll
        for (int q = 0; q < 4; ++q) sW[q * 64 + lane] = -finq[q] * A[q];       // read behind the barrier that ends the step
        const double wj = wcol * iN, ew = ee * iN;
#pragma unroll
        for (int q = 0; q < 4; ++q) Gm[q] += finq[q] * (fma(hrow[q], wj, wrow[q] * (hcol * iN)) + ew * (wrow[q] * wj));
      }
      // ================= phase B, wave 1: rank-1 updates of the coefficient matrices (lane = row) =================
      if (wv == 1) {
        const int m = lane;
        const double iN = s_sc[3];
        const double am = s_a[m] * iN, km = s_Ka[m] * iN;
        double av[F6_RMAX], kv[F6_RMAX], wc[F6_RMAX];
#pragma unroll
        for (int c = 0; c < F6_RMAX; ++c) { av[c] = sA[m * RS + c]; kv[c] = sKA[m * RS + c]; wc[c] = s_w[c]; }     // (w is zero beyond r)
#pragma unroll
        for (int c = 0; c < F6_RMAX; ++c) { sA[m * RS + c] = fma(am, wc[c], av[c]); sKA[m * RS + c] = fma(km, wc[c], kv[c]); }
      }
      // ================= phase B, waves 1-3: gradsum += J_theta^T g_f =================
      if (has_bw) {                                       // both end with a barrier
        if (trig16) f6_dyn_backward(p, s_theta, s_grad, (double)kstep, s_mu, s_gf, s_val, s_tp, tid - 64);
        else dyn_backward<WG - 64>(pd, (double)kstep, s_mu, s_gf, s_val, s_tp, tid - 64);
      }
    }
    BLK_T(6);
    __syncthreads();           // the step's mu, A, K A are complete; every read of s_mu, s_w, s_h, s_a, s_Ka is done
    if (tid < r) {
      const double mu_new = s_munew[tid];
      s_mu[tid] = mu_new;
      if (p.mu_hist) p.mu_hist[(size_t)(kstep - p.series_t0) * r + tid] = mu_new;
    }
    __syncthreads();
    // PSMFRecursive: optimiser step on theta every update_every observations (psmf.py:299-304)
    if (p.recursive && p.n_theta > 0 && (kstep % p.update_every) == 0) {
      if (wv == 0) __syncthreads();
      else dyn_adam_step<WG - 64>(pd, kstep, tid - 64);        // ends with a barrier
    }
    BLK_T(7);
  }
  BLK_TOUT();

  // ---- block end: coefficients and state back to memory ----
  for (int idx = tid; idx < RB * r; idx += WG) { const int m = idx / r; b.Acoef[idx] = sA[m * RS + (idx - m * r)]; }
  if (th_lds)
    for (int idx = tid; idx < p.n_theta; idx += WG) { p.gradsum[idx] = s_grad[idx]; if (p.recursive) p.theta[idx] = s_theta[idx]; }
  if (wv == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = lk + 4 * q;
      if (i < r && lr < r) {
        const int idx = i * r + lr;
        st->P[idx] = Pm[q];
        st->Q[idx] = Qm[q];
        st->G[idx] = Gm[q];
      }
    }
    if (bad) *errflag = 1;
  }
  if (wv == 3 && bad) *errflag = 1;
  if (wv == 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = lk + 4 * q;
      if (i < r && lr < r) st->V[i * r + lr] = Vm[q];
    }
  }
  if (tid < r) st->mu[tid] = s_mu[tid];
  __syncthreads();
  if (tid == 0) {
    st->k = b.k0 + b.nb;
    st->rho = rho; st->lam = lam; st->phi = phi; st->omega = omega; st->ee = ee_last;
    st->s_done = s_last; st->eta_done = eta_last; st->N_done = N_last;
    if (*errflag && st->err == 0) st->err = (int)(b.k0 + 1);
    st->ns_valid = 0;          // nothing the two-inversion kernels carry from block to block describes this state
  }
}

__global__ __launch_bounds__(WG) void psmf_blk_filter6(BlockParams b) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w == 0) f6_program<0, false>(b);
  else if (w == 1) f6_program<1, false>(b);
  else if (w == 2) f6_program<2, false>(b);
  else f6_program<3, false>(b);
}

// the random walk with Q = q I (PSMF_FILTER6_DUAL=1)
__global__ __launch_bounds__(WG) void psmf_blk_filter6d(BlockParams b) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w == 0) f6_program<0, true>(b);
  else if (w == 1) f6_program<1, true>(b);
  else if (w == 2) f6_program<2, true>(b);
  else f6_program<3, true>(b);
}

}  // namespace psmf
