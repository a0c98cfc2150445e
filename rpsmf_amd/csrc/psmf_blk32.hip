// psmf_blk_filter7: psmf_blk_filter6's design (psmf_blk16.hip: one program per wave, the r x r state in ONE wave's registers in the
// MFMA output layout, wave-local sweeps, stages that do not depend on each other side by side) for the ranks 17 <= r <= 32 --
// everything the general one-group kernel psmf_blk_filter<32> took there: dense-Jacobian dynamics (scaled walk, scaled sinusoid,
// Fourier basis), a general Q, and -- measured faster than filter4's sweep regime -- nothing else for now (psmf_capi.hip decides).
// A 32 x 32 matrix is 2 x 2 tiles of 16 x 16: tile (ti, tj), lane l, register q = element (16 ti + (l >> 4) + 4 q, 16 tj + (l & 15)).
//   dynamics forward (waves 1-3, psmf_dyn.hip; the matrix wave only joins the barriers)
//   A  wave 0: Pbar = F P F^T + Q -- T = P F^T and F T, 32 + 32 float64 MFMAs whose operands need no shuffling (a symmetric matrix in
//        the output layout IS the A operand: tile (tk, ti) serves A block (ti, tk); T in the output layout IS the B operand of F T) --
//        <G, Pbar>, eta (published), first sweep -Pbar^-1 (wave_sweep_tiles_m<2>, psmf_ns.hip: no LDS, no barrier)
//      wave 2: w = V mu_bar (16 MFMAs), s; with wave 0's eta: N, kappa (published)
//      wave 1 (lane = coefficient row): b = A mu_bar, K a, a, h = A^T K a, e'e, g_f                                       | barrier
//   B  wave 0: kappa G + Pbar^-1, augmented with kappa h in row / column r2 (r <= 30; else kappa P+ h as a product) -> second sweep:
//        P+, mu - mu_bar and 1 - kappa^2 h'P+h together; omega, phi; P, G, Q updates in registers;  wave 2: V;  wave 1: rank-1
//        updates of A and K A;  waves 1-3: gradsum += J_theta^T g_f                                                        | barrier
//   mu, in-loop Adam
// Same recursion and float64 arithmetic as psmf_blk_filter (summation orders differ).  PSMF_FILTER7=0: back to psmf_blk_filter<32>.
#pragma once
#include "psmf_blk16.hip"
#ifndef PSMF_F7_FWD
#define PSMF_F7_FWD 1
#endif

namespace psmf {

// dyn_forward (psmf_dyn.hip) for the dense trigonometric kinds (scaled sinusoid, Fourier basis with N <= 2) at 17 <= r <= 32, theta in
// LDS: what f6_dyn_forward (psmf_blk16.hip) is for r <= 16, on 32-wide index maps -- thread = (term, column) for the trig values,
// thread = (row group, column) for six elements of F, no integer division by the runtime r, the matrix elements and gains an element
// of F needs cached in registers for the block when theta cannot change inside it (no in-loop optimiser), TWO barriers instead of
// three (mu_bar_i: one thread per row walks the terms, no partial sums across threads).  The generic routine took 7 000 of a
// FourierBasis step's 31 700 cycles at r = 20 (tools/blk32_prof.hip): every sum waited for its LDS operands one by one and every
// element divided twice.  Same sums in the same order.  tid = 0 .. 191 (waves 1-3); the matrix wave keeps the barrier count (2).
struct F7Jac {
  double m[6][4], c[4];
};
__device__ __forceinline__ bool f7_fwd_ok(const StepParams& p) {
  if (!(p.dyn_kind == DYN_FOURIER || p.dyn_kind == DYN_SINUSOID) || !dyn_dense(p.dyn_kind, p.dyn_flags)) return false;
  const int nt = dyn_n_terms(p.dyn_kind, p.dyn_terms);
  if (nt < 1 || nt > 4 || p.r > 32) return false;
  for (int t = 0; t < nt; ++t)
    if (dyn_term(p.dyn_kind, p.dyn_flags, p.dyn_terms, p.r, t).m_off < 0) return false;      // (every term of these kinds has its matrix)
  return true;
}
__device__ __forceinline__ void f7_jac_cache(F7Jac& jc, const StepParams& p, const double* th, const int tid) {
  const int r = p.r, nt = dyn_n_terms(p.dyn_kind, p.dyn_terms), j = tid & 31, jc_ = min(j, r - 1);
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const DynTerm dt = dyn_term(p.dyn_kind, p.dyn_flags, p.dyn_terms, r, min(tt, nt - 1));
    jc.c[tt] = tt < nt ? (dt.c_off >= 0 ? th[dt.c_off + jc_] : 1.0) : 0.0;                   // a slot beyond nt adds nothing
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = (tid >> 5) + 6 * k;
      jc.m[k][tt] = (tt < nt && i < r && j < r) ? th[dt.m_off + i * r + j] : 0.0;
    }
  }
}
__device__ __forceinline__ void f7_dyn_forward(const StepParams& p, const double* th, const F7Jac& jc, const bool cached, const double tk,
                                               const double* s_x, double* s_mub, double* sF, const int ldf, double* s_val, double* s_tp,
                                               const int tid) {
  const int r = p.r, kind = p.dyn_kind, flags = p.dyn_flags, N = p.dyn_terms;
  const int nt = dyn_n_terms(kind, N);
  const int t = tid >> 5, j = tid & 31;
  if (t < nt && j < r) {
    const DynTerm d = dyn_term(kind, flags, N, r, t);
    const double c = d.c_off >= 0 ? th[d.c_off + j] : 1.0;
    double sn, cs;
    dyn_sincospi(2.0 * th[d.b_off + j] * tk + (c * s_x[j]) * 0.31830988618379067154, sn, cs);
    s_val[t * RM + j] = d.is_cos ? cs : sn;
    s_tp[t * RM + j] = d.is_cos ? -sn : cs;
  }
  __syncthreads();
  {     // F[i][j] = sum_t M_t[i][j] trig_t'(arg_tj) c_t[j]: six elements per thread, the four term slots straight-line
    const int jl = min(j, r - 1);
    double tp[4];
    if (cached) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) tp[tt] = s_tp[min(tt, nt - 1) * RM + jl] * jc.c[tt];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int i = (tid >> 5) + 6 * k;
        double a = 0.0;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) a += jc.m[k][tt] * tp[tt];
        if (i < r && j < r) sF[i * ldf + j] = a;
      }
    } else {            // theta moves inside the block (PSMFRecursive): the operands from LDS, all in flight before the first use
      int mo[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const DynTerm dt = dyn_term(kind, flags, N, r, min(tt, nt - 1));
        mo[tt] = dt.m_off;
        const double cj = dt.c_off >= 0 ? th[dt.c_off + jl] : 1.0;
        tp[tt] = tt < nt ? s_tp[tt * RM + jl] * cj : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int i = (tid >> 5) + 6 * k, il = min(i, r - 1);
        double mv[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) mv[tt] = th[mo[tt] + il * r + jl];
        double a = 0.0;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) a += mv[tt] * tp[tt];
        if (i < r && j < r) sF[i * ldf + j] = a;
      }
    }
  }
  if (tid < 64) {       // mu_bar_i = sum_t (M_t trig_t)_i: lane = (row i, half h); half h walks the terms h, h + 2 -- a lone wave pays per
                        // INSTRUCTION (one per 5-9 cycles), so the 32-long sums of two terms side by side, then one lane swap
    const int i = min(tid & 31, r - 1), hf = tid >> 5;
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tt = hf + 2 * u;
      if (tt < nt) {
        const DynTerm dt = dyn_term(kind, flags, N, r, tt);
        const double* row = th + dt.m_off + i * r;
        double mv[32], sv[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) { mv[q] = row[min(q, r - 1)]; sv[q] = s_val[tt * RM + q]; }      // s_val is zero beyond column r
        double at = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) at += mv[q] * sv[q];
        a += at;
      }
    }
    a = xor32_sum_f64(a);
    if (tid < r) s_mub[tid] = a;
  }
  __syncthreads();
}

template <int ROLE>
__device__ __forceinline__ void f7_program(const BlockParams& b) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const StepParams& p = b.sp;
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x;
  constexpr int wv = ROLE;
  const int lane = tid & 63, lk = lane >> 4, lr = lane & 15;
  const int r2 = r + (r & 1);
  const double dd = (double)p.d;
  // ---- LDS carve: psmf_blk_filter's (blk_filter_lds_bytes) ----
  double* sK = sm;                    // RB x RB
  double* sA = sK + RB * RB;          // RB x r, row stride RS
  double* sKA = sA + RB * RS;         // RB x r
  double* s_img = sKA + RB * RS;      // WG (unused here)
  double* s_mub = s_img + WG;         // RM each below
  double* s_f = s_mub + RM;
  double* s_w = s_f + RM;
  double* s_h = s_w + RM;
  double* s_munew = s_h + RM;
  double* s_mu = s_munew + RM;
  double* s_a = s_mu + RM;            // RB
  double* s_Ka = s_a + RB;            // RB
  double* s_sc = s_Ka + RB;           // 0 s, 1 eta, 2 N, 3 1 / N, 4 kappa, 5 lambda of the step, 6 e'e, 7 rho of the step
  int* s_flag = reinterpret_cast<int*>(s_sc + 16);     // steps whose w, s, N, 1 / N, kappa wave 2 has published
  int* s_flagA = s_flag + 1;                           // ... whose eta, lambda, rho wave 0 has published
  double* rowbuf = s_sc + 8 * RB;     // 4 * RM (unused here)
  double* s4 = rowbuf + 4 * RM;       // 4 (+ errflag)
  int* errflag = reinterpret_cast<int*>(s4 + 4);
  double* sF = s4 + 6;                // RM/2 x RS: dense Jacobian
  double* sPm = sF + (RM / 2) * RS;   // here: the four tiles of F P F^T + Q as images (4 x 256 doubles <= 32 RS), for the symmetrisation
  double* sT = sPm + (RM / 2) * RS;   // scratch of the dynamics
  double* s_val = sT + (RM / 2) * RS; // DYN_MAX_TERMS x RM
  double* s_tp = s_val + DYN_MAX_TERMS * RM;
  double* s_gf = s_tp + DYN_MAX_TERMS * RM;   // RM
  double* s_u = s_gf + RM;            // RM
  double* s_theta = s_u + RM;         // BLK_TH_CAP
  double* s_grad = s_theta + BLK_TH_CAP;
  const bool dense = dyn_dense(p.dyn_kind, p.dyn_flags);
  const bool th_lds = p.n_theta > 0 && p.n_theta <= BLK_TH_CAP;
  const bool has_bw = p.n_theta > 0 && p.dyn_kind != DYN_RANDOM_WALK;
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
  for (int idx = tid; idx < (RM / 2) * RS; idx += WG) sF[idx] = 0.0;        // wave 0 reads whole tiles: zero outside r x r
  for (int idx = tid; idx < DYN_MAX_TERMS * RM; idx += WG) { s_val[idx] = 0.0; s_tp[idx] = 0.0; }
  // the 32-wide forward pass of the dense trigonometric kinds (PSMF_F7_FWD=0 at build time: the generic dyn_forward)
  const bool f7fwd = PSMF_F7_FWD && th_lds && f7_fwd_ok(p);
  const int nbar_fwd = f7fwd ? 2 : f6_dyn_barriers(p, false);
  if (tid < r) s_mu[tid] = st->mu[tid];
  // ---- lane predicates as multipliers, per dimension: element (16 ti + lk + 4 q, 16 tj + lr) ----
  double frow[2][4], fcol[2], dgq[4], raug[2][4], caug[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    fcol[t] = (16 * t + lr < r) ? 1.0 : 0.0;
    caug[t] = 0.0;                  // (round 5: no augmentation -- the inversion by blocks, psmf_ns.hip, is a plain inverse and saves more than
                                    //  kappa P+ h as a product costs; the augmented sweep was  (16 * t + lr == r2) ? 1.0 : 0.0  here)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 16 * t + lk + 4 * q;
      frow[t][q] = i < r ? 1.0 : 0.0;
      raug[t][q] = 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) dgq[q] = (lk + 4 * q == lr) ? 1.0 : 0.0;   // diagonal position inside a diagonal tile
  // finq = frow * fcol;  fdg (ti == tj) = dgq * frow;  fpad (ti == tj) = dgq * (1 - frow)
  double Vm[2][2][4], Pm[2][2][4], Qm[2][2][4], Gm[2][2][4];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Vm[ti][tj][q] = Pm[ti][tj][q] = Qm[ti][tj][q] = Gm[ti][tj][q] = 0.0;
        const int i = 16 * ti + lk + 4 * q, c = 16 * tj + lr;
        const bool in = i < r && c < r;
        const int idx = in ? i * r + c : 0;
        if (wv == 0) {
          const double lq = st->Q[idx], lp = st->P[idx];
          Qm[ti][tj][q] = in ? lq : 0.0;
          Pm[ti][tj][q] = in ? lp : 0.0;
        }
        if (wv == 2) {
          const double lv = st->V[idx];
          Vm[ti][tj][q] = in ? lv : 0.0;
        }
      }
  const bool aug_fits = false;              // (r2 < 32 with the augmented sweep)
  Sw16K swk;
  if (wv == 0) sw16k_init(swk, lk, lr);
  double rho = st->rho, lam = st->lam;
  bool bad = false;
  __syncthreads();
  F7Jac jcache;
  if (wv != 0 && f7fwd && !p.recursive) f7_jac_cache(jcache, p, s_theta, tid - 64);      // (s_theta is complete behind the barrier above)
  // A_0 = [I; 0], K A_0 = first r columns of K, G_0 = K[0:r, 0:r]; columns r .. RS - 1 zero
  for (int idx = tid; idx < RB * RS; idx += WG) {
    const int m = idx / RS, c = idx - m * RS;
    sA[idx] = (m == c && c < r) ? 1.0 : 0.0;
    sKA[idx] = c < r ? sK[m * RB + c] : 0.0;
  }
  if (wv == 0) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 16 * ti + lk + 4 * q, c = 16 * tj + lr;
          Gm[ti][tj][q] = (i < r && c < r) ? sK[i * RB + c] : 0.0;
        }
  }
  __syncthreads();

  double s_last = 0.0, eta_last = 0.0, N_last = 0.0, phi = 1.0, omega = 1.0, ee_last = 0.0;
  BLK_T0();
  for (int jb = 0; jb < b.nb; ++jb) {
    const long long kstep = b.k0 + jb + 1;   // 1-based step index
    // ---- mu_bar = f(theta, mu, k), F = df/dx (psmf.py:104-115; psmf_dyn.hip): waves 1-3; the matrix wave joins the barriers ----
    if (wv == 0) { for (int q = 0; q < nbar_fwd; ++q) __syncthreads(); }
    else if (f7fwd) f7_dyn_forward(p, s_theta, jcache, !p.recursive, (double)kstep, s_mu, s_mub, sF, RS, s_val, s_tp, tid - 64);
    else dyn_forward<WG - 64>(pd, (double)kstep, s_mu, s_mub, s_f, sF, RS, s_val, s_tp, sT, tid - 64);
    BLK_T(0);
    const double qs = p.q_sched ? p.q_sched[kstep - p.series_t0] : 1.0;
    if (p.rho_sched) rho = p.rho_sched[kstep - p.series_t0];
    double A[2][2][4], Pb[2][2][4], wrow[2][4], mb[2][4];
    double s = 0.0, eta = rho, N = 1.0, invN = 1.0, kappa = 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) { wrow[t][q] = 0.0; mb[t][q] = 0.0; }
    if (wv == 0) {
      // ================= phase A, matrix wave =================
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) mb[t][q] = s_mub[16 * t + lk + 4 * q];     // mu_bar of this lane's rows
      if (p.pbar_predict) {
        if (dense) {
          // Pbar = F P F^T + Q.  Fr[a][b][q] = F[16 a + lr][16 b + lk + 4 q]: B operand of (F^T)(b, a) and A operand of F(a, b)
          double Fr[2][2][4];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) Fr[a][c][q] = sF[(16 * a + lr) * RS + 16 * c + lk + 4 * q];
          f64x4 T[2][2];
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
              f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (16 * tk + 4 * q < r)            // (uniform) inner indices beyond r are zero padding: r = 20 runs 5 of the 8 k-steps
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pm[tk][ti][q], Fr[tj][tk][q], acc, 0, 0, 0);
              T[ti][tj] = acc;                                        // (P F^T)(ti, tj)
            }
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
              f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (16 * tk + 4 * q < r)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Fr[ti][tk][q], T[tk][tj][q], acc, 0, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                Pb[ti][tj][q] = fma(qs, Qm[ti][tj][q], acc[q]);
                sPm[(ti * 2 + tj) * 256 + q * 64 + lane] = Pb[ti][tj][q];
              }
            }
          // symmetrise through the tile images: (X + X^T) / 2, bitwise symmetric.  Element (i, c) of tile (ti, tj) has its transpose
          // in tile (tj, ti) at register lr >> 2, lane (lr & 3) * 16 + lk + 4 q
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                Pb[ti][tj][q] = 0.5 * (Pb[ti][tj][q] + sPm[(tj * 2 + ti) * 256 + (lr >> 2) * 64 + (lr & 3) * 16 + lk + 4 * q]);
        } else {
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
              const double fc = s_f[16 * tj + lr];
#pragma unroll
              for (int q = 0; q < 4; ++q) Pb[ti][tj][q] = fma(s_f[16 * ti + lk + 4 * q] * Pm[ti][tj][q], fc, qs * Qm[ti][tj][q]);
            }
        }
      } else {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) Pb[ti][tj][q] = Pm[ti][tj][q];
      }
      if (p.eta_full) {
        double tr = 0.0;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) tr = fma(Gm[ti][tj][q], Pb[ti][tj][q], tr);
        eta += wave_sum_f64_dpp(tr) / dd;
      }
      if (lane == 0) { s_sc[1] = eta; s_sc[5] = lam; s_sc[7] = rho; }
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(s_flagA, jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      BLK_T(1);
      if (p.coef_update) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) A[ti][tj][q] = Pb[ti][tj][q] + (ti == tj ? dgq[q] * (1.0 - frow[ti][q]) : 0.0);
        wave_invert_tiles<2>(A, r2, swk, bad);               // -Pbar^-1 (by blocks: psmf_ns.hip)
      }
    } else if (wv == 2) {
      // ================= phase A, V wave: w = V mu_bar, s; N, kappa when wave 0's eta is there =================
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) mb[t][q] = s_mub[16 * t + lk + 4 * q];       // = the B operand of V mu_bar (rows 16 t + 4 q + lk)
      double sp = 0.0;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        f64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vm[0][ti][q], mb[0][q], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vm[1][ti][q], mb[1][q], a1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { wrow[ti][q] = a0[q] + a1[q]; sp = fma(mb[ti][q], wrow[ti][q], sp); }     // w_i, i = 16 ti + lk + 4 q, in every column
      }
      s = xor32_sum_f64(xor16_sum_f64(sp));
      if (lr == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) s_w[16 * t + lk + 4 * q] = wrow[t][q];
      }
      while (__hip_atomic_load(s_flagA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < jb + 1) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      N = s + s_sc[1];
      invN = fast_rcp(N);
      kappa = fast_rcp(s_sc[7] + s);
      if (lane == 0) { s_sc[0] = s; s_sc[2] = N; s_sc[3] = invN; s_sc[4] = kappa; }
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(s_flag, jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (wv == 1) {
      // ================= phase A, coefficient wave: lane = coefficient row =================
      const int m = lane;
      double pb = 0.0, pk = 0.0;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {              // two halves of 16 columns (zero beyond r)
        double av[16], kv[16], mv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { av[c] = sA[m * RS + 16 * hb + c]; kv[c] = sKA[m * RS + 16 * hb + c]; mv[c] = s_mub[16 * hb + c]; }
#pragma unroll
        for (int c = 0; c < 16; ++c) { pb = fma(av[c], mv[c], pb); pk = fma(kv[c], mv[c], pk); }
      }
      const double am = (m == r + jb ? 1.0 : 0.0) - pb;
      const double kam = sK[m * RB + r + jb] - pk;
      s_a[m] = am;
      s_Ka[m] = kam;
      b.Bcoef[(size_t)jb * RB + m] = pb;
      // h = A^T Ka: columns j = lr and 16 + lr, the sixteen rows 16 lk .. 16 lk + 15 per lane, then across the four lane rows
      double ph0 = 0.0, ph1 = 0.0;
      {
        double a0[16], a1[16], kv[16];
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) { a0[mm] = sA[(16 * lk + mm) * RS + lr]; a1[mm] = sA[(16 * lk + mm) * RS + 16 + lr]; kv[mm] = s_Ka[16 * lk + mm]; }
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) { ph0 = fma(a0[mm], kv[mm], ph0); ph1 = fma(a1[mm], kv[mm], ph1); }
      }
      const double hj0 = xor32_sum_f64(xor16_sum_f64(ph0)), hj1 = xor32_sum_f64(xor16_sum_f64(ph1));     // columns lr, 16 + lr (every lane row)
      if (lane < 16) { s_h[lane] = hj0; s_h[16 + lane] = hj1; }
      const double ee1 = wave_sum_f64_dpp(am * kam);
      if (lane == 0) s_sc[6] = ee1;
      // theta gradient at the pre-update state (psmf.py:57-64, rpsmf.py:62-71): lane j < r holds h_j
      if (has_bw) {
        while (__hip_atomic_load(s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < jb + 1) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (lane < r) {
          const double hj = lane < 16 ? hj0 : hj1;           // lanes 16 .. 31: lr = lane - 16, column 16 + lr
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
      double hrow[2][4], hcol[2], wcol[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        hcol[t] = s_h[16 * t + lr];
        wcol[t] = s_w[16 * t + lr];
#pragma unroll
        for (int q = 0; q < 4; ++q) { hrow[t][q] = s_h[16 * t + lk + 4 * q]; wrow[t][q] = s_w[16 * t + lk + 4 * q]; }
      }
      double Pp[2][2][4], quad = kappa * ee;
      if (p.coef_update) {
        // M = Pbar^-1 + kappa G (A holds -Pbar^-1 inside, garbage-free padding: rebuilt), augmented with kappa h in row / column r2
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double fin = frow[ti][q] * fcol[tj];
              const double pad = ti == tj ? dgq[q] * (1.0 - frow[ti][q]) : 0.0;
              A[ti][tj][q] = fma(kappa, Gm[ti][tj][q], pad - fin * A[ti][tj][q]) +
                             (caug[tj] * frow[ti][q] * (kappa * hrow[ti][q]) + raug[ti][q] * fcol[tj] * (kappa * hcol[tj]));
            }
        wave_invert_tiles<2>(A, r2, swk, bad);               // -P+ (kappa P+ h as a product below)
        BLK_T(4);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) Pp[ti][tj][q] = -(frow[ti][q] * fcol[tj]) * A[ti][tj][q];
        if (aug_fits) {
          // element (r2, r2): tile (r2 >> 4, r2 >> 4), register (r2 & 15) >> 2, lane ((r2 & 3) << 4) | (r2 & 15)
          const int tq = (r2 & 15) >> 2, ln = ((r2 & 3) << 4) | (r2 & 15);
          double a_c;
          if (r2 < 16) a_c = tq == 0 ? A[0][0][0] : (tq == 1 ? A[0][0][1] : (tq == 2 ? A[0][0][2] : A[0][0][3]));
          else a_c = tq == 0 ? A[1][1][0] : (tq == 1 ? A[1][1][1] : (tq == 2 ? A[1][1][2] : A[1][1][3]));
          quad += readlane_f64(a_c, ln) - 1.0;                 // kappa e'e - kappa^2 h'P+h  (psmf.py:155-165)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
            if (16 * tj + lr == r2) {                          // the lanes of column r2 hold kappa P+ h of their rows
#pragma unroll
              for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (16 * ti + lk + 4 * q < r) s_munew[16 * ti + lk + 4 * q] = A[ti][tj][q] + mb[ti][q];      // mu = mu_bar + kappa P+ h
            }
        } else {
          // r = 31, 32: no column left for the augmentation -- kappa P+ h as a product (the swept matrix, symmetric, is its own A
          // operand: tile (tk, ti) serves block (ti, tk); kappa h of the lane's rows in every column the B operand)
          double kh[2][4];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) kh[t][q] = kappa * hrow[t][q];
          double part = 0.0;
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            f64x4 z0 = {0.0, 0.0, 0.0, 0.0}, z1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0][ti][q], kh[0][q], z0, 0, 0, 0);
              z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[1][ti][q], kh[1][q], z1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double dz = -frow[ti][q] * (z0[q] + z1[q]);          // (kappa P+ h)_i, i = 16 ti + lk + 4 q
              part = fma(kh[ti][q], dz, part);
              if (lr == 0 && 16 * ti + lk + 4 * q < r) s_munew[16 * ti + lk + 4 * q] = mb[ti][q] + dz;
            }
          }
          quad -= xor32_sum_f64(xor16_sum_f64(part));
        }
      } else {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) Pp[ti][tj][q] = Pb[ti][tj][q];
        if (lr == 0) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (16 * t + lk + 4 * q < r) s_munew[16 * t + lk + 4 * q] = mb[t][q];
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
      // P, G, Q of the step (psmf.py:150-170; G: the tracked Gram of C)
      const double ew = ee * invN;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const double wj = wcol[tj] * invN, hjn = hcol[tj] * invN;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Pm[ti][tj][q] = pscale * Pp[ti][tj][q];
            Gm[ti][tj][q] += (frow[ti][q] * fcol[tj]) * (fma(hrow[ti][q], wj, wrow[ti][q] * hjn) + ew * (wrow[ti][q] * wj));
            Qm[ti][tj][q] *= qscale;
          }
        }
      s_last = s; eta_last = eta; N_last = N; ee_last = ee;
      BLK_T(5);
      if (has_bw) __syncthreads();                       // (the barrier that ends dyn_backward on the other waves)
    } else {
      // ================= phase B, wave 2: V of the step (psmf.py:166-170, rpsmf.py: phi) =================
      if (wv == 2) {
        double vscale = 1.0;
        if (p.robust) { const double lm = s_sc[5]; vscale = p.alpha * ((lm + s_sc[6] * invN) * fast_rcp(lm + dd)); }
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const double wc = s_w[16 * tj + lr];
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int q = 0; q < 4; ++q) Vm[ti][tj][q] = vscale * fma(-(wrow[ti][q] * wc), invN, Vm[ti][tj][q]);      // (w_i w_j first: bitwise symmetric)
        }
      }
      // ================= phase B, wave 1: rank-1 updates of the coefficient matrices (lane = row) =================
      if (wv == 1) {
        const int m = lane;
        const double iN = s_sc[3];
        const double am = s_a[m] * iN, km = s_Ka[m] * iN;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          double av[16], kv[16], wc[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) { av[c] = sA[m * RS + 16 * hb + c]; kv[c] = sKA[m * RS + 16 * hb + c]; wc[c] = s_w[16 * hb + c]; }     // (w is zero beyond r)
#pragma unroll
          for (int c = 0; c < 16; ++c) { sA[m * RS + 16 * hb + c] = fma(am, wc[c], av[c]); sKA[m * RS + 16 * hb + c] = fma(km, wc[c], kv[c]); }
        }
      }
      // ================= phase B, waves 1-3: gradsum += J_theta^T g_f =================
      if (has_bw) dyn_backward<WG - 64>(pd, (double)kstep, s_mu, s_gf, s_val, s_tp, tid - 64);      // ends with a barrier
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
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * ti + lk + 4 * q, c = 16 * tj + lr;
        if (i < r && c < r) {
          const int idx = i * r + c;
          if (wv == 0) { st->P[idx] = Pm[ti][tj][q]; st->Q[idx] = Qm[ti][tj][q]; st->G[idx] = Gm[ti][tj][q]; }
          if (wv == 2) st->V[idx] = Vm[ti][tj][q];
        }
      }
  if (wv == 0 && bad) *errflag = 1;
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

__global__ __launch_bounds__(WG) void psmf_blk_filter7(BlockParams b) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w == 0) f7_program<0>(b);
  else if (w == 1) f7_program<1>(b);
  else if (w == 2) f7_program<2>(b);
  else f7_program<3>(b);
}

}  // namespace psmf
