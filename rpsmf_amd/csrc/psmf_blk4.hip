// Role-specialised coefficient-space filter of the blocked engine for state transitions with a DIAGONAL Jacobian
// ("filter4"; included from psmf_blk3.hip, whose helpers, LDS layout and T- / P-layouts it shares):
//
//     random walk with per-step R_k / Q_k schedules          psmf.py:115,123,141
//     f = cos(2 pi theta t + x)                               ExperimentSynthetic/synthetic_psmf.py:105-106 (full filter)
//     f = sin(2 pi b t + [c o] x)   (Sinusoid, unscaled)      nonlinearities.py:81-114
//
// with Q = q I, full filter, r <= 32, PSMF and rPSMF, the theta gradient (psmf.py:167-177, rpsmf.py:173-184) and the in-loop
// Adam step of the recursive classes (psmf.py:287-304) -- everything psmf_blk_filter (psmf_block.hip) does for these kinds at
// 14.7 us per timestep (two LDS-and-barrier sweep inversions, a dozen workgroup barriers, theta in global memory).
//
// Why not filter3's two PARALLEL inversions: Pbar_{k+1} = F_{k+1} (beta omega P+_k) F_{k+1} + q I with F_{k+1} = F(mu_k), and mu_k
// needs P+_k -- the Woodbury trick that made W_k = (M_k / beta + I / q)^-1 independent of P+_k needs F = I.  So the two inversions
// of a step run one after the other, each as a Newton-Schulz refinement on the matrix cores with a start that needs no history:
//
//   Y pair (waves 2-3), phase 0, beside wave 4's  w = V mu_bar, s, kappa:
//       Lbar_k = Pbar_k^-1,  Pbar = q I + E,  E = pscale F P+ F  (elementwise from the X pair's columns).  In the filter's
//       working regime ||E|| << q (P+ ~ (kappa G)^-1), so the first Newton-Schulz step from I / q is free,
//       X_0 = (2 I - Pbar / q) / q  with residual (E / q)^2, and ONE real iteration finishes (f32 storage; two for f64).
//   X pair (waves 0-1), phase 1:  M = Lbar + kappa G,  P+ = M^-1 from filter3's start predictor (rank-2 downdate of the previous
//       inverse rescaled by the kappa ratio), one iteration.
//
// Either falls back to the direct symmetric sweep on all 8 waves when its start is too far (transients: P ~ q).
// The vector waves 4-7 are filter3's (V and the scalars; A by rows; KA by rows; A^T by columns); wave 4 also owns the r-sized
// theta path (g_f, gradient sums, Adam), mu_bar_{k+1} = f(theta, mu_k, k + 1) and F_{k+1} are formed by the lanes of the X pair
// that have just formed mu_k.  Three workgroup barriers per timestep in the steady state, as in filter3.

struct F4Lds {
  double* fd;       // RM: diagonal of F of the current step (0 beyond r)
  double* mu;       // RM: mu_{k-1}
  double* tp;       // RM: trig'(arg) of the current step (gradient)
  double* th;       // 2 RM: frequencies b (theta for cos-phase) | gains c
  double* rs;       // 48: rho_k of the block's steps (schedule) -- valid iff p.rho_sched
  double* qs;       // 48: q_k multipliers -- valid iff p.q_sched
  double* kc;       // F4_NKC: constants of the trig polynomials (f4_fill_trig_constants)
};

#define F4_DECIDE(base_, par_, it_, done_, failed_, last_)                                                  \
  do {                                                                                                     \
    const double w_ = L.nrm[(par_) * 4 + (base_)] + L.nrm[(par_) * 4 + (base_) + 1];                       \
    ++ctl.c_it;                                                                                            \
    if (w_ < L.nrm[8]) done_ = true;              /* ||R|| below the tolerance BEFORE the update just made */ \
    else if (!(w_ < L.nrm[9]) || (it_) == F3_MAXIT - 1) failed_ = true;                                    \
    else last_ = w_ * w_ < L.nrm[10];             /* one more iteration is the last: no check needed */     \
  } while (0)

// The direct symmetric sweep of ONE 32 x 32 image (in place, A <- A^-1).  Round 4: by ONE vector wave (wave 4), wave-local on the
// matrix cores (wave_sweep_tiles, psmf_ns.hip: the image as 2 x 2 tiles of 16 x 16 in its registers, no LDS exchange, no barrier
// inside) -- where P ~ q both inversions of every step end up here, and a pivot round of the LDS-and-barrier sweep that the four
// vector waves ran before cost 1 150 cycles against ~450.  (PSMF_F4_WAVE_SWEEP=0 at build time: that sweep, kept for A/B runs.)
// Called by all 512 threads at the same point; the image was published before the barrier that precedes it.
#ifndef PSMF_F4_WAVE_SWEEP
#define PSMF_F4_WAVE_SWEEP 1
#endif
__device__ __forceinline__ void f4_sweep_image(const F3Lds& L, double* im, const int r2, const int tid) {
#if PSMF_F4_WAVE_SWEEP
  if ((tid >> 6) == 4) {
    const int lane = tid & 63, lk = lane >> 4, lr = lane & 15;
    bool bad = false;
    if (r2 <= 16) {
      double A[1][1][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lk + 4 * q;
        A[0][0][q] = (i < r2 && lr < r2) ? im[i * F3_S + lr] : (i == lr ? 1.0 : 0.0);
      }
      Sw16K swk;
      sw16k_init(swk, lk, lr);
      wave_sweep_tiles_m<1>(A, r2, swk, bad);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lk + 4 * q;
        if (i < r2 && lr < r2) im[i * F3_S + lr] = -A[0][0][q];
      }
    } else {
      double A[2][2][4];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 16 * ti + lk + 4 * q, c = 16 * tj + lr;
            A[ti][tj][q] = (i < r2 && c < r2) ? im[i * F3_S + c] : (i == c ? 1.0 : 0.0);
          }
      Sw16K swk;
      sw16k_init(swk, lk, lr);
      wave_invert_tiles<2>(A, r2, swk, bad);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 16 * ti + lk + 4 * q, c = 16 * tj + lr;
            if (i < r2 && c < r2) im[i * F3_S + c] = -A[ti][tj][q];
          }
    }
    if (__any((int)bad) && lane == 0) *L.errflag = 1;
  }
#else
  if (tid >= WG) {
    const int lt = tid - WG, c32 = lt & 31, rg = lt >> 5;
    double A1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) A1[m] = im[(rg + 8 * m) * F3_S + c32];
    sweep_all<32>(A1, r2, c32, rg, L.rowbufX, L.errflag);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = rg + 8 * m;
      if (i < r2 && c32 < r2) im[i * F3_S + c32] = -A1[m];
    }
  } else {
    __syncthreads();                                   // sweep_all: one barrier before the pivot loop, one per 2 x 2 pivot
    for (int kk = 0; kk < r2; kk += 2) __syncthreads();
  }
#endif
  __syncthreads();
}

// x_skip / y_skip: steps for which the iteration is not even tried after a failure; x_back / y_back: the length of the next
// such pause -- doubled with every failure in a row (3, 6, ... 48), back to 3 with the first success -- so that in a regime in
// which a start is useless (P ~ q: Pbar has no small parameter) the kernel degrades to the direct sweeps, not to both
struct F4Ctl {
  bool have_prev;
  int x_skip, y_skip;
  int c_ns, c_sw, c_it, c_fail;
  int x_back, y_back;
};
#define F4_FAILED(skip_, back_) do { skip_ = back_; back_ = back_ < 48 ? 2 * back_ : 48; } while (0)

// Coefficients of the trig polynomials below, kept in LDS (F4Lds.kc; filled by f4_fill_trig_constants at block start).  As
// literals the compiler hoists the twelve 64-bit constants out of the time loop, where an inversion wave has no registers to
// spare: they were spilled and came back one dependent scratch load at a time (~2 000 cycles per timestep, seen in the ISA).
// LDS reads are issued together and cost one round trip.
constexpr int F4_NKC = 16;
__device__ __forceinline__ void f4_fill_trig_constants(double* kc, const int i) {
  // [0..5] sin: S6 .. S1, [6..11] cos: C6 .. C1 (fdlibm __kernel_sin / __kernel_cos), [12] pi, [13] 1 / pi
  const double v[F4_NKC] = {1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,
                            8.33333333332248946124e-03, -1.66666666666666324348e-01,
                            -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05,
                            -1.38888888888741095749e-03, 4.16666666666666019037e-02,
                            3.14159265358979323846, 0.31830988618379067154, 0.0, 0.0};
  double x = 0.0;
#pragma unroll
  for (int q = 0; q < F4_NKC; ++q) x = (i == q) ? v[q] : x;
  kc[i] = x;
}

// sin(pi a), cos(pi a) in ~40 instructions: exact reduction to |f| <= 1/4 half-turns (a - n / 2 is exact in float64), the
// fdlibm kernel polynomials on |pi f| <= pi / 4, quadrant rotation.  (sincospi() of the device library cost this call site --
// lanes of an inversion wave with ~200 live registers -- 4 000 cycles per timestep in spills and branches.)
__device__ __forceinline__ void f4_sincospi(const double* kc, const double a, double& sn, double& cs) {
  const double n = rint(2.0 * a);
  const double f = fma(-0.5, n, a);
  const double x = f * kc[12], z = x * x;
  double ps = kc[0], pc = kc[6];
#pragma unroll
  for (int q = 1; q < 6; ++q) { ps = fma(z, ps, kc[q]); pc = fma(z, pc, kc[6 + q]); }
  const double s0 = fma(x * z, ps, x);                       // sin x = x + x^3 (S1 + z (S2 + ... S6))
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));       // cos x = 1 - z / 2 + z^2 (C1 + z (C2 + ... C6))
  const int qd = (int)((long long)n) & 3;
  const double s1 = (qd & 1) ? c0 : s0, c1 = (qd & 1) ? s0 : c0;
  sn = (qd == 2 || qd == 3) ? -s1 : s1;
  cs = (qd == 1 || qd == 2) ? -c1 : c1;
}

// mu_bar, diag F, trig' for step t from the posterior mean x of the step before (psmf_dyn.hip: same expressions)
__device__ __forceinline__ void f4_dyn_eval(const StepParams& p, const F4Lds& D, const int j, const double t, const double x,
                                            double& mub, double& fd, double& tp) {
  if (p.dyn_kind == DYN_RANDOM_WALK) { mub = x; fd = 1.0; tp = 0.0; return; }
  const bool phased = p.dyn_kind == DYN_SINUSOID && (p.dyn_flags & 2);
  const double c = phased ? D.th[RM + j] : 1.0;
  // sin / cos of 2 pi b t + c x from the angle in half-turns (exact, short range reduction); agrees with sincos() of the radian
  // argument to ~1e-12 at |arg| ~ 1e4, which is that argument's own rounding
  const double arg = 2.0 * D.th[j] * t + (c * x) * D.kc[13];
  double sn, cs;
  f4_sincospi(D.kc, arg, sn, cs);
  if (p.dyn_kind == DYN_COS_PHASE) { mub = cs; tp = -sn; fd = -sn; }
  else { mub = sn; tp = cs; fd = cs * c; }
}

// ------------------------------------------------------------------------------------------------------------
// The uniform control of a step, identical in the three programs (same LDS values -> same decisions -> same barriers).
// Y_WORK(par): the Y pair fetches its partner's column of parity `par` and iterates into parity par ^ 1; others: nothing.
// Y_TO_IMAGE(): Y pair: Pbar (own column) into image Y; X pair: identity (own column) into image X; others: nothing.
// ------------------------------------------------------------------------------------------------------------
#define F4_Y_CONTROL()                                                                                     \
  bool ydone = false, yfailed = !try_y, ylast = false;                                                     \
  int ypar = 0, yit = 0;                                                                                   \
  if (try_y) {                                                                                             \
    F4_DECIDE(2, 0, yit, ydone, yfailed, ylast);                                                           \
    while (!ydone && !yfailed) {                                                                           \
      Y_WORK(ypar);                                                                                        \
      ypar ^= 1;                                                                                           \
      ++yit;                                                                                               \
      f3_barrier();                                                                                        \
      if (ylast) { ydone = true; ++ctl.c_it; break; }                                                      \
      F4_DECIDE(2, ypar, yit, ydone, yfailed, ylast);                                                      \
    }                                                                                                      \
  }                                                                                                        \
  if (!ydone) {                                                                                            \
    if (try_y) { ++ctl.c_fail; F4_FAILED(ctl.y_skip, ctl.y_back); }                                        \
    Y_TO_IMAGE();                                                                                          \
    f3_barrier();                                                                                          \
    f4_sweep_image(L, L.img + 32 * F3_S, r2, tid);                                                         \
  }                                                                                                        \
  else ctl.y_back = 3;                                                                                     \
  const bool y_img = !ydone;

// ------------------------------------------------------------------------------------------------------------
// X pair: P+ = (Lbar + kappa G)^-1, mu, the next step's mu_bar and F.  C = own tile column.
// ------------------------------------------------------------------------------------------------------------
template <int C, int FULL>
__device__ __forceinline__ void f4_x_program(const BlockParams& b, const F3Blk& k, const F3Lds& L, const F4Lds& D, const int role, const int lane,
                                             const bool carried, const bool warm) {
  const StepParams& p = b.sp;
  DevState* st = p.st;
  const int r = p.r, r2 = r + (r & 1), tid = 64 * role + lane;
  const int lrow = lane >> 4, lcol = lane & 15;
  constexpr int inv = 0;
  double* imgX = L.img;
  double* imgY = L.img + 32 * F3_S;
  double G[16], Xc[8];
  float Xa[16];
  const int pcol = (lcol >> 2) + 4 * (lcol & 3);
  F3Mask<FULL> mk;
  const int rt = FULL == 2 ? r : r - 16;
  mk.c1 = lcol < rt;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) { mk.r1[qq] = lrow + 4 * qq < rt; mk.dg[qq] = lcol == lrow + 4 * qq; }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int row = 16 * ti + lrow + 4 * qq, col = 16 * tj + lcol;
        G[(ti * 2 + tj) * 4 + qq] = (row < r && col < r) ? L.sK[row * RB + col] : 0.0;
      }
  // the iterate the block starts from: the carried register dump, or P_{k0} itself (row-major; then pscale = 1 and the first
  // inversion is a direct sweep: P is the right INPUT of the step, not a start for M^-1)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = 16 * (e >> 2) + lrow + 4 * (e & 3), col = 16 * C + lcol;
    const bool in = row < r && col < r;
    const double pv = st->P[in ? row * r + col : 0];
    Xc[e] = carried ? st->f3_Xc[role][e * 64 + lane] : (in ? pv : (row == col ? 1.0 : 0.0));
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = 16 * ((e >> 2) & 1) + lrow + 4 * (e & 3), cp = 16 * (e >> 3) + pcol;
    const bool in = row < r && cp < r;
    const double pv = st->P[in ? row * r + cp : 0];
    Xa[e] = carried ? st->f3_Xa[role][e * 64 + lane] : (in ? (float)pv : (row == cp ? 1.f : 0.f));
  }
  {   // publish the own column (parity 0): the Y pair builds Pbar_1 from it
    f64x2* dp_ = reinterpret_cast<f64x2*>(L.dump) + (size_t)((0 * 4 + role) * 4) * 64 + lane;
#pragma unroll
    for (int e_ = 0; e_ < 4; ++e_) dp_[e_ * 64] = f64x2{Xc[2 * e_], Xc[2 * e_ + 1]};
    f32x4s* pp_ = reinterpret_cast<f32x4s*>(L.dumpP) + (size_t)((0 * 4 + role) * 2) * 64 + 16 * lrow + pcol;
#pragma unroll
    for (int t_ = 0; t_ < 2; ++t_)
      pp_[t_ * 64] = f32x4s{(float)Xc[t_ * 4], (float)Xc[t_ * 4 + 1], (float)Xc[t_ * 4 + 2], (float)Xc[t_ * 4 + 3]};
  }
  const double kap_warm = L.sc[F3_KAPPA];          // (wave 4 rewrites it in phase 0, after the init barrier)
  {   // mu_bar, F of the block's first step
    const int j = 16 * C + lcol;
    if (lrow == 0 && j < r) {
      double mb, fd, tp;
      f4_dyn_eval(p, D, j, (double)(k.k0 + 1), D.mu[j], mb, fd, tp);
      L.mub[j] = mb; D.fd[j] = fd; D.tp[j] = tp;
    }
  }
  f3_barrier();                                                       // ---- init barrier

#define F4_ITERATE(parity_out)                                                                             \
  do {                                                                                                     \
    const double nr_ = f3_ns_iter<C, FULL>(Mf, Xc, Xa, Xn, mk);                                            \
    const float nw_ = wave_sum_f32_dpp((float)nr_);                                                        \
    f64x2* dp_ = reinterpret_cast<f64x2*>(L.dump) + (size_t)(((parity_out) * 4 + role) * 4) * 64 + lane;   \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) dp_[e_ * 64] = f64x2{Xn[2 * e_], Xn[2 * e_ + 1]};    \
    f32x4s* pp_ = reinterpret_cast<f32x4s*>(L.dumpP) + (size_t)(((parity_out) * 4 + role) * 2) * 64 + 16 * lrow + pcol; \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                                       \
      pp_[t_ * 64] = f32x4s{(float)Xn[t_ * 4], (float)Xn[t_ * 4 + 1], (float)Xn[t_ * 4 + 2], (float)Xn[t_ * 4 + 3]}; \
    if (lane == 0) L.nrm[(parity_out) * 4 + role] = (double)nw_;                                           \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) Xc[e_] = Xn[e_];                                      \
  } while (0)
#define F4_FETCH_PARTNER(parity_in)                                                                        \
  do {                                                                                                     \
    _Pragma("unroll") for (int w_ = 0; w_ < 2; ++w_) {                                                     \
      const int src_ = w_ == 0 ? role : (role ^ 1);                                                        \
      const int to_ = w_ == 0 ? C : 1 - C;                                                                 \
      const f32x4s* pq_ = reinterpret_cast<const f32x4s*>(L.dumpP) + (size_t)(((parity_in) * 4 + src_) * 2) * 64 + lane; \
      _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) {                                                   \
        const f32x4s v_ = pq_[t_ * 64];                                                                    \
        Xa[to_ * 8 + t_ * 4 + 0] = v_[0]; Xa[to_ * 8 + t_ * 4 + 1] = v_[1];                                 \
        Xa[to_ * 8 + t_ * 4 + 2] = v_[2]; Xa[to_ * 8 + t_ * 4 + 3] = v_[3];                                 \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
#define Y_WORK(par_) do { } while (0)
#define Y_TO_IMAGE() do { } while (0)

  F4Ctl ctl = {carried, 0, 0, 0, 0, 0, 0, 3, 3};
  int w_par = 0;                                            // parity of the dump that holds the pair's latest columns
  bool fetch_late = false;
  // warm: the previous block of this launch left the start predictor's inputs in LDS (a = Z h, b = Z w, h, w, ee, N, kappa of
  // its last step): the first step of this block starts like any other (a plain start leaves ||I - M Z|| ~ 1 where the
  // innovations are large, and the block would open with a direct sweep)
  double kap_prev = warm ? kap_warm : 1.0;
  bool smw_ok = warm;
  BLK_T0();
  for (int jb = 0; jb < k.nb; ++jb) {
    // =============================== phase 0: <G, F P+ F>, tr G of the own column (eta of this step) ===============================
    if (fetch_late) F4_FETCH_PARTNER(w_par);
    {
      const double fc = D.fd[16 * C + lcol];
      double g1 = 0.0, t1 = 0.0;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const double fr = D.fd[16 * ti + lrow + 4 * qq];
          g1 += G[(ti * 2 + C) * 4 + qq] * Xc[ti * 4 + qq] * (fr * fc);          // G is zero outside r x r
          if (ti == C) t1 += F3_DIAG(mk, C, C, qq) ? G[(C * 2 + C) * 4 + qq] : 0.0;
        }
      g1 = wave_sum_f64_dpp(g1);
      t1 = wave_sum_f64_dpp(t1);
      if (lane == 0) { L.gp[C] = g1; L.tr[C] = t1; }
    }
    BLK_T(7);
    f3_barrier();                                                     // ---- B1
    BLK_T(1);
    const bool try_y = p.use_ns && ctl.y_skip == 0;
    if (!try_y && ctl.y_skip > 0) --ctl.y_skip;
#ifdef F4_DEBUG
    if (role == 0 && lane == 0) { st->GR[jb * 8 + 0] = L.nrm[2] + L.nrm[3]; st->GR[jb * 8 + 1] = try_y ? 1.0 : 0.0; }
#endif
    F4_Y_CONTROL();
#ifdef F4_DEBUG
    if (role == 0 && lane == 0) { st->GR[jb * 8 + 2] = ydone ? 1.0 : 0.0; st->GR[jb * 8 + 3] = (double)yit; }
#endif
    BLK_T(6);
    // =============================== phase 1: M = Lbar + kappa G, first iteration ===============================
    const bool try_ns = ctl.have_prev && p.use_ns && ctl.x_skip == 0;
    if (!try_ns && ctl.x_skip > 0) --ctl.x_skip;
    double Mf[16], Xn[8];
    double kap_k = 0.0;
    int par = 0;
    {
      const double kap = L.sc[F3_KAPPA];
      kap_k = kap;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          double lb[4];
          if (y_img) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) lb[qq] = imgY[(16 * ti + lrow + 4 * qq) * F3_S + 16 * tj + lcol];
          } else {
            const f64x2* d2_ = reinterpret_cast<const f64x2*>(L.dump) + (size_t)((ypar * 4 + 2 + tj) * 4) * 64 + lane;
            const f64x2 v0_ = d2_[(ti * 2) * 64], v1_ = d2_[(ti * 2 + 1) * 64];
            lb[0] = v0_[0]; lb[1] = v0_[1]; lb[2] = v1_[0]; lb[3] = v1_[1];
          }
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int e = (ti * 2 + tj) * 4 + qq;
            const double m = F3_VALID(mk, ti, tj, qq) ? lb[qq] + kap * G[e] : ((ti == tj && F3_DIAG(mk, ti, tj, qq)) ? 1.0 : 0.0);
            Mf[e] = m;
          }
        }
    }
    if (F3_PREDICT && try_ns && smw_ok && (p.ns_predict & 4)) {
      // start predictor of filter3 (psmf_blk3.hip, phase 1): Z_0 = (kappa_{k-1} / kappa_k) (Z - alpha a^T - beta b^T)
      const double sc = kap_prev * fast_rcp(kap_k);
      const f64x2 ab = *reinterpret_cast<const f64x2*>(L.sab + 2 * (32 * inv + 16 * C + lcol));
      const double aj = ab[0], bj = ab[1];
      const float* f32b = L.s32 + 128 * inv;
      const f32x4s al32[2] = {*reinterpret_cast<const f32x4s*>(f32b + 8 * lrow), *reinterpret_cast<const f32x4s*>(f32b + 8 * lrow + 4)};
      const f32x4s be32[2] = {*reinterpret_cast<const f32x4s*>(f32b + 32 + 8 * lrow), *reinterpret_cast<const f32x4s*>(f32b + 32 + 8 * lrow + 4)};
      const f32x2s abp0 = *reinterpret_cast<const f32x2s*>(f32b + 64 + 2 * pcol), abp1 = *reinterpret_cast<const f32x2s*>(f32b + 64 + 2 * (16 + pcol));
      const f64x2* alp = reinterpret_cast<const f64x2*>(L.sal + 32 * inv + 8 * lrow);
      const f64x2* bep = reinterpret_cast<const f64x2*>(L.sbe + 32 * inv + 8 * lrow);
      f64x2 al64[4], be64[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { al64[e] = alp[e]; be64[e] = bep[e]; }
      const float scf = (float)sc;
      const bool pc0 = FULL == 0 || (FULL == 2 ? pcol < rt : true), pc1 = FULL == 0 || (FULL == 2 ? false : pcol < rt);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int e = ti * 4 + qq;
          const double al = al64[e >> 1][e & 1], be = be64[e >> 1][e & 1];
          const bool rok = FULL == 0 || (FULL == 2 ? (ti == 0 && mk.r1[qq]) : (ti == 0 || mk.r1[qq]));
          const double xc = sc * fma(-be, bj, fma(-al, aj, Xc[e]));
          Xc[e] = F3_VALID(mk, ti, C, qq) ? xc : Xc[e];
          const float alf = al32[ti][qq], bef = be32[ti][qq];
          const float x0 = scf * fmaf(-bef, abp0[1], fmaf(-alf, abp0[0], Xa[e]));
          const float x1 = scf * fmaf(-bef, abp1[1], fmaf(-alf, abp1[0], Xa[8 + e]));
          Xa[e] = (rok && pc0) ? x0 : Xa[e];
          Xa[8 + e] = (rok && pc1) ? x1 : Xa[8 + e];
        }
    }
    kap_prev = kap_k;
    if (try_ns) F4_ITERATE(0);
    BLK_T(2);
    f3_barrier();                                                     // ---- B2
    BLK_T(1);
    // =============================== phase 2: second iteration, G update ===============================
    bool done = false, failed = !try_ns, last = false;
    int it = 0;
    fetch_late = false;
    if (try_ns) {
      F4_FETCH_PARTNER(0);
#ifdef F4_DEBUG
      if (role == 0 && lane == 0) { st->GR[jb * 8 + 4] = L.nrm[0] + L.nrm[1]; }
#endif
      F4_DECIDE(0, 0, it, done, failed, last);
      if (!done && !failed) {
        F4_ITERATE(1);
        par = 1;
        it = 1;
        if (last) { done = true; fetch_late = true; ++ctl.c_it; }
      }
    }
    {
      const double iN = L.sc[F3_INVN], ee = L.sc[F3_EE];
      const double e2 = ee * iN * iN;
      double hcol[2], wcol[2], hn[2];
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) { hcol[tj] = L.h[16 * tj + lcol]; wcol[tj] = L.w[16 * tj + lcol]; hn[tj] = hcol[tj] * iN; }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const double hi = L.h[16 * ti + lrow + 4 * qq], wi = L.w[16 * ti + lrow + 4 * qq];
          const double ui = hi * iN + e2 * wi;
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) G[(ti * 2 + tj) * 4 + qq] += ui * wcol[tj] + wi * hn[tj];
        }
    }
    BLK_T(3);
    if (!done) {
      f3_barrier();                                                   // ---- B3
      BLK_T(1);
      while (try_ns && !done && !failed) {
        F4_DECIDE(0, par, it, done, failed, last);
        if (done) fetch_late = true;
        if (done || failed) break;
        F4_FETCH_PARTNER(par);
        F4_ITERATE(par ^ 1);
        par ^= 1;
        ++it;
        if (last) { done = true; fetch_late = true; ++ctl.c_it; break; }
        f3_barrier();
      }
    }
    BLK_T(4);
    if (!done) {
      if (try_ns) { ++ctl.c_fail; F4_FAILED(ctl.x_skip, ctl.x_back); }
      ++ctl.c_sw;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) imgX[(16 * ti + lrow + 4 * qq) * F3_S + 16 * C + lcol] = Mf[(ti * 2 + C) * 4 + qq];
      f3_barrier();
      f4_sweep_image(L, imgX, r2, tid);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int row = 16 * ti + lrow + 4 * qq;
          const bool ok = F3_VALID(mk, ti, C, qq);
          const double xv = imgX[(ok ? row : 0) * F3_S + 16 * C + lcol];
          Xc[ti * 4 + qq] = ok ? xv : (F3_DIAG(mk, ti, C, qq) ? 1.0 : 0.0);
#pragma unroll
          for (int to = 0; to < 2; ++to) {
            const int cp = 16 * to + pcol;
            const bool okp = row < r && cp < r;
            const double xp = imgX[(okp ? row : 0) * F3_S + (okp ? cp : 0)];
            Xa[to * 8 + ti * 4 + qq] = okp ? (float)xp : (row == cp ? 1.f : 0.f);
          }
        }
      // the Y pair builds the next Pbar from the dump: republish the swept columns there
      par = 0;
      f64x2* dp_ = reinterpret_cast<f64x2*>(L.dump) + (size_t)((0 * 4 + role) * 4) * 64 + lane;
#pragma unroll
      for (int e_ = 0; e_ < 4; ++e_) dp_[e_ * 64] = f64x2{Xc[2 * e_], Xc[2 * e_ + 1]};
      f32x4s* pp_ = reinterpret_cast<f32x4s*>(L.dumpP) + (size_t)((0 * 4 + role) * 2) * 64 + 16 * lrow + pcol;
#pragma unroll
      for (int t_ = 0; t_ < 2; ++t_)
        pp_[t_ * 64] = f32x4s{(float)Xc[t_ * 4], (float)Xc[t_ * 4 + 1], (float)Xc[t_ * 4 + 2], (float)Xc[t_ * 4 + 3]};
    } else {
      ++ctl.c_ns;
      ctl.x_back = 3;
    }
    ctl.have_prev = true;
    w_par = par;
#ifdef F4_DEBUG
    if (role == 0 && lane == 0) { st->GR[jb * 8 + 5] = try_ns ? 1.0 : 0.0; st->GR[jb * 8 + 6] = done ? 1.0 : 0.0; st->GR[jb * 8 + 7] = (double)it; }
#endif
    const long long kstep = k.k0 + jb + 1;
    if (p.recursive && p.n_theta > 0 && (kstep % p.update_every) == 0) f3_barrier();     // ---- BA: wave 4 has stepped theta
    // =============================== phase F: v = P+ h, mu_k, mu_bar_{k+1}, F_{k+1} ===============================
    {
      // (h, w of this step stay in LDS until the next step's phase 1: read here, not carried in registers through the iterations)
      double vp0 = 0.0, vp1 = 0.0, vq0 = 0.0, vq1 = 0.0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const double h0 = L.h[lrow + 4 * qq], h1 = L.h[16 + lrow + 4 * qq], w0 = L.w[lrow + 4 * qq], w1 = L.w[16 + lrow + 4 * qq];
        vp0 += Xc[qq] * h0; vp1 += Xc[4 + qq] * h1;
        vq0 += Xc[qq] * w0; vq1 += Xc[4 + qq] * w1;
      }
      const double vp = xor32_sum_f64(xor16_sum_f64(vp0 + vp1));
      const int j = 16 * C + lcol;
      const double h_j = L.h[j], mub_j = L.mub[j];
      const double mu_new = mub_j + kap_k * vp;
      if (lrow == 0 && j < r) {
        double mb, fd, tp;
        f4_dyn_eval(p, D, j, (double)(kstep + 1), mu_new, mb, fd, tp);
        L.mub[j] = mb; D.fd[j] = fd; D.tp[j] = tp; D.mu[j] = mu_new;
        if (p.mu_hist) p.mu_hist[(size_t)(kstep - p.series_t0) * r + j] = mu_new;
      }
      if (p.robust) {
        const double hvp = wave_sum_f64_dpp((lrow == 0) ? h_j * vp : 0.0);
        if (lane == 0) L.hv[C] = hvp;
      }
      if (F3_PREDICT && (p.ns_predict & 1)) {
        const double vq = xor32_sum_f64(xor16_sum_f64(vq0 + vq1));
        if (lrow == 0) *reinterpret_cast<f64x2*>(L.sab + 2 * (32 * inv + j)) = f64x2{vp, vq};
      }
      smw_ok = true;
    }
    BLK_T(5);
    f3_barrier();                                                     // ---- BF
    BLK_T(1);
  }
  BLK_TOUT();
  // ---- block end ----
  if (fetch_late) F4_FETCH_PARTNER(w_par);
  f3_barrier();                       // wave 4 has published pscale of the last step
  const double ps = L.sc[F3_PSCALE];
#pragma unroll
  for (int e = 0; e < 8; ++e) st->f3_Xc[role][e * 64 + lane] = Xc[e];
#pragma unroll
  for (int e = 0; e < 16; ++e) st->f3_Xa[role][e * 64 + lane] = Xa[e];
  if (role == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) st->f3_G[e * 64 + lane] = G[e];
  }
  if (k.last) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int row = 16 * ti + lrow + 4 * qq, col = 16 * C + lcol;
        if (row < r && col < r) {
          const int idx = row * r + col;
          st->XpX[idx] = Xc[ti * 4 + qq];
          st->P[idx] = ps * Xc[ti * 4 + qq];
          st->G[idx] = G[(ti * 2 + C) * 4 + qq];
        }
      }
  }
  if (role == 0 && lane == 0) { st->cnt[0] += ctl.c_ns; st->cnt[1] += ctl.c_sw; st->cnt[2] += ctl.c_it; st->cnt[3] += ctl.c_fail; }
#undef Y_WORK
#undef Y_TO_IMAGE
}

// ------------------------------------------------------------------------------------------------------------
// Y pair: Lbar = Pbar^-1 of the current step.  C = own tile column; role 2 / 3.
// ------------------------------------------------------------------------------------------------------------
template <int C, int FULL>
__device__ __forceinline__ void f4_y_program(const BlockParams& b, const F3Blk& k, const F3Lds& L, const F4Lds& D, const int role, const int lane,
                                             const bool carried, const bool warm) {
  const StepParams& p = b.sp;
  DevState* st = p.st;
  const int r = p.r, r2 = r + (r & 1), tid = 64 * role + lane;
  const int lrow = lane >> 4, lcol = lane & 15;
  double* imgY = L.img + 32 * F3_S;
  const int pcol = (lcol >> 2) + 4 * (lcol & 3);
  const double dd = (double)p.d;
  F3Mask<FULL> mk;
  const int rt = FULL == 2 ? r : r - 16;
  mk.c1 = lcol < rt;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) { mk.r1[qq] = lrow + 4 * qq < rt; mk.dg[qq] = lcol == lrow + 4 * qq; }
  double Xc[8];
  float Xa[16];
#pragma unroll
  for (int e = 0; e < 8; ++e) Xc[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 16; ++e) Xa[e] = 0.f;
  // the scalars wave 4 runs (same expressions, same order: the same bits): q, lambda, pscale
  double q = st->Q[0], lam = st->lam;
  double pscale = carried ? st->f3_sc[2] : 1.0;
  double kap_done = 0.0, ee_done = 0.0;      // kappa, ee of the step that just ended
  f3_barrier();                                                       // ---- init barrier

#define Y_WORK(par_) do { F4_FETCH_PARTNER(par_); F4_ITERATE((par_) ^ 1); } while (0)
#define Y_TO_IMAGE()                                                                                       \
  do {                                                                                                     \
    _Pragma("unroll") for (int ti_ = 0; ti_ < 2; ++ti_)                                                    \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                     \
        imgY[(16 * ti_ + lrow + 4 * q_) * F3_S + 16 * C + lcol] = Mf[(ti_ * 2 + C) * 4 + q_];              \
  } while (0)

  F4Ctl ctl = {carried, 0, 0, 0, 0, 0, 0, 3, 3};
  int w_par = 0;               // parity of the X pair's latest columns (mirrors the X program)
  BLK_T0();
  for (int jb = 0; jb < k.nb; ++jb) {
    // =============================== phase 0: Pbar, start, first iteration ===============================
    if (jb > 0 && p.robust) {              // rpsmf.py:155-171, as wave 4 (F3_V0_FINISH_PREV)
      const double hPh = L.hv[0] + L.hv[1];
      const double quad = kap_done * ee_done - kap_done * kap_done * hPh;
      const double omega = (lam + quad) * fast_rcp(lam + dd);
      pscale = p.beta * omega;
      q *= omega;
      if (!p.fixed_lambda) lam += dd;
    } else if (jb > 0) {
      pscale = 1.0;
    }
    const double qk = p.q_sched ? q * D.qs[jb] : q;
    const bool try_y = p.use_ns && ctl.y_skip == 0;
    if (!try_y && ctl.y_skip > 0) --ctl.y_skip;
    double Mf[16], Xn[8];
    {
      const double iq = fast_rcp(qk), psq = pscale * iq;
      double fr[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) fr[e] = D.fd[16 * (e >> 2) + lrow + 4 * (e & 3)];
      const double fcol[2] = {D.fd[lcol], D.fd[16 + lcol]};
      const float fpc[2] = {(float)D.fd[pcol], (float)D.fd[16 + pcol]};
      // Pbar (all four tiles: the A operand) from the X pair's float64 columns
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const f64x2* d2_ = reinterpret_cast<const f64x2*>(L.dump) + (size_t)((w_par * 4 + tj) * 4) * 64 + lane;
          const f64x2 v0_ = d2_[(ti * 2) * 64], v1_ = d2_[(ti * 2 + 1) * 64];
          const double pp[4] = {v0_[0], v0_[1], v1_[0], v1_[1]};
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const bool dg = ti == tj && F3_DIAG(mk, ti, tj, qq);
            const double e_ = pscale * (fr[ti * 4 + qq] * fcol[tj]) * pp[qq];
            Mf[(ti * 2 + tj) * 4 + qq] = F3_VALID(mk, ti, tj, qq) ? e_ + (dg ? qk : 0.0) : (dg ? 1.0 : 0.0);
            if (tj == C) {
              // X_0 = (2 I - Pbar / q) / q = (I - E / q) / q: the first Newton-Schulz step from I / q
              Xc[ti * 4 + qq] = F3_VALID(mk, ti, C, qq) ? ((dg ? 1.0 : 0.0) - psq * (fr[ti * 4 + qq] * fcol[tj]) * pp[qq]) * iq : (dg ? 1.0 : 0.0);
            }
          }
        }
      // the same start as float32 A operands (P-layout): from the X pair's float32 columns
      const float psqf = (float)psq, iqf = (float)iq;
      const bool pc0 = FULL == 0 || (FULL == 2 ? pcol < rt : true), pc1 = FULL == 0 || (FULL == 2 ? false : pcol < rt);
#pragma unroll
      for (int to = 0; to < 2; ++to) {
        const f32x4s* pq_ = reinterpret_cast<const f32x4s*>(L.dumpP) + (size_t)((w_par * 4 + to) * 2) * 64 + lane;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const f32x4s v_ = pq_[kt * 64];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const bool rok = FULL == 0 || (FULL == 2 ? (kt == 0 && mk.r1[kk]) : (kt == 0 || mk.r1[kk]));
            const bool ok = rok && (to == 0 ? pc0 : pc1);
            const bool dg = kt == to && (lrow + 4 * kk == pcol);
            const float x0 = ((dg ? 1.f : 0.f) - psqf * ((float)fr[kt * 4 + kk] * fpc[to]) * v_[kk]) * iqf;
            Xa[to * 8 + kt * 4 + kk] = ok ? x0 : (dg ? 1.f : 0.f);
          }
        }
      }
    }
    if (try_y) F4_ITERATE(0);
    BLK_T(7);
    f3_barrier();                                                     // ---- B1
    BLK_T(1);
    F4_Y_CONTROL();
    (void)y_img;
    BLK_T(6);
    // =============================== phase 1 (idle) ===============================
    const bool try_ns = ctl.have_prev && p.use_ns && ctl.x_skip == 0;
    if (!try_ns && ctl.x_skip > 0) --ctl.x_skip;
    kap_done = L.sc[F3_KAPPA];                    // stable from B1 to the next step's phase 0
    int par = 0;
    BLK_T(2);
    f3_barrier();                                                     // ---- B2
    BLK_T(1);
    // =============================== phase 2: mirror the X pair's decisions ===============================
    bool done = false, failed = !try_ns, last = false;
    int it = 0;
    ee_done = L.sc[F3_EE];                        // written in phase 1, stable until the next step's phase 1
    if (try_ns) {
      F4_DECIDE(0, 0, it, done, failed, last);
      if (!done && !failed) {
        par = 1;
        it = 1;
        if (last) done = true;
      }
    }
    BLK_T(3);
    if (!done) {
      f3_barrier();                                                   // ---- B3
      while (try_ns && !done && !failed) {
        F4_DECIDE(0, par, it, done, failed, last);
        if (done || failed) break;
        par ^= 1;
        ++it;
        if (last) { done = true; break; }
        f3_barrier();
      }
    }
    if (!done) {
      if (try_ns) F4_FAILED(ctl.x_skip, ctl.x_back);
      f3_barrier();
      f4_sweep_image(L, L.img, r2, tid);          // the X pair has published M in image X: waves 0-3 sweep it
      par = 0;
    } else {
      ctl.x_back = 3;
    }
    ctl.have_prev = true;
    w_par = par;
    BLK_T(4);
    const long long kstep = k.k0 + jb + 1;
    if (p.recursive && p.n_theta > 0 && (kstep % p.update_every) == 0) f3_barrier();     // ---- BA
    BLK_T(5);
    f3_barrier();                                                     // ---- BF
    BLK_T(1);
  }
  BLK_TOUT();
  f3_barrier();
  (void)st; (void)warm;
#undef Y_WORK
#undef Y_TO_IMAGE
#undef F4_ITERATE
#undef F4_FETCH_PARTNER
}

// ------------------------------------------------------------------------------------------------------------
// Vector waves: filter3's program (4 = V and every scalar, 5 = A by rows, 6 = KA by rows, 7 = A^T by columns and the start
// predictor's 2 x 2 core) with the step control of this kernel; wave 4 also runs the R_k / Q_k schedules and the theta path.
// ------------------------------------------------------------------------------------------------------------
template <int ROLE>       // compile-time role: one loop per wave, holding only its own registers (see f5_v_program)
__device__ __forceinline__ void f4_v_program(const BlockParams& b, const F3Blk& k, const F3Lds& L, const F4Lds& D, const int lane,
                                             const bool carried, const bool warm) {
  const StepParams& p = b.sp;
  DevState* st = p.st;
  constexpr int role = ROLE;
  const int r = p.r, r2 = r + (r & 1), tid = 64 * role + lane;
  const double dd = (double)p.d, idd = 1.0 / dd;
  constexpr bool isV0 = ROLE == 4, isV1 = ROLE == 5, isV2 = ROLE == 6, isV3 = ROLE == 7;
  __builtin_amdgcn_s_setprio(3);
  double pr[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) pr[i] = 0.0;
  double kappa = 0.0, Nk = 0.0, invN = 0.0, s_k = 0.0, eta_k = 0.0, ee_k = 0.0, phi = 1.0, omega = 1.0, pscale = 1.0, wj = 0.0;
  double q = st->Q[0];
  double rho = st->rho, lam = st->lam;
  double kap7 = warm ? L.sc[F3_KAPPA] : 1.0;      // wave 7: kappa of the step that just ended (read before the init barrier)
  // theta path (wave 4, lanes j < r of the lower half own theta_j): gradient sums, Adam moments
  const bool has_th = p.n_theta > 0;
  const bool phased = p.dyn_kind == DYN_SINUSOID && (p.dyn_flags & 2);
  const int jth = lane & 31;
  const bool own_th = isV0 && has_th && lane < 32 && jth < r;
  double gs_b = 0.0, gs_c = 0.0, am_b = 0.0, av_b = 0.0, am_c = 0.0, av_c = 0.0, th_b = 0.0, th_c = 0.0;
  double b1k = 1.0, b2k = 1.0, lr_k = p.lr, lr_g = 1.0;
  if (own_th) {
    gs_b = p.gradsum[jth]; th_b = p.theta[jth];
    if (phased) { gs_c = p.gradsum[r + jth]; th_c = p.theta[r + jth]; }
    if (p.recursive) {
      am_b = p.adam_m[jth]; av_b = p.adam_v[jth];
      if (phased) { am_c = p.adam_m[r + jth]; av_c = p.adam_v[r + jth]; }
    }
  }
  if (isV0 && has_th && p.recursive) {
    // psmf.py:224-242 with the step index: b^k and the decayed learning rate carried as running products within the block
    b1k = pow(p.b1, (double)k.k0); b2k = pow(p.b2, (double)k.k0);
    if (p.lr_steps > 0.0) { lr_k = p.lr * pow(p.lr_end / p.lr, (double)k.k0 / p.lr_steps); lr_g = pow(p.lr_end / p.lr, 1.0 / p.lr_steps); }
  }
  if (isV0) {
    const int j = lane & 31, hf = lane >> 5;
    if (carried) {
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] = st->f3_V[t * 64 + lane];
      pscale = st->f3_sc[2];
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = 16 * hf + t;
        const bool in = i < r && j < r;
        const double v = st->V[in ? i * r + j : 0];
        pr[t] = in ? v : 0.0;
      }
    }
  } else if (isV1) {
#pragma unroll
    for (int c = 0; c < 32; ++c) pr[c] = (lane == c && c < r) ? 1.0 : 0.0;
  } else if (isV2) {
#pragma unroll
    for (int c = 0; c < 32; ++c) pr[c] = (c < r) ? L.sK[lane * RB + c] : 0.0;
  } else {
    const int c = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int t = 0; t < 32; ++t) pr[t] = (32 * hf + t == c && c < r) ? 1.0 : 0.0;
  }
  f3_barrier();                                                       // ---- init barrier

#define F4_V0_FINISH_PREV()                                               \
  do {                                                                    \
    omega = 1.0;                                                          \
    pscale = 1.0;                                                         \
    if (p.robust) {                                                       \
      const double hPh_ = L.hv[0] + L.hv[1];                              \
      const double quad_ = kappa * ee_k - kappa * kappa * hPh_;           \
      omega = (lam + quad_) * fast_rcp(lam + dd);                         \
      pscale = p.beta * omega;                                            \
      rho *= omega;                                                       \
      q *= omega;                                                         \
      if (!p.fixed_lambda) lam += dd;                                     \
    }                                                                     \
  } while (0)
#define Y_WORK(par_) do { } while (0)
#define Y_TO_IMAGE() do { } while (0)

  F4Ctl ctl = {carried, 0, 0, 0, 0, 0, 0, 3, 3};
  if (role == 4 && lane == 0) L.tick[0] = (long long)__builtin_amdgcn_s_memrealtime();
  BLK_T0();
  for (int jb = 0; jb < k.nb; ++jb) {
    const long long kstep = k.k0 + jb + 1;
    // =============================== phase 0 ===============================
    double cm = 0.0;
    double tp_j = 0.0, x_j = 0.0;          // wave 4: trig'(arg_j) and mu_{k-1,j} of THIS step (the X pair rewrites them in phase F)
    if (isV0) {
      const int j = lane & 31, hf = lane >> 5;
      if (jb > 0) F4_V0_FINISH_PREV();
      if (p.rho_sched) rho = D.rs[jb];                                  // PSMFIter reads R[k], Q[k] of the step (psmf.py:115,123,141)
      double part0 = 0.0, part1 = 0.0;
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        part0 += pr[t] * L.mub[16 * hf + t];
        part1 += pr[t + 1] * L.mub[16 * hf + t + 1];
      }
      const double part = part0 + part1;
      s_k = wave_sum_f64_dpp(part * L.mub[j]);
      kappa = fast_rcp(rho + s_k);
      if (lane == 0) L.sc[F3_KAPPA] = kappa;
      wj = part;
      if (has_th) { tp_j = D.tp[j]; x_j = D.mu[j]; }
    } else if (isV1 || isV2) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        a0 += pr[c] * L.mub[c];
        a1 += pr[c + 1] * L.mub[c + 1];
        a2 += pr[c + 2] * L.mub[c + 2];
        a3 += pr[c + 3] * L.mub[c + 3];
      }
      const double dot = (a0 + a1) + (a2 + a3);
      if (isV1) {
        cm = (lane == r + jb ? 1.0 : 0.0) - dot;
        L.a[lane] = cm;
        coef_store(k.Bcoef + (size_t)jb * RB + lane, dot);
      } else {
        cm = L.sK[lane * RB + r + jb] - dot;
        L.Ka[lane] = cm;
      }
    } else if (F3_PREDICT && isV3 && (p.ns_predict & 2)) {
      // 2 x 2 core of the start predictor of P+ (psmf_blk3.hip, f3_v_program); only the lower half (P+) is used here
      const int j = lane & 31, hf = lane >> 5;
      double al = 0.0, be = 0.0;
      f64x2 ab7 = {0.0, 0.0};
      if (jb > 0 || warm) {  // (the upper half computes on the unused W slots: never stored)
        ab7 = *reinterpret_cast<const f64x2*>(L.sab + 2 * lane);
        const double a = ab7[0], bb = ab7[1], hj = L.h[j], wv = L.w[j];
        const double ha = xor16_sum_f64(row_sum_f64_dpp(hj * a)), hb = xor16_sum_f64(row_sum_f64_dpp(hj * bb));
        const double wb = xor16_sum_f64(row_sum_f64_dpp(wv * bb));
        const double ik = fast_rcp(kap7);
        const double s11 = ha - L.sc[F3_EE] * ik, s12 = hb + L.sc[F3_N] * ik;
        const double idet = fast_rcp(s11 * wb - s12 * s12);
        const double t11 = wb * idet, t12 = -s12 * idet, t22 = s11 * idet;
        al = t11 * a + t12 * bb;
        be = t12 * a + t22 * bb;
      }
      if (hf == 0) {
        const int pj = 8 * (j & 3) + 4 * (j >> 4) + ((j >> 2) & 3);
        L.sal[pj] = al;
        L.sbe[pj] = be;
        float* f32b = L.s32;
        f32b[pj] = (float)al;
        f32b[32 + pj] = (float)be;
        *reinterpret_cast<f32x2s*>(f32b + 64 + 2 * j) = f32x2s{(float)ab7[0], (float)ab7[1]};
      }
    }
    BLK_T(0);
    f3_barrier();                                                     // ---- B1
    BLK_T(1);
    const bool try_y = p.use_ns && ctl.y_skip == 0;
    if (!try_y && ctl.y_skip > 0) --ctl.y_skip;
    // phase 1 work that does not wait for the Y pair comes first: it overlaps with the Y pair's further iterations, if any
    const bool try_ns = ctl.have_prev && p.use_ns && ctl.x_skip == 0;
    if (!try_ns && ctl.x_skip > 0) --ctl.x_skip;
    int par = 0;
    if (isV3) kap7 = L.sc[F3_KAPPA];
    if (isV3) {
      const int c = lane & 31, hf = lane >> 5;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int t = 0; t < 32; t += 4) {
        a0 += pr[t] * L.Ka[32 * hf + t];
        a1 += pr[t + 1] * L.Ka[32 * hf + t + 1];
        a2 += pr[t + 2] * L.Ka[32 * hf + t + 2];
        a3 += pr[t + 3] * L.Ka[32 * hf + t + 3];
      }
      const double part = (a0 + a1) + (a2 + a3);
      const double hc = xor32_sum_f64(part);                            // h = A^T K a
      if (hf == 0) L.h[c] = hc;
    } else if (isV1) {
      const double e1 = wave_sum_f64_dpp(cm * L.Ka[lane]);
      if (lane == 0) L.sc[F3_EE] = e1;
    } else if (isV0) {
      wj = xor32_sum_f64(wj);                                           // w = V mu_bar
      if (lane < 32) L.w[lane] = wj;
      const double qk = p.q_sched ? q * D.qs[jb] : q;
      const double gpv = pscale * (L.gp[0] + L.gp[1]) + qk * (L.tr[0] + L.tr[1]);
      eta_k = rho + gpv * idd;
      Nk = s_k + eta_k;
      invN = fast_rcp(Nk);
      if (lane == 0) { L.sc[F3_N] = Nk; L.sc[F3_INVN] = invN; }
    }
    F4_Y_CONTROL();
    (void)y_img;
    BLK_T(2);
    f3_barrier();                                                     // ---- B2
    BLK_T(1);
    // =============================== phase 2 ===============================
    bool done = false, failed = !try_ns, last = false;
    int it = 0;
    if (try_ns) {
      F4_DECIDE(0, 0, it, done, failed, last);
      if (!done && !failed) {
        par = 1;
        it = 1;
        if (last) done = true;
      }
    }
    if (isV3) {
      const int c = lane & 31, hf = lane >> 5;
      const double wn = L.w[c] * L.sc[F3_INVN];
#pragma unroll
      for (int t = 0; t < 32; ++t) pr[t] += L.a[32 * hf + t] * wn;
    } else if (isV1 || isV2) {
      const double cn = cm * L.sc[F3_INVN];
#pragma unroll
      for (int c = 0; c < 32; ++c) pr[c] += cn * L.w[c];
    } else if (isV0) {
      const int hf = lane >> 5;
      ee_k = L.sc[F3_EE];
      phi = 1.0;
      double vscale = 1.0;
      if (p.robust) {
        phi = (lam + ee_k * invN) * fast_rcp(lam + dd);
        vscale = p.alpha * phi;
      }
      const double wjn = wj * invN;
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] = vscale * (pr[t] - L.w[16 * hf + t] * wjn);
      if (has_th) {
        // theta gradient at the pre-update state (psmf.py:57-64, rpsmf.py:62-71): g_f, then J_theta^T g_f (diagonal)
        const double hj = L.h[jth];
        double gf;
        if (p.robust) {
          const double Dn = lam * Nk;
          gf = dd * wj / Nk + 0.5 * (dd + lam) * (-2.0 * hj / Dn - 2.0 * lam * ee_k * wj / (Dn * Dn)) / (1.0 + ee_k / Dn);
        } else {
          gf = dd * wj * invN - hj * invN - ee_k * wj * invN * invN;
        }
        const double ut = gf * tp_j;
        gs_b += ut * (2.0 * M_PI * (double)kstep);
        if (phased) gs_c += ut * x_j;
        if (p.recursive) {
          b1k *= p.b1; b2k *= p.b2;
          if (p.lr_steps > 0.0) lr_k *= lr_g;
          if ((kstep % p.update_every) == 0) {
            const double c1 = 1.0 / (1.0 - b1k), c2 = 1.0 / (1.0 - b2k);
            am_b = p.b1 * am_b + (1.0 - p.b1) * gs_b;
            av_b = p.b2 * av_b + (1.0 - p.b2) * gs_b * gs_b;
            th_b = fmax(th_b - lr_k * (am_b * c1) / (sqrt(av_b * c2) + 1e-8), 0.0);
            gs_b = 0.0;
            if (phased) {
              am_c = p.b1 * am_c + (1.0 - p.b1) * gs_c;
              av_c = p.b2 * av_c + (1.0 - p.b2) * gs_c * gs_c;
              th_c = fmax(th_c - lr_k * (am_c * c1) / (sqrt(av_c * c2) + 1e-8), 0.0);
              gs_c = 0.0;
            }
            if (own_th) { D.th[jth] = th_b; if (phased) D.th[RM + jth] = th_c; }
          }
        }
      }
    }
    BLK_T(3);
    if (!done) {
      f3_barrier();                                                   // ---- B3
      while (try_ns && !done && !failed) {
        F4_DECIDE(0, par, it, done, failed, last);
        if (done || failed) break;
        par ^= 1;
        ++it;
        if (last) { done = true; break; }
        f3_barrier();
      }
    }
    if (!done) {
      if (try_ns) F4_FAILED(ctl.x_skip, ctl.x_back);
      f3_barrier();
      f4_sweep_image(L, L.img, r2, tid);
    } else {
      ctl.x_back = 3;
    }
    ctl.have_prev = true;
    BLK_T(4);
    if (p.recursive && p.n_theta > 0 && (kstep % p.update_every) == 0) f3_barrier();     // ---- BA: theta of the next step is in LDS
    f3_barrier();                                                     // ---- BF
    BLK_T(1);
  }
  BLK_TOUT();
  if (role == 4 && lane == 0) L.tick[1] = (long long)__builtin_amdgcn_s_memrealtime();

  // ---- block end ----
  if (isV0) {
    if (k.nb > 0) F4_V0_FINISH_PREV();
    if (lane == 0) { L.sc[F3_PSCALE] = pscale; L.sc[F3_Q] = q; }
  } else if (isV1) {
#pragma unroll
    for (int c = 0; c < 32; ++c) L.sA[lane * F3_AS + c] = pr[c];
  }
  f3_barrier();
  if (isV0) {
    const int j = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int t = 0; t < 16; ++t) st->f3_V[t * 64 + lane] = pr[t];
    if (k.last) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = 16 * hf + t;
        if (i < r && j < r) {
          st->V[i * r + j] = pr[t];
          st->Q[i * r + j] = (i == j) ? q : 0.0;
        }
      }
    } else if (hf == 0 && j < r) {
      st->Q[j * r + j] = q;
    }
    if (lane < r) st->mu[lane] = D.mu[lane];
    if (own_th) {
      p.gradsum[jth] = gs_b;
      if (phased) p.gradsum[r + jth] = gs_c;
      if (p.recursive) {
        p.theta[jth] = th_b; p.adam_m[jth] = am_b; p.adam_v[jth] = av_b;
        if (phased) { p.theta[r + jth] = th_c; p.adam_m[r + jth] = am_c; p.adam_v[r + jth] = av_c; }
      }
    }
    if (lane == 0) {
      st->f3_sc[2] = pscale;
      st->k = k.k0 + k.nb;
      st->rho = rho; st->lam = lam; st->phi = phi; st->omega = omega; st->ee = ee_k;
      st->s_done = s_k; st->eta_done = eta_k; st->N_done = Nk;
      if (*L.errflag && st->err == 0) st->err = (int)(k.k0 + 1);
      st->ns_valid = 4;
    }
  }
  for (int idx = tid - 256; idx < RB * r; idx += 256) { const int m = idx / r, c = idx - m * r; coef_store(k.Acoef + idx, L.sA[m * F3_AS + c]); }
#undef F4_V0_FINISH_PREV
#undef Y_WORK
#undef Y_TO_IMAGE
}

// ------------------------------------------------------------------------------------------------------------
// "filter5": the SIMPLIFIED hook configuration of the synthetic experiments (ExperimentSynthetic/synthetic_psmf.py:78-100,
// synthetic_rpsmf.py:82-118; SURVEY App. A mode table: P_bar = P_{k-1}, eta = tr(R_{k-1}) / d, no coefficient update: mu_k = mu_bar_k,
// P_k = P_bar_k; rPSMF: omega from (R + s I)^-1 alone, R scaled by it, Q and P carried) with diagonal-Jacobian dynamics.  No r x r
// inversion is left in a timestep, so the whole step is the vector program of filter3 / filter4 -- V and the scalars (wave 4), A by
// rows (5), K A by rows (6), A^T by columns (7) -- plus mu_bar_{k+1} = f(theta, mu_bar_k, k + 1), which wave 4 iterates by itself
// (it depends on nothing the step computes), the theta gradient and, for the recursive classes, Adam.  Three barriers per timestep.
// The general kernel ran these modes at 5.4 us per timestep (seven barrier-separated stages of a 256-thread group).
// Launched with the 512 threads of the common skeleton (K assembly, chained blocks); waves 0-3 only keep the barrier count.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void f5_idle_program(const F3Blk& k) {
  f3_barrier();                                                       // init barrier
  for (int jb = 0; jb < k.nb; ++jb) {
    f3_barrier();                                                     // B1
    f3_barrier();                                                     // B2
    f3_barrier();                                                     // BF
  }
  f3_barrier();                                                       // block end
}

// ROLE is a compile-time constant: each of the four waves gets a loop that holds only its own registers (with the role as a
// run-time value the allocator kept the union of all four live: 227 spilled registers in a program that needs ~100)
template <int ROLE>
__device__ __forceinline__ void f5_v_program(const BlockParams& b, const F3Blk& k, const F3Lds& L, const F4Lds& D, const int lane,
                                             const bool carried) {
  const StepParams& p = b.sp;
  DevState* st = p.st;
  constexpr int role = ROLE;
  const int r = p.r, tid = 64 * role + lane;
  const double dd = (double)p.d;
  constexpr bool isV0 = ROLE == 4, isV1 = ROLE == 5, isV2 = ROLE == 6, isV3 = ROLE == 7;
  double pr[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) pr[i] = 0.0;
  double kappa = 0.0, Nk = 0.0, invN = 0.0, s_k = 0.0, eta_k = 0.0, ee_k = 0.0, phi = 1.0, omega = 1.0, wj = 0.0;
  double rho = st->rho, lam = st->lam;
  const bool has_th = p.n_theta > 0;
  const bool phased = p.dyn_kind == DYN_SINUSOID && (p.dyn_flags & 2);
  const int jth = lane & 31;
  const bool own_th = isV0 && has_th && lane < 32 && jth < r;
  double gs_b = 0.0, gs_c = 0.0, am_b = 0.0, av_b = 0.0, am_c = 0.0, av_c = 0.0;
  double b1k = 1.0, b2k = 1.0, lr_k = p.lr, lr_g = 1.0;
  double mu_j = 0.0, mub_j = 0.0, tp_j = 0.0;         // wave 4, lane j: mu_{k-1,j}, mu_bar_{k,j}, trig'(arg_kj)
  // wave 6 also tracks G = C^T C (rank-2 update per timestep): the next block's K is assembled from it
  // (K A_0 = the first r columns of K).  Layout: lane = (column c = lane & 31, half hf = lane >> 5), g[t] = G[16 hf + t][c].
  double g[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) g[t] = 0.0;
  if (own_th) {
    gs_b = p.gradsum[jth];
    if (phased) gs_c = p.gradsum[r + jth];
    if (p.recursive) {
      am_b = p.adam_m[jth]; av_b = p.adam_v[jth];
      if (phased) { am_c = p.adam_m[r + jth]; av_c = p.adam_v[r + jth]; }
    }
  }
  if (isV0 && has_th && p.recursive) {
    b1k = pow(p.b1, (double)k.k0); b2k = pow(p.b2, (double)k.k0);
    if (p.lr_steps > 0.0) { lr_k = p.lr * pow(p.lr_end / p.lr, (double)k.k0 / p.lr_steps); lr_g = pow(p.lr_end / p.lr, 1.0 / p.lr_steps); }
  }
  if (isV0) {
    const int j = lane & 31, hf = lane >> 5;
    if (carried) {
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] = st->f3_V[t * 64 + lane];
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = 16 * hf + t;
        const bool in = i < r && j < r;
        const double v = st->V[in ? i * r + j : 0];
        pr[t] = in ? v : 0.0;
      }
    }
    // mu_bar and trig' of the block's first step (D.mu = mu_{k0}, D.th = theta: filled by the skeleton)
    mu_j = D.mu[j];
    double fd;
    if (j < r) f4_dyn_eval(p, D, j, (double)(k.k0 + 1), mu_j, mub_j, fd, tp_j);
    if (lane < 32) L.mub[j] = j < r ? mub_j : 0.0;
  } else if (isV1) {
#pragma unroll
    for (int c = 0; c < 32; ++c) pr[c] = (lane == c && c < r) ? 1.0 : 0.0;
  } else if (isV2) {
    const int c = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) pr[cc] = (cc < r) ? L.sK[lane * RB + cc] : 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) g[t] = (16 * hf + t < r && c < r) ? L.sK[(16 * hf + t) * RB + c] : 0.0;    // G_0: exact Gram / tracked G
  } else {
    const int c = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int t = 0; t < 32; ++t) pr[t] = (32 * hf + t == c && c < r) ? 1.0 : 0.0;
  }
  f3_barrier();                                                       // ---- init barrier
  if (role == 4 && lane == 0) L.tick[0] = (long long)__builtin_amdgcn_s_memrealtime();
  BLK_T0();
  for (int jb = 0; jb < k.nb; ++jb) {
    const long long kstep = k.k0 + jb + 1;
    // =============================== phase 0: w = V mu_bar, s | a = u - A mu_bar, K a ===============================
    double cm = 0.0;
    bool mub_next_ready = false;
    double tp_prev = tp_j;
    const double mu_prev = mu_j;
    if (isV0) {
      const int j = lane & 31, hf = lane >> 5;
      if (p.rho_sched) rho = D.rs[jb];
      double part0 = 0.0, part1 = 0.0;
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        part0 += pr[t] * L.mub[16 * hf + t];
        part1 += pr[t + 1] * L.mub[16 * hf + t + 1];
      }
      const double part = part0 + part1;
      s_k = wave_sum_f64_dpp(part * L.mub[j]);
      kappa = fast_rcp(rho + s_k);
      eta_k = rho;                                  // tr(R) / d (synthetic_psmf.py:86-87)
      Nk = s_k + eta_k;
      invN = fast_rcp(Nk);
      if (lane == 0) { L.sc[F3_N] = Nk; L.sc[F3_INVN] = invN; }
      wj = xor32_sum_f64(part);                     // w = V mu_bar
      if (lane < 32) L.w[lane] = wj;
    } else if (isV1 || isV2) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        a0 += pr[c] * L.mub[c];
        a1 += pr[c + 1] * L.mub[c + 1];
        a2 += pr[c + 2] * L.mub[c + 2];
        a3 += pr[c + 3] * L.mub[c + 3];
      }
      const double dot = (a0 + a1) + (a2 + a3);
      if (isV1) {
        cm = (lane == r + jb ? 1.0 : 0.0) - dot;
        L.a[lane] = cm;
        coef_store(k.Bcoef + (size_t)jb * RB + lane, dot);
      } else {
        cm = L.sK[lane * RB + r + jb] - dot;
        L.Ka[lane] = cm;
      }
    }
    BLK_T(0);
    f3_barrier();                                                     // ---- B1
    BLK_T(1);
    // =============================== phase 1: h = A^T K a, ee; rank-1 updates that need only w, N ===============================
    if (isV3) {
      const int c = lane & 31, hf = lane >> 5;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int t = 0; t < 32; t += 4) {
        a0 += pr[t] * L.Ka[32 * hf + t];
        a1 += pr[t + 1] * L.Ka[32 * hf + t + 1];
        a2 += pr[t + 2] * L.Ka[32 * hf + t + 2];
        a3 += pr[t + 3] * L.Ka[32 * hf + t + 3];
      }
      const double hc = xor32_sum_f64((a0 + a1) + (a2 + a3));
      if (hf == 0) L.h[c] = hc;
    } else if (isV1 || isV2) {
      if (isV1) {
        const double e1 = wave_sum_f64_dpp(cm * L.Ka[lane]);
        if (lane == 0) L.sc[F3_EE] = e1;
      }
      const double cn = cm * L.sc[F3_INVN];
#pragma unroll
      for (int c = 0; c < 32; ++c) pr[c] += cn * L.w[c];                // A, K A by rows
    } else {
      // wave 4 has nothing to wait for here: the rank-1 part of the V update (w, N are its own) and -- unless theta is stepped
      // inside the loop -- the next step's mu_bar, which depends on nothing this step computes
      const int j = lane & 31, hf = lane >> 5;
      const double wjn = wj * invN;
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] -= L.w[16 * hf + t] * wjn;    // psmf.py:135-138
      if (!(p.recursive && has_th)) {
        mu_j = mub_j;                                                   // mu_k = mu_bar_k (no coefficient update)
        tp_prev = tp_j;
        if (lane < 32 && j < r && p.mu_hist) p.mu_hist[(size_t)(kstep - p.series_t0) * r + j] = mu_j;
        double fd;
        if (j < r) f4_dyn_eval(p, D, j, (double)(kstep + 1), mu_j, mub_j, fd, tp_j);
        mub_next_ready = true;
      }
    }
    BLK_T(2);
    f3_barrier();                                                     // ---- B2
    BLK_T(1);
    // =============================== phase 2 (wave 4): V, the scalars, gradient, Adam, the next mu_bar; (wave 7): G ===============================
    if (isV3) {
      // A^T by columns (psmf.py:130-133 in coefficient space): a, w, N of this step are stable until the next step's phases 0 / 0 / 0
      const int c = lane & 31, hf = lane >> 5;
      const double wn = L.w[c] * L.sc[F3_INVN];
#pragma unroll
      for (int t = 0; t < 32; ++t) pr[t] += L.a[32 * hf + t] * wn;
    } else if (isV2) {
      // G_k = G_{k-1} + (h w^T + w h^T) / N + ee w w^T / N^2 = G + u w^T + w hn^T,  u = h / N + (ee / N^2) w,  hn = h / N
      const int c = lane & 31, hf = lane >> 5;
      const double iN = L.sc[F3_INVN], e2 = L.sc[F3_EE] * iN * iN;
      const double wc = L.w[c], hnc = L.h[c] * iN;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const double hi = L.h[16 * hf + t], wi = L.w[16 * hf + t];
        g[t] += (hi * iN + e2 * wi) * wc + wi * hnc;
      }
    } else if (isV0) {
      const int j = lane & 31, hf = lane >> 5;
      ee_k = L.sc[F3_EE];
      phi = 1.0; omega = 1.0;
      double vscale = 1.0;
      const double lam0 = lam, N0 = Nk;
      if (p.robust) {
        const double ild = fast_rcp(lam + dd);
        phi = (lam + ee_k * invN) * ild;                                // rpsmf.py:133-138
        omega = (lam + kappa * ee_k) * ild;                             // synthetic_rpsmf.py:91-107: S^-1 = (R + s I)^-1
        vscale = p.alpha * phi;
      }
      if (p.robust) {
#pragma unroll
        for (int t = 0; t < 16; ++t) pr[t] *= vscale;
      }
      (void)hf;
      if (has_th) {
        const double hj = L.h[j];
        double gf;
        if (p.robust) {
          const double Dn = lam0 * N0;
          gf = dd * wj / N0 + 0.5 * (dd + lam0) * (-2.0 * hj / Dn - 2.0 * lam0 * ee_k * wj / (Dn * Dn)) / (1.0 + ee_k / Dn);
        } else {
          gf = dd * wj * invN - hj * invN - ee_k * wj * invN * invN;
        }
        // (trig' and mu_{k-1} of THIS step: when the next mu_bar was formed in phase 1 they are the saved ones)
        const double tpk = mub_next_ready ? tp_prev : tp_j, xk = mub_next_ready ? mu_prev : mu_j;
        const double ut = gf * tpk;
        gs_b += ut * (2.0 * M_PI * (double)kstep);
        if (phased) gs_c += ut * xk;
        if (p.recursive) {
          b1k *= p.b1; b2k *= p.b2;
          if (p.lr_steps > 0.0) lr_k *= lr_g;
          if ((kstep % p.update_every) == 0) {
            const double c1 = 1.0 / (1.0 - b1k), c2 = 1.0 / (1.0 - b2k);
            am_b = p.b1 * am_b + (1.0 - p.b1) * gs_b;
            av_b = p.b2 * av_b + (1.0 - p.b2) * gs_b * gs_b;
            const double thb = fmax(D.th[j] - lr_k * (am_b * c1) / (sqrt(av_b * c2) + 1e-8), 0.0);
            gs_b = 0.0;
            double thc = 0.0;
            if (phased) {
              am_c = p.b1 * am_c + (1.0 - p.b1) * gs_c;
              av_c = p.b2 * av_c + (1.0 - p.b2) * gs_c * gs_c;
              thc = fmax(D.th[RM + j] - lr_k * (am_c * c1) / (sqrt(av_c * c2) + 1e-8), 0.0);
              gs_c = 0.0;
            }
            if (own_th) { D.th[j] = thb; if (phased) D.th[RM + j] = thc; }      // (each lane reads back only what it wrote)
          }
        }
      }
      if (p.robust) { rho *= omega; if (!p.fixed_lambda) lam += dd; }
      // mu_k = mu_bar_k (no coefficient update); mu_bar_{k+1} = f(theta, mu_k, k + 1)
      if (!mub_next_ready) {
        mu_j = mub_j;
        if (lane < 32 && j < r && p.mu_hist) p.mu_hist[(size_t)(kstep - p.series_t0) * r + j] = mu_j;
        double fd;
        if (j < r) f4_dyn_eval(p, D, j, (double)(kstep + 1), mu_j, mub_j, fd, tp_j);
      }
      if (lane < 32 && j < r) L.mub[j] = mub_j;       // (read by the other waves in the next phase 0 only)
    }
    BLK_T(3);
    f3_barrier();                                                     // ---- BF
    BLK_T(1);
  }
  BLK_TOUT();
  if (role == 4 && lane == 0) L.tick[1] = (long long)__builtin_amdgcn_s_memrealtime();
  // ---- block end ----
  if (isV1) {
#pragma unroll
    for (int c = 0; c < 32; ++c) L.sA[lane * F3_AS + c] = pr[c];
  }
  f3_barrier();
  if (isV0) {
    const int j = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int t = 0; t < 16; ++t) st->f3_V[t * 64 + lane] = pr[t];
    if (k.last) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = 16 * hf + t;
        if (i < r && j < r) st->V[i * r + j] = pr[t];
      }
    }
    if (lane < r) st->mu[lane] = mu_j;
    if (own_th) {
      p.gradsum[jth] = gs_b;
      if (phased) p.gradsum[r + jth] = gs_c;
      if (p.recursive) {
        p.theta[jth] = D.th[jth]; p.adam_m[jth] = am_b; p.adam_v[jth] = av_b;
        if (phased) { p.theta[r + jth] = D.th[RM + jth]; p.adam_m[r + jth] = am_c; p.adam_v[r + jth] = av_c; }
      }
    }
    if (lane == 0) {
      st->k = k.k0 + k.nb;
      st->rho = rho; st->lam = lam; st->phi = phi; st->omega = omega; st->ee = ee_k;
      st->s_done = s_k; st->eta_done = eta_k; st->N_done = Nk;
      st->ns_valid = 5;
      st->cnt[0] += k.nb;            // (no inversion in this mode: every timestep counts as "iterated", none swept)
    }
  }
  if (isV2) {
    // the tracked G where the K assembly of the next block (f3_assemble_K) and other kernels expect it: T-layout dump, row-major
    const int c = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = hf * 8 + (c >> 4) * 4 + (t >> 2), ln = 16 * (t & 3) + (c & 15);
      st->f3_G[e * 64 + ln] = g[t];
      if (k.last && 16 * hf + t < r && c < r) st->G[(16 * hf + t) * r + c] = g[t];
    }
  }
  for (int idx = tid - 256; idx < RB * r; idx += 256) { const int m = idx / r, c = idx - m * r; coef_store(k.Acoef + idx, L.sA[m * F3_AS + c]); }
}
