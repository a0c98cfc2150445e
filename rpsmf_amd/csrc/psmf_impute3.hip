// Version 3 of the masked column loop, for the small shapes of ExperimentImpute (d <= 80 rows, r <= 14; config D: 19 x 10).
//
// Everything in this loop is a chain of dependent r x r steps run by single waves, each alone on its SIMD: such a wave
// issues one instruction per 5-7 cycles whatever the instruction is, so a column costs what its critical wave has
// INSTRUCTIONS (tools/impute_prof.hip, and the instruction counts of the stamped segments in the ISA).  Version 2 spent
// them on moving data between waves and on selects; this version removes both:
//   * every wave forms the masked augmented Gram ITSELF: with d <= 32 it is at most eight float64 MFMAs (16x16x4, one per
//     group of four rows; twenty at d = 80, still less than the exchange it replaces), and its output layout is the layout the sweep and the trace <G, P + Q> want.  Version 2 spread
//     the MFMAs over three waves, exchanged partial tiles through LDS and had wave 0 reduce them and hand eta, N, phi to
//     the others (two barriers, ~2 700 cycles of wave 0's chain per column).  The augmented columns e and 1 are STORED in
//     the zero padding of C's LDS rows (columns r2, r2 + 1; r2 = r rounded up to even), so the operands are b = row,
//     a = m * row: the product holds G, b = C^T e (column r2), e^T e at (r2, r2) and sum(m) at (r2 + 1, r2 + 1) -- no
//     reductions for the last two.
//   * lane predicates as NUMBERS in VGPRs (0.0 / 1.0 multipliers, computed once): a select of a double is two v_cndmask
//     plus, here, the reload of its lane mask from a spilled SGPR pair (two v_readlane) -- the kernel had 1 200 of those;
//     a multiply-add is one instruction.  wave_sweep16m: 45 instructions per 2 x 2 pivot round instead of ~110.
//   * MFMAs in VGPR form (the build flag -amdgpu-mfma-vgpr-form: with 512 registers per wave on offer the compiler otherwise puts
//     the accumulators in AGPRs -- 16 copies and a 16-cycle stall per pivot round).
//   * C, the scalars, W and P + Q are double-buffered by column parity: what a column's tail writes is the NEXT column's
//     slot, so the only barriers are "residual / w / s ready" and "end of column".
//   * the sweep of M_t runs on the matrix AUGMENTED with kappa b = kappa C^T e (row / column r2 of the 16 x 16 tile, which the
//     rank-2 updates of the pivot rounds cover anyway): when the r2 pivots are done that column holds kappa P+ b = x_t - x_p
//     and its diagonal element 1 - kappa^2 b^T P+ b (for omega) -- no product P+ b afterwards.
//   P1   wave 2 (two lanes per row): masked residual rows, e into C's column r2; wave 3 (four lanes per row): w = V x,
//        s = x^T V x; wave 1: X of the previous column to global memory; waves 0, 1: Lbar_t from W_{t-1}        | barrier 1
//   G    every wave: operands LDS -> registers, Gram, the scalars it needs from it
//   wave 0: [M_t = Lbar_t + kappa G | kappa b] -> sweep -> x_t, omega, P_t;  publishes x_t, 1/omega, 1/q, rho, lambda, P_t + Q_{t+1}
//   wave 1: W_t = (M_t + I / q_t)^-1 beside it (Q = q I)
//   waves 2, 3: <G, P + Q>, eta, N, phi for themselves;  rank-1 updates C_t -> next buffer, V in place;  wave 2: bands and
//        metrics of its rows                                                                                     | barrier 2
// Same recursion, same float64 arithmetic as version 2 up to summation order (G: one accumulating MFMA chain instead of three
// partial tiles; e^T e on the matrix cores).  psmf_impute_run picks this kernel when the shape allows (PSMF_IMPUTE_V3=0:
// version 2).
#pragma once
#include "psmf_wave16.hip"     // wave_sweep16m: the single-wave sweep with the lane predicates as multipliers

namespace psmf {

inline bool impute3_ok(int d, int r) { return d <= 80 && r <= IR - 2; }
inline int impute3_groups(int d) { return d <= 12 ? 3 : (d <= 20 ? 5 : (d <= 32 ? 8 : (d <= 48 ? 12 : 20))); }     // template parameter NG: 4 NG rows of LDS

inline size_t impute3_lds_bytes(int d, int r) {
  const size_t d4 = 4 * (size_t)impute3_groups(d);
  const size_t doubles = 2 * d4 * IR + IR * IR + 3 * IR + 2 * d4 + 2 * 16 + 2 * 256 + 2 * 256 + 16 + 2;
  return (doubles * 8 + 15) & ~(size_t)15;
}

template <int WV, int NG>
__device__ __forceinline__ void impute3_wave(const ImputeParams& p) {
  int nbar = 0;          // barriers this wave has executed (imp_barrier_check, psmf_impute.hip)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int d = p.d, n = p.n, r = p.r, tid = threadIdx.x, rep = blockIdx.x;
  const int lane = tid & 63, lk = lane >> 4, lr = lane & 15;
  constexpr int D4 = 4 * NG;
  // ---- LDS carve (doubles); rows of C, V and the r-vectors padded to IR entries, C / e / m to D4 rows ----
  double* sC = sm;                            // 2 x D4 x IR: [C | e | 1 | 0 ...], by column parity
  double* sV = sC + 2 * D4 * IR;              // IR x IR
  double* sx = sV + IR * IR;                  // 2 x IR: prior mean of the current / next column
  double* sw = sx + 2 * IR;                   // V x
  double* se = sw + IR;                       // D4: masked residual
  double* smk = se + D4;                      // D4: mask as 0 / 1
  double* ssc = smk + D4;                     // 2 x 16 scalars: 0 s | 4 1 / omega_{t-1}, 5 1 / q_{t-1}, 6 1 / q_t | 8 rho_t, 9 lambda_t
  double* sW = ssc + 2 * 16;                  // 2 x 256: W_{t-1} = (M_{t-1} + I / q_{t-1})^-1, MFMA output layout
  double* sPP = sW + 2 * 256;                 // 2 x 256: P_{t-1} + Q_t, MFMA output layout (for <G, P + Q>)
  double* sred = sPP + 2 * 256;               // 16: end-of-pass reductions
  int* errflag = reinterpret_cast<int*>(sred + 16);

  const double* Yorg = p.Yorg;
  const uint8_t* Mk = p.M + (size_t)rep * n * d;
  const uint8_t* Mm = p.Mmiss + (size_t)rep * n * d;
  double* Cg = p.C + (size_t)rep * d * r;
  double* Xg = p.X + (size_t)rep * n * r;

  for (int idx = tid; idx < 2 * D4 * IR; idx += WG) {
    const int i = (idx >> 4) % D4, l = idx & 15;
    sC[idx] = (i < d && l < r) ? Cg[i * r + l] : ((i < d && l == r + (r & 1) + 1) ? 1.0 : 0.0);
  }
  for (int idx = tid; idx < IR * IR; idx += WG) { const int i = idx >> 4, l = idx & 15; sV[idx] = (i < r && l < r) ? p.V0[i * r + l] : 0.0; }
  if (tid < 2 * IR) sx[tid] = (tid < r) ? Xg[(size_t)(n - 1) * r + tid] : 0.0;   // t = 0 wraps to the last column (PSMF.py:65)
  if (tid < IR) sw[tid] = 0.0;
  if (tid < 2 * 16) ssc[tid] = 0.0;
  for (int idx = tid; idx < 2 * 256; idx += WG) { sW[idx] = 0.0; sPP[idx] = 0.0; }
  for (int idx = tid; idx < D4; idx += WG) { se[idx] = 0.0; smk[idx] = 0.0; }
  if (tid == 0) *errflag = 0;
  const double dd = (double)d, idd = 1.0 / dd;
  const bool sgd = p.method >= 2;     // MLE-SMF / TMF: gradient step on C along x_p, no V
  const bool tmf = p.method == 3;
  const bool par = p.q_iso && !tmf;   // Q = q I: the two inversions of a column are independent (see version 2)
  const int r2 = r + (r & 1);         // sweep size: even, identity-padded; also the tile column of e / of kappa b
  // where the scalar by-products of the Gram sit: element (r2, r2) = e^T e, (r2 + 1, r2 + 1) = sum(m)
  const int rq_e = r2 >> 2, ln_e = ((r2 & 3) << 4) | r2, rq_m = (r2 + 1) >> 2, ln_m = (((r2 + 1) & 3) << 4) | (r2 + 1);
  // lane predicates as multipliers: inside the r x r matrix; its diagonal; the identity padding's diagonal; the matrix plus
  // the augmented row / column r2 (b)
  double finq[4], fdgin[4], fpad[4], fga[4], fxr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = lk + 4 * q;
    fxr[q] = i < r ? 1.0 : 0.0;
    const bool in = i < r && lr < r, dg = i == lr;
    finq[q] = in ? 1.0 : 0.0;
    fdgin[q] = (in && dg) ? 1.0 : 0.0;
    fpad[q] = (!in && dg) ? 1.0 : 0.0;
    fga[q] = (in || (lr == r2 && i < r) || (i == r2 && lr < r)) ? 1.0 : 0.0;
  }
  const bool ce = lr == r2;
  Sw16K swk;
  if (WV < 2) sw16k_init(swk, lk, lr);
  // wave 0: P, Q in the MFMA output layout (element (lk + 4 q, lr)), rho, lambda, q
  double Pm[4] = {0.0, 0.0, 0.0, 0.0}, Qm[4] = {0.0, 0.0, 0.0, 0.0};
  double rho = p.rho0, lam = p.lambda0, qv = p.Q0[0], iqv = 1.0;
  if (WV == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = lk + 4 * q;
      const bool in = i < r && lr < r;
      const int a = in ? i * r + lr : 0, b = in ? lr * r + i : 0;
      Pm[q] = in ? 0.5 * (p.P0[a] + p.P0[b]) : 0.0;
      Qm[q] = in ? 0.5 * (p.Q0[a] + p.Q0[b]) : 0.0;
    }
  }
  bool bad = false;
  unsigned long long nmiss_l = 0;
  int cur = 0;
  // rows: d <= 32 (NG <= 8): wave 2, two lanes per row (columns 8 half .. 8 half + 7); wider shapes: one lane per row, rows
  // 0 .. 63 on wave 2 and 64 .. 79 on wave 1 (which has nothing else to do in that phase)
  constexpr bool WIDE = NG > 8;
  constexpr bool ROWS = WV == 2 || (WIDE && WV == 1);       // this wave owns rows
  const int rowi = WIDE ? (WV == 1 ? 64 + lane : lane) : (lane & 31);       // this lane's row
  const int row = min(rowi, d - 1), half = WIDE ? 0 : lane >> 5;
  const bool rown = ROWS && rowi < d && (WIDE || lane < 32);
  imp_barrier_full(nbar);
  IMP_T0();
  for (int it = 0; it < p.n_iter; ++it) {
    const double gam = 1e-6 / pow((double)(it + 1), 0.7);     // MLESMF.py:59-60, TMF.py:46-48
    if (WV == 0) {
      if (p.robust) {                 // rPSMF.py:77-79: Q, R, lambda restart every pass; V, P, C carry over
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = lk + 4 * q;
          const bool in = i < r && lr < r;
          const int a = in ? i * r + lr : 0, b = in ? lr * r + i : 0;
          Qm[q] = in ? 0.5 * (p.Q0[a] + p.Q0[b]) : 0.0;
        }
        rho = p.rho0;
        lam = p.lambda0;
        qv = p.Q0[0];
      }
      if (it == 0 || p.robust) {      // this column parity's slot of everything a column's tail publishes
        if (par) {
          // Lbar_0 = (P + q I)^-1 by one sweep, handed over as the W that reproduces it: W = q I - q^2 Lbar (omega = 1)
          double A[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = Pm[q] + fdgin[q] * qv + fpad[q];
          wave_sweep16m(A, r2, swk, bad);            // -(P + q I)^-1
#pragma unroll
          for (int q = 0; q < 4; ++q) sW[cur * 256 + q * 64 + lane] = fdgin[q] * qv + finq[q] * qv * qv * A[q];
          iqv = 1.0 / qv;
          if (lane == 0) { ssc[4] = 1.0; ssc[5] = iqv; ssc[6] = iqv; ssc[16 + 4] = 1.0; ssc[16 + 5] = iqv; ssc[16 + 6] = iqv; }   // (PSMF: constant)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double pp = tmf ? 0.5 * fdgin[q] : Pm[q] + (par ? fdgin[q] * qv : Qm[q]);   // TMF: P + Q := I / nu, nu = 2 (TMF.py:47,60)
          sPP[cur * 256 + q * 64 + lane] = pp;
          if (tmf) sPP[(cur ^ 1) * 256 + q * 64 + lane] = pp;
        }
        if (lane == 0) { ssc[8] = rho; ssc[9] = lam; ssc[16 + 8] = rho; ssc[16 + 9] = lam; }
      }
    }
    imp_barrier_full(nbar);
    double sse_pred = 0.0;
    unsigned long long inside_l = 0;
    nmiss_l = 0;
    // prefetch column 0 (wave 2; unconditional loads, row index clamped -- see version 2)
    double ny = 0.0;
    uint8_t nm = 0, nmm = 0;
    if (ROWS) { ny = Yorg[row]; nm = Mk[row]; nmm = Mm[row]; }
    for (int t = 0; t < n; ++t) {
      const double* sxc = sx + cur * IR;
      double* sxn = sx + (cur ^ 1) * IR;
      double* sCc = sC + cur * (D4 * IR);
      double* sCn = sC + (cur ^ 1) * (D4 * IR);
      const double* scc = ssc + cur * 16;
      double* scn = ssc + (cur ^ 1) * 16;
      double yv = 0.0, yh = 0.0;
      uint8_t mv = 0, mmv = 0;
      // ---- P1: residual rows (wave 2), w = V x and s = x^T V x (wave 3), X of the previous column (wave 1), Lbar (waves 0, 1) ----
      double Lb[4] = {0.0, 0.0, 0.0, 0.0};       // Lbar_t (+ the identity padding's diagonal)
      double iqt = 0.0, xc[4] = {0.0, 0.0, 0.0, 0.0};
      if (WV < 2) {
        if (WV == 1 && t > 0 && lane < r) Xg[(size_t)(t - 1) * r + lane] = sxc[lane];    // the reference overwrites X[:, t] in place
        if (par) {
          const double iom = scc[4], iq = scc[5];
          iqt = scc[6];
          const double k1 = iom * iq, k2 = k1 * iq;
#pragma unroll
          for (int q = 0; q < 4; ++q) Lb[q] = fma(-k2, sW[cur * 256 + q * 64 + lane], fma(fdgin[q], k1, fpad[q]));
        }
        if (WV == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xc[q] = sxc[lk + 4 * q];
        }
      }
      if (ROWS) {
        yv = ny; mv = nm; mmv = nmm;
        if (WIDE) {
          double cr[IR], xr[IR];
#pragma unroll
          for (int l = 0; l < IR; ++l) { cr[l] = sCc[row * IR + l]; xr[l] = sxc[l]; }   // (x is zero in the columns of e and 1)
          double d0 = 0.0, d1 = 0.0;
#pragma unroll
          for (int l = 0; l < IR; l += 2) { d0 = fma(cr[l], xr[l], d0); d1 = fma(cr[l + 1], xr[l + 1], d1); }
          yh = d0 + d1;
        } else {
          double cr[8], xr[8];
#pragma unroll
          for (int l = 0; l < 8; ++l) { cr[l] = sCc[row * IR + 8 * half + l]; xr[l] = sxc[8 * half + l]; }   // (x is zero in the columns of e and 1)
          double d0 = 0.0, d1 = 0.0;
#pragma unroll
          for (int l = 0; l < 8; l += 2) { d0 = fma(cr[l], xr[l], d0); d1 = fma(cr[l + 1], xr[l + 1], d1); }
          yh = xor32_sum_f64(d0 + d1);
        }
        const double mi = mv ? 1.0 : 0.0;
        const double yi = mv ? yv : 0.0;     // Y is 0 where unobserved (PSMF.py:147-148)
        const double ei = mi * (yi - yh);
        if (rown) { se[rowi] = ei; smk[rowi] = mi; sCc[rowi * IR + r2] = ei; }
      }
      if (WV == 3) {
        double vr[4], xr[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) { vr[l] = sV[lr * IR + 4 * lk + l]; xr[l] = sxc[4 * lk + l]; }     // (rows >= r of V are zero)
        const double wl = xor32_sum_f64(xor16_sum_f64(fma(vr[0], xr[0], vr[1] * xr[1]) + fma(vr[2], xr[2], vr[3] * xr[3])));   // every 16-lane row holds w
        if (lane < IR) sw[lane] = wl;
        const double sv = row_sum_f64_dpp(sxc[lr] * wl);
        if (lane == 0) ssc[cur * 16] = sv;
      }
      IMP_T(0);
      imp_barrier_lds(nbar);                                          // ---- barrier 1
      IMP_T(1);
      if (ROWS) {          // next column's inputs: issued here, off the path to barrier 1, a whole column before their use
        const size_t cbase = (size_t)min(t + 1, n - 1) * d;      // (the last column is simply loaded twice)
        ny = Yorg[cbase + row];
        nm = Mk[cbase + row];
        nmm = Mm[cbase + row];
      }
      // ---- operands -> registers ----
      double cv[NG], mk[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        cv[g] = sCc[(4 * g + lk) * IR + lr];
        mk[g] = smk[4 * g + lk];
      }
      const double s = scc[0], rho_t = scc[8], lam_t = scc[9];
      double PPv[4] = {0.0, 0.0, 0.0, 0.0};
      if (WV >= 2 || ROWS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) PPv[q] = sPP[cur * 256 + q * 64 + lane];
      }
      IMP_T(2);
      // ---- augmented masked Gram [C | e | 1]^T diag(m) [C | e | 1], two accumulators ----
      f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mk[g] * cv[g], cv[g], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(mk[g] * cv[g], cv[g], acc0, 0, 0, 0);
      }
      double G[4], Bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { Bq[q] = acc0[q] + acc1[q]; G[q] = finq[q] * Bq[q]; }   // b_i = (C^T e)_i: column r of rows i
      double ee = 0.0, msum = 0.0;
      if (WV >= 2 || ROWS || (WV == 0 && p.robust)) {
        const double ee_r = rq_e == 0 ? Bq[0] : (rq_e == 1 ? Bq[1] : (rq_e == 2 ? Bq[2] : Bq[3]));
        ee = readlane_f64(ee_r, ln_e);
      }
      if (WV >= 2 || ROWS) {
        const double ms_r = rq_m == 0 ? Bq[0] : (rq_m == 1 ? Bq[1] : (rq_m == 2 ? Bq[2] : Bq[3]));
        msum = readlane_f64(ms_r, ln_m);
      }
      // weights of the observed rows: PSMF / rPSMF 1 / (rho + s) (PSMF.py:71-72), MLE-SMF 1 / rho (MLESMF.py:70), TMF 1
      const double kappa = tmf ? 1.0 : fast_rcp(sgd ? rho_t : rho_t + s);
      // eta, N, phi: the updating waves form them for themselves
      double eta = 0.0, N = 0.0, phi = 1.0;
      if (WV >= 2 || ROWS) {
        double tr = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) tr = fma(G[q], PPv[q], tr);
        const double trGP = wave_sum_f64_dpp(tr);
        eta = (rho_t * msum + trGP) * idd;     // divide by d, not by #observed (PSMF.py:77)
        N = s + eta;
        if (p.robust) phi = (lam_t + ee * fast_rcp(N)) * fast_rcp(lam_t + dd);           // rPSMF.py:112-114 (e = 0 on unobserved rows)
      }
      IMP_T(3);
      if (WV == 0) {
        // ---- P+ = ((P + Q)^-1 + kappa G)^-1 on the matrix augmented with kappa b; x_t, omega, P, Q ----
        double A[4];
        if (par) {
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = fma(kappa, fga[q] * Bq[q], Lb[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = (tmf ? 0.5 * fdgin[q] : Pm[q] + Qm[q]) + fpad[q];
          wave_sweep16m(A, r2, swk, bad);                  // -(P + Q)^-1
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = fma(kappa, fga[q] * Bq[q], fpad[q] - finq[q] * A[q]);
        }
        wave_sweep16m(A, r2, swk, bad);                  // [[-P+, kappa P+ b], [., 1 - kappa^2 b^T P+ b]]
        IMP_T(4);
        if (ce) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sxn[lk + 4 * q] = fma(fxr[q], A[q], xc[q]);      // x_p + kappa P+ b (entries >= r stay zero)
        }
        if (p.robust) {
          const double a_r = rq_e == 0 ? A[0] : (rq_e == 1 ? A[1] : (rq_e == 2 ? A[2] : A[3]));
          const double nk2bPb = readlane_f64(a_r, ln_e) - 1.0;                      // -kappa^2 b^T P+ b
          const double omega = (lam + kappa * ee + nk2bPb) * fast_rcp(lam + dd);   // rPSMF.py:105
#pragma unroll
          for (int q = 0; q < 4; ++q) { Pm[q] = -omega * finq[q] * A[q]; Qm[q] *= omega; }
          const double io = fast_rcp(omega);
          if (par && lane == 0) { scn[4] = io; scn[5] = iqv; scn[6] = iqv * io; }
          iqv *= io;
          rho *= omega; lam += dd; qv *= omega;
          if (lane == 0) { scn[8] = rho; scn[9] = lam; }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) Pm[q] = -finq[q] * A[q];
        }
        if (!tmf) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sPP[(cur ^ 1) * 256 + q * 64 + lane] = Pm[q] + (par ? fdgin[q] * qv : Qm[q]);
        }
      } else if (WV == 1) {
        if (par) {
          // ---- W_t = (M_t + I / q_t)^-1 for the next column's Lbar, beside wave 0's inversion of M_t ----
          double A[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) A[q] = fma(kappa, G[q], fma(fdgin[q], iqt, Lb[q]));
          wave_sweep16m(A, r2, swk, bad);
#pragma unroll
          for (int q = 0; q < 4; ++q) sW[(cur ^ 1) * 256 + q * 64 + lane] = -finq[q] * A[q];
        }
        IMP_T(4);
        // ---- (wide shapes) bands, metrics of the rows this wave owns ----
        if (rown) {
          const double band = p.sig * sqrt(p.robust ? (s * (mv ? 1.0 : 0.0) + eta) : (sgd ? eta : N));   // rPSMF.py:121-123 / PSMF.py:83-84 / MLESMF.py:81-82
          const double lo = yh - band, hi = yh + band;
          if (mmv) {
            const double dl = yh - yv;
            sse_pred += dl * dl;
            nmiss_l += 1;
            if (it == p.n_iter - 1 && !tmf && yv < hi && lo < yv) inside_l += 1;
          }
          if (p.want_bands) {
            const size_t off = ((size_t)rep * n + t) * d + rowi;
            p.Yrec[off] = yh;
            p.YrecL[off] = lo;
            p.YrecH[off] = hi;
          }
        }
      } else {
        // ---- waves 2, 3: rank-1 updates with N, phi of this column: C into the next column's buffer, V in place ----
        IMP_T(4);
        const double wsc = fast_rcp(N);
        const double csc = tmf ? gam : gam * fast_rcp(eta);        // MLESMF.py:79, TMF.py:63
        const int ul = tid & 15, ui0 = (tid - 128) >> 4;            // column l, rows ui0 + 8 m
        if (ul < r) {
          const double cl = sgd ? sxc[ul] * csc : sw[ul] * wsc;
#pragma unroll
          for (int m = 0; m < (D4 + 7) / 8; ++m) {
            const int i = ui0 + 8 * m;
            if (i < d) sCn[i * IR + ul] = fma(se[i], cl, sCc[i * IR + ul]);
          }
          if (!sgd) {
            const double wl = sw[ul] * wsc;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const int i = ui0 + 8 * m;
              if (i < r) sV[i * IR + ul] = phi * (sV[i * IR + ul] - sw[i] * wl);
            }
          }
        }
        // ---- wave 2: bands, metrics of its rows ----
        if (rown) {
          const double band = p.sig * sqrt(p.robust ? (s * (mv ? 1.0 : 0.0) + eta) : (sgd ? eta : N));   // rPSMF.py:121-123 / PSMF.py:83-84 / MLESMF.py:81-82
          const double lo = yh - band, hi = yh + band;
          if (mmv) {
            const double dl = yh - yv;
            sse_pred += dl * dl;
            nmiss_l += 1;
            if (it == p.n_iter - 1 && !tmf && yv < hi && lo < yv) inside_l += 1;
          }
          if (p.want_bands) {
            const size_t off = ((size_t)rep * n + t) * d + rowi;
            p.Yrec[off] = yh;
            p.YrecL[off] = lo;
            p.YrecH[off] = hi;
          }
        }
      }
      cur ^= 1;
      IMP_T(5);
      imp_barrier_lds(nbar);                                          // ---- barrier 2
      IMP_T(6);
    }
    if (WV == 1 && lane < r) Xg[(size_t)(n - 1) * r + lane] = sx[cur * IR + lane];
    // ---- end of pass: RMSE of the one-step predictions, RMSE of C @ X, coverage ----
    imp_barrier_full(nbar);                 // (drains the X stores)
    double nm_d = (double)nmiss_l;
    const double sse_full = held_out_sse(sC + cur * (D4 * IR), IR, Xg, Yorg, Mm, d, n, r, tid);
    double v0 = wave_sum(sse_pred), v1 = wave_sum(sse_full), v2 = wave_sum(nm_d), v3 = wave_sum((double)inside_l);
    imp_barrier_full(nbar);
    if (lane == 0) { sred[WV * 4 + 0] = v0; sred[WV * 4 + 1] = v1; sred[WV * 4 + 2] = v2; sred[WV * 4 + 3] = v3; }
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
  for (int idx = tid; idx < d * r; idx += WG) { const int i = idx / r, l = idx - i * r; Cg[idx] = sC[cur * (D4 * IR) + i * IR + l]; }
  if (WV < 2 && bad) *errflag = 1;           // (benign race: every writer stores 1)
  imp_barrier_full(nbar);
  imp_barrier_check(nbar, errflag);
  if (tid == 0) p.err[rep] = *errflag;
  IMP_TOUT();
}

template <int NG>
__global__ __launch_bounds__(WG) void psmf_impute_kernel3(ImputeParams p) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wv == 0) impute3_wave<0, NG>(p);
  else if (wv == 1) impute3_wave<1, NG>(p);
  else if (wv == 2) impute3_wave<2, NG>(p);
  else impute3_wave<3, NG>(p);
}

inline const void* impute3_kernel(int d) {
  const int ng = impute3_groups(d);
  switch (ng) {
    case 3: return (const void*)psmf_impute_kernel3<3>;
    case 5: return (const void*)psmf_impute_kernel3<5>;
    case 8: return (const void*)psmf_impute_kernel3<8>;
    case 12: return (const void*)psmf_impute_kernel3<12>;
    default: return (const void*)psmf_impute_kernel3<20>;
  }
}

}  // namespace psmf
