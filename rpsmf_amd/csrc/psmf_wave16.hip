// Single-wave r x r routines on a 16 x 16 float64 tile in the MFMA output layout (lane l: column lr = l & 15, rows lk + 4 q,
// lk = l >> 4) -- used by the masked small-shape column loop (psmf_impute3.hip) and by the small-rank block filter (psmf_blk16.hip).
//
// Cost model of a wave that has its SIMD to itself (measured with knock-outs of a pivot round, tools/impute_prof.hip): it issues
// one instruction per 4-8 cycles whatever the instruction is -- a float64 VALU operation ~8, a 32-bit one ~4, a 16x16x4 float64
// MFMA ~76 including the wait for its result -- so these routines are written for INSTRUCTION COUNT: lane predicates are kept as
// 0.0 / 1.0 multipliers in VGPRs (a select of a double is two v_cndmask plus, in a kernel that has run out of SGPRs, the reload
// of its lane mask from a spilled SGPR pair: two v_readlane), and the library is built with -mllvm -amdgpu-mfma-vgpr-form
// (rpsmf_amd/build.py) so that the 256-thread kernels get their MFMAs with VGPR accumulators (left to itself the compiler, with 512
// registers per wave on offer, puts them in AGPRs: 16 copies and a 16-cycle stall per pivot round).
#pragma once
#include "psmf_blk3.hip"      // readlane_f64, DPP sums, f64x4

namespace psmf {

// (Sw16K / sw16k_init, the per-lane constants of the multiplier-form sweeps: psmf_ns.hip)

// wave_sweep16 (psmf_impute.hip) with the lane predicates as multipliers: A <- -A^-1 of the leading r2 x r2 block by 2 x 2
// SPD block pivots, the rank-2 update of a round on the matrix cores, the pivot block by v_readlane.
//   Ki = K^-1 of the pivot block;  t_j = Ki [u_j; w_j] (u, w = rows 2 j, 2 j + 1);  the MFMA's B operand holds -t in the
//   pivot rows' lanes (+Ki at the pivot columns, whose C input is zeroed), its A operand the pivot rows as they stand
//   (= the pivot columns, by symmetry);  afterwards the pivot rows are overwritten with t (-Ki inside the block).
__device__ __forceinline__ void wave_sweep16m(double (&A)[4], const int r2, const Sw16K& c, bool& bad) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (2 * j < r2) {                              // uniform
      const int k = 2 * j, h = j & 1, kq = j >> 1, b0 = h << 5, b1 = b0 + 16;
      const double rk = A[kq];
      const double ka = readlane_f64(rk, b0 | k), kb = readlane_f64(rk, b0 | (k + 1)), ke = readlane_f64(rk, b1 | (k + 1));
      const double det = ka * ke - kb * kb;
      bad |= !(ka > 0.0) | !(det > 0.0);
      // 1 / det: v_rcp_f64 (~1e-8) and ONE cubic step x (1 + e + e^2), e = 1 - det x -- three dependent operations
      // instead of the four of two Newton steps (this chain is the round's critical path)
      const double x0 = __builtin_amdgcn_rcp(det);
      const double e1 = fma(-det, x0, 1.0);
      const double dinv = fma(x0 * e1, 1.0 + e1, x0);
      // u_j (even row of the pair) and w_j (odd row) in both rows of each pair
      const unsigned lo = __double2loint(rk), hi = __double2hiint(rk);
      const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      const double uj = __hiloint2double(h2[0], l2[0]), wj = __hiloint2double(h2[1], l2[1]);
      // det * (row of Ki that belongs to this lane's pivot row): even row [ke, -kb], odd row [-kb, ka]
      const double cu = c.fu * ke - c.fw * kb, cw = c.fw * ka - c.fu * kb;
      const double u1 = fma(uj, c.fnp[j], c.pc0[j]), w1 = fma(wj, c.fnp[j], c.pc1[j]);     // pivot columns: unit vectors -> the entries of Ki
      const double pre = fma(cu, u1, cw * w1) * c.sg[j];       // everything of the B operand but 1 / det
      const double aop = rk * c.fpiv[h];
      const double bop = pre * dinv;
      f64x4 acc = {c.fnp[j] * A[0], c.fnp[j] * A[1], c.fnp[j] * A[2], c.fnp[j] * A[3]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) A[q] = acc[q];
      A[kq] = fma(acc[kq], c.fnpiv[h], -bop);      // the pivot rows: t; pivot block: -Ki
    }
  }
}

// (wave_sweep_tiles<NT>, the same sweep for NT x NT tiles in one wave's registers, lives in psmf_ns.hip: the block filters use it too)

// ------------------------------------------------------------------------------------------------------------
// Solve block of the per-step engine for r <= 32 (declared in psmf_kernels.hip, which psmf_sweep_solve's block 0 calls): the two
// r x r inversions of a step (psmf.py:140-165 in r x r form) as wave-local sweeps.
//   StepParams.solve_dual (random walk, Q = q I):  wave 0  P+ = M^-1, M = Lbar + kappa G;  wave 1  W = (M / beta + I / q)^-1, side by
//     side, Lbar carried from the previous step's W by the serial stage (ns_valid == 7); a run's first step: wave 0 alone, three sweeps.
//   otherwise:  wave 0  -Pbar^-1, then M = Pbar^-1 + kappa G (or the weighted Gram of a non-uniform R), P+ = M^-1.
// All other waves of the block retire at once.
// ------------------------------------------------------------------------------------------------------------
// masked step, block 0 of the sweep: what masked_prep_block(publish) does, by one lane of a wave of its own (the solve waves of the same
// block start at once) -- eta, N, kappa, the next Gram's step index, the step's (s, eta) for the bands
__device__ __forceinline__ void masked_prep_wave(const StepParams& p) {
  double sc[3];
  if ((threadIdx.x & 63) == 0) {        // (one lane: masked_prep_block's publishing branch looks at threadIdx.x == 0 -- do it here)
    DevState* st = p.st;
    const int r = p.r;
    double tr = 0.0;
    for (int w = 0; w < p.mg_ntr; ++w) tr += p.mg_tr[w];
    const int meth = p.masked_method;
    const double s = meth ? 0.0 : st->s, rho = st->rho;
    const double eta = (rho * p.mg[r * r] + tr) / (double)p.d;
    st->eta = eta;
    st->N = s + eta;
    st->kappa = meth == 3 ? 1.0 : fast_rcp(rho + s);
    st->kq = st->k + 1;
    if (p.sc_hist) {
      const long long t = st->k - p.series_t0;
      p.sc_hist[2 * t] = s;
      p.sc_hist[2 * t + 1] = eta;
    }
  }
  (void)sc;
}

template <int NT>
__device__ __forceinline__ void solve_block_wave_t(const StepParams& p) {
  DevState* st = p.st;
  const int r = p.r, r2 = r + (r & 1), w = threadIdx.x >> 6, lane = threadIdx.x & 63, lk = lane >> 4, lr = lane & 15;
  const bool dual = p.solve_dual != 0;
  if (p.mask && w == 2) { masked_prep_wave(p); return; }     // masked step: one more wave publishes eta, N, G, ... beside the solve
  if (w > 1) return;
  // Every load the block depends on is issued HERE, before the first branch on a loaded value (carried): the scalars, G and Lbar
  // arrive in ONE memory round trip instead of two (the block is a latency chain: 11 us of which the sweep itself is 4.7).
  // masked step: the Gram of the step is p.mg (reduced over workgroups and ranks), not yet in st->G -- the publishing wave writes it
  // there while this one reads it at the source (both triangles: the reduced partials are symmetric up to the order of one product);
  // kappa = 1 / (rho + s) (MLE-SMF: 1 / rho, TMF: 1) does not wait for eta either.  st->G and st->Lbar are written bitwise symmetric
  // by the serial stage: one load per element.
  const int nsv = st->ns_valid;
  const double l_rho = st->rho, l_s = st->s, l_kap = st->kappa, q0 = st->Q[0];
  double A[NT][NT][4], Gk[NT][NT][4], Ll[NT][NT][4];
  bool bad = false;
  auto at = [&](const double* Mx, const int i, const int c) { return 0.5 * (Mx[i * r + c] + Mx[c * r + i]); };
#define WS_FOR(body)                                                                   \
  _Pragma("unroll") for (int ti = 0; ti < NT; ++ti)                                    \
    _Pragma("unroll") for (int tj = 0; tj < NT; ++tj)                                  \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                  \
        const int i = 16 * ti + lk + 4 * q, c = 16 * tj + lr;                          \
        const bool in = i < r && c < r, pad = (i == c) && i >= r;                      \
        const int ic = in ? i : 0, cc = in ? c : 0;                                    \
        (void)pad; (void)ic; (void)cc;                                                 \
        body                                                                           \
      }
  WS_FOR({
    Gk[ti][tj][q] = p.rho_rows ? st->GR[ic * r + cc] : (p.mask ? at(p.mg, ic, cc) : st->G[ic * r + cc]);
    Ll[ti][tj][q] = dual ? st->Lbar[ic * r + cc] : 0.0;
  })
  const bool carried = dual && nsv == 7;
  if (w > (carried ? 1 : 0)) return;
  const double kappa = p.mask ? (p.masked_method == 3 ? 1.0 : fast_rcp(l_rho + (p.masked_method ? 0.0 : l_s))) : l_kap;
  Sw16K swk;
  sw16k_init(swk, lk, lr);          // the sweeps' lane predicates as multipliers (psmf_ns.hip)
  WS_FOR({ Gk[ti][tj][q] = in ? (p.rho_rows ? Gk[ti][tj][q] : kappa * Gk[ti][tj][q]) : 0.0; })
  if (carried) {
    const double iq = 1.0 / q0, ib = p.robust ? 1.0 / p.beta : 1.0;
    WS_FOR({
      const double mv = Ll[ti][tj][q] + Gk[ti][tj][q];
      A[ti][tj][q] = in ? (w == 0 ? mv : mv * ib + (i == c ? iq : 0.0)) : (pad ? 1.0 : 0.0);
    })
    wave_invert_tiles<NT>(A, r2, swk, bad);
    double* dst = w == 0 ? st->Pplus : st->XpY;
    WS_FOR({ if (in) dst[i * r + c] = -A[ti][tj][q]; })
  } else {
    WS_FOR({ A[ti][tj][q] = in ? at(st->Pbar, ic, cc) : (pad ? 1.0 : 0.0); })
    wave_invert_tiles<NT>(A, r2, swk, bad);                       // -Pbar^-1
    WS_FOR({ Gk[ti][tj][q] = in ? Gk[ti][tj][q] - A[ti][tj][q] : (pad ? 1.0 : 0.0); A[ti][tj][q] = Gk[ti][tj][q]; })     // Gk now holds M
    wave_invert_tiles<NT>(A, r2, swk, bad);                       // -P+
    WS_FOR({ if (in) st->Pplus[i * r + c] = -A[ti][tj][q]; })
    if (dual) {
      const double iq = 1.0 / q0, ib = p.robust ? 1.0 / p.beta : 1.0;
      WS_FOR({ A[ti][tj][q] = in ? Gk[ti][tj][q] * ib + (i == c ? iq : 0.0) : (pad ? 1.0 : 0.0); })
      wave_invert_tiles<NT>(A, r2, swk, bad);                     // -W
      WS_FOR({ if (in) st->XpY[i * r + c] = -A[ti][tj][q]; })
    }
  }
#undef WS_FOR
  if (__builtin_amdgcn_readfirstlane(__any((int)bad)) && lane == 0 && st->err == 0) st->err = (int)(st->k + 1);
}

__device__ void solve_block_wave(const StepParams& p) {
  if (p.r <= 16) solve_block_wave_t<1>(p);
  else solve_block_wave_t<2>(p);
}

// 33 <= r <= 64: the same on 3 x 3 (r <= 48) or 4 x 4 tiles.  Only in the 256-thread instances of psmf_sweep_solve (r > 32 runs no
// other): there a wave may hold 512 registers, which the 64-double tile arrays need -- inlined into the 512-thread instances they took
// the row-sweep workgroups' registers with them (docs/HISTORY.md, round 4).  Leaner on registers than solve_block_wave_t: operands are
// loaded where they are used (a second memory round trip on a run's first step only; the carried step still loads everything at once).
template <int NT>
__device__ __forceinline__ void solve_block_wave_big_t(const StepParams& p) {
  DevState* st = p.st;
  const int r = p.r, r2 = r + (r & 1), w = threadIdx.x >> 6, lane = threadIdx.x & 63, lk = lane >> 4, lr = lane & 15;
  const bool dual = p.solve_dual != 0;
  if (p.mask && w == 2) { masked_prep_wave(p); return; }
  if (w > 1) return;
  const int nsv = st->ns_valid;
  const double l_rho = st->rho, l_s = st->s, l_kap = st->kappa, q0 = st->Q[0];
  const bool carried = dual && nsv == 7;
  if (w > (carried ? 1 : 0)) return;
  const double kappa = p.mask ? (p.masked_method == 3 ? 1.0 : fast_rcp(l_rho + (p.masked_method ? 0.0 : l_s))) : l_kap;
  const double* __restrict__ gsrc = p.rho_rows ? st->GR : (p.mask ? p.mg : st->G);
  const double gsc = p.rho_rows ? 1.0 : kappa;
  const bool gsym = p.mask != nullptr;              // the masked step's reduced Gram: both triangles averaged
  const double iq = 1.0 / q0, ib = p.robust ? 1.0 / p.beta : 1.0;
  Sw16K swk;
  sw16k_init(swk, lk, lr);
  double A[NT][NT][4];
  bool bad = false;
#define WB_FOR(body)                                                                   \
  _Pragma("unroll") for (int ti = 0; ti < NT; ++ti)                                    \
    _Pragma("unroll") for (int tj = 0; tj < NT; ++tj)                                  \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                  \
        const int i = 16 * ti + lk + 4 * q, c = 16 * tj + lr;                          \
        const bool in = i < r && c < r, pad = (i == c) && i >= r;                      \
        const int ic = in ? i : 0, cc = in ? c : 0;                                    \
        (void)pad; (void)ic; (void)cc;                                                 \
        body                                                                           \
      }
#define WB_GK() ((gsym ? 0.5 * (gsrc[ic * r + cc] + gsrc[cc * r + ic]) : gsrc[ic * r + cc]) * gsc)
  if (carried) {
    WB_FOR({
      const double mv = st->Lbar[ic * r + cc] + WB_GK();
      A[ti][tj][q] = in ? (w == 0 ? mv : mv * ib + (i == c ? iq : 0.0)) : (pad ? 1.0 : 0.0);
    })
    wave_invert_tiles<NT>(A, r2, swk, bad);
    double* dst = w == 0 ? st->Pplus : st->XpY;
    WB_FOR({ if (in) dst[i * r + c] = -A[ti][tj][q]; })
  } else {
    WB_FOR({ A[ti][tj][q] = in ? 0.5 * (st->Pbar[ic * r + cc] + st->Pbar[cc * r + ic]) : (pad ? 1.0 : 0.0); })
    wave_invert_tiles<NT>(A, r2, swk, bad);                       // -Pbar^-1
    if (dual) {
      double Mx[NT][NT][4];
      WB_FOR({ Mx[ti][tj][q] = in ? WB_GK() - A[ti][tj][q] : (pad ? 1.0 : 0.0); A[ti][tj][q] = Mx[ti][tj][q]; })
      wave_invert_tiles<NT>(A, r2, swk, bad);                     // -P+
      WB_FOR({ if (in) st->Pplus[i * r + c] = -A[ti][tj][q]; })
      WB_FOR({ A[ti][tj][q] = in ? Mx[ti][tj][q] * ib + (i == c ? iq : 0.0) : (pad ? 1.0 : 0.0); })
      wave_invert_tiles<NT>(A, r2, swk, bad);                     // -W
      WB_FOR({ if (in) st->XpY[i * r + c] = -A[ti][tj][q]; })
    } else {
      WB_FOR({ A[ti][tj][q] = in ? WB_GK() - A[ti][tj][q] : (pad ? 1.0 : 0.0); })
      wave_invert_tiles<NT>(A, r2, swk, bad);                     // -P+
      WB_FOR({ if (in) st->Pplus[i * r + c] = -A[ti][tj][q]; })
    }
  }
#undef WB_GK
#undef WB_FOR
  if (__builtin_amdgcn_readfirstlane(__any((int)bad)) && lane == 0 && st->err == 0) st->err = (int)(st->k + 1);
}

__device__ __forceinline__ void solve_block_wave_big(const StepParams& p) {
  if (p.r <= 48) solve_block_wave_big_t<3>(p);
  else solve_block_wave_big_t<4>(p);
}

}  // namespace psmf
