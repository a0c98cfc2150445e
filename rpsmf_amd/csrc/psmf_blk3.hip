// Role-specialised coefficient-space filter of the blocked engine ("filter3") for the common
// configuration (full filter, random-walk dynamics, Q = q I, r <= 32; blocks of at most 48 timesteps): same recursion as
// psmf_blk_filter2 (psmf_block.hip), re-laid out around what bounds a step on MI355X.
//
// On gfx950 v_mfma_f64_16x16x4_f64 runs at the float64 VECTOR rate (64 cycles per instruction and SIMD),
// so a step is bound by the 2 x 32 MFMAs of each Newton-Schulz matrix product, by LDS operand traffic
// and by the number of barrier-separated phases -- not by flops elsewhere.  Hence:
//
//   * 8 waves, one ROLE each.  Waves 0-3 (one per SIMD) do nothing but the two r x r inversions of a
//     step (X: P+ = M^-1, Y: W = (M/beta + I/q)^-1, psmf_block.hip), TWO waves per inversion, wave c owning
//     the 16-column tile column c of its iterate.  Every matrix lives in registers in the MFMA OUTPUT
//     layout ("T-layout": tile (ti, tj), register q of lane l  <->  row 16 ti + (l >> 4) + 4 q,
//     column 16 tj + (l & 15)), which is at the same time the B-operand layout of the next product and
//     -- for a symmetric matrix -- the A-operand layout of the transposed tile.  So
//         R_c = I - M X_c            A = M (registers, built once per step), B = own column of X
//         X'_c = X_c + X^T R_c       A = X (own column + the partner's), B = R_c (registers)
//     need NO LDS operand reads; per iteration a wave publishes its new column (8 doubles per lane)
//     and reads its partner's: one barrier per iteration instead of two, 16 MFMAs per product and SIMD.
//     (X^T instead of X: the iterate is symmetric up to round-off and R' = (I - X M)^T R still
//     squares the residual.)
//   * waves 4-7 own the vector work, laid out so that no product needs a cross-lane reduction over
//     more than two lanes: V (lane = column, 16 rows), A by rows, KA by rows, A^T by columns.
//   * G, Lbar, the rank-2 / rank-1 updates and everything else O(r^2) is elementwise in registers and off
//     the critical path, which per step is:  w = V mu_bar, s, kappa (wave 4) | barrier | M, Newton-
//     Schulz iterations (one barrier each) | v = P+ h, mu (waves 0-1) | barrier.
//
// The direct symmetric sweep (psmf_kernels.hip) stays the fallback whenever the previous inverse is
// not a usable start (first step, transients, no convergence); it runs on all 8 waves through LDS images.
// Reference equations: pypsmf/psmf/psmf.py:104-165, rpsmf.py:116-171.
#pragma once
#include "psmf_block.hip"

namespace psmf {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));

constexpr int F3_NT = 512;
constexpr int F3_S = 34;            // row stride of the row-major 32 x 32 LDS images
constexpr int F3_MAXIT = 12;

#define F3_DPP64(x, ctrl)                                                                                  \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, true),                \
                   __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, true))

// sum over each 16-lane row, float64, DPP only; every lane of a row ends up with its row's sum
__device__ __forceinline__ double row_sum_f64_dpp(double v) {
  v += F3_DPP64(v, 0xB1);    // quad_perm [1,0,3,2]
  v += F3_DPP64(v, 0x4E);    // quad_perm [2,3,0,1]
  v += F3_DPP64(v, 0x141);   // row_half_mirror
  v += F3_DPP64(v, 0x140);   // row_mirror
  return v;
}

// (readlane_f64: psmf_ns.hip)

// sum over the 64 lanes, float64, fixed order; result uniform
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
  v = row_sum_f64_dpp(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Which T-layout elements lie inside the r x r matrix / on the diagonal.  16 < r <= 32: tile (0, 0) is always inside,
// column tile 1 is inside iff (l & 15) < r - 16, row tile 1 iff (l >> 4) + 4 q < r - 16.  r <= 16 (SMALL): only tile (0, 0)
// holds matrix elements -- column (l & 15) < r, row (l >> 4) + 4 q < r -- the rest of the 32 x 32 iterate is identity
// padding (the same kernel, same speed per timestep as r = 32: the r x r work of a step is latency, not throughput).
// Only the tiles (t, t) hold diagonal elements, where (l & 15) == (l >> 4) + 4 q.  Nine lane predicates (SGPR pairs) in all.
template <int MODE>       // 0: r == 32, nothing to mask (five predicates fewer to keep in SGPR pairs); 1: 16 < r < 32; 2: r <= 16
struct F3Mask {
  static constexpr bool full = MODE == 0;
  static constexpr bool small = MODE == 2;
  bool c1, r1[4], dg[4];      // MODE 2: c1 / r1 describe tile 0 (column / row < r)
};
#define F3_VALID(mk, ti, tj, q) ((mk).full || ((mk).small ? ((ti) == 0 && (tj) == 0 && (mk).r1[q] && (mk).c1) \
                                                          : (((ti) == 0 || (mk).r1[q]) && ((tj) == 0 || (mk).c1))))
#define F3_DIAG(mk, ti, tj, q) (((ti) == (tj)) && (mk).dg[q])

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: the y_hat coefficients and the
// mean history are stored to global memory inside the step loop, and every barrier behind such a store waited for its
// acknowledgement (~400 cycles per barrier, measured with the stamps).  Nothing in the loop READS global memory written
// in the loop, so LDS order is all the steps need.
__device__ __forceinline__ void f3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// x + (x of the lane 16 rows over) and x + (x of the lane in the other half): v_permlane16_swap / v_permlane32_swap
// (gfx950) exchange whole 16-lane rows between two registers in the VALU; ds_bpermute costs an LDS round trip.
__device__ __forceinline__ double xor16_sum_f64(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(h2[0], l2[0]) + __hiloint2double(h2[1], l2[1]);
}
__device__ __forceinline__ double xor32_sum_f64(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(h2[0], l2[0]) + __hiloint2double(h2[1], l2[1]);
}

// scalar slots
enum { F3_KAPPA = 0, F3_N, F3_INVN, F3_EE, F3_IOM, F3_Q, F3_IQ, F3_PSCALE, F3_NSC = 16 };

struct F3Lds {
  double* sK;       // RB x RB
  double* sA;       // RB x F3_AS  staging of the K assembly (Aprev)
  double* sKA;      // RB x F3_AS  (upper part of XG)
  double* dump;     // [2 parities][4 NS waves][8 registers][64 lanes]
  float* dumpP;     // [2 parities][4 NS waves][2 tiles][64 lane slots][4]: the same columns, float32, P-layout
  double* img;      // [2 inversions][32 x F3_S]
  double* mub;      // RM
  double* w;        // RM
  double* h;        // RM
  double* a;        // RB
  double* Ka;       // RB
  double* sc;       // F3_NSC
  double* nrm;      // [2][4] residual norms by dump parity and NS wave | 8: ns_tol2, 9: ns_far2, 10: ns_tol2 / 4
  double* hv;       // 2
  double* gp;       // 2
  double* tr;       // 2
  // start predictor of the two inversions (f3_ns_program phase 1; Z = P+ / W):
  double* sab;      // [2 inversions][32][2]  (a_j, b_j) = ((Z h)_j, (Z w)_j) of the step that just ended, by the wave that owns column j
  double* sal;      // [2][32]  alpha = T11 a + T12 b (wave 7, phase 0), ROW-PERMUTED: row 16 ti + lrow + 4 q at 8 lrow + 4 ti + q,
  double* sbe;      // [2][32]  beta  = T12 a + T22 b                    so that a lane's eight rows are 64 contiguous bytes
  float* s32;       // [2][128] the same in float32 for the A operands: alpha (32, permuted) | beta (32, permuted) | (a_j, b_j) pairs (64)
  double* rowbufX;  // 4 RM
  double* rowbufY;  // 4 RM
  int* errflag;
  long long* tick;  // 2: loop start / loop end (s_memrealtime), diagnostics
};

constexpr int F3_AS = 48;           // row stride of the K-assembly staging (16 mod 32: conflict-free MFMA operand reads)

// dynamic part (block Gram, assembly staging, sweep images and row buffers); the per-step vectors are static LDS
inline size_t blk_filter3_lds_bytes() {
  const size_t doubles = (size_t)RB * RB + 2 * (size_t)RB * F3_AS + 2 * 32 * F3_S + 8 * RM + 2;
  return (doubles * 8 + 15) & ~(size_t)15;
}

// K of a pipelined block from the previous block's quantities (BlockParams; same result as assemble_K in
// psmf_block.hip):  K[0:r,0:r] = G (tracked), K[r+q, r+q'] = Y^T Y (lower part of XG), and the cross block
// K[i, r+q] = sum_m Aprev[m, i] XG[m, q] on the float64 matrix cores: Aprev (RB x r) and the upper part of XG
// (RB x nb) staged once in LDS, one 16 x 16 output tile per wave (16 MFMAs).  Ends with a barrier.
// What changes from block to block of a chained launch (uniform scalars; a modified COPY of BlockParams per block sent
// the whole parameter set through VGPRs and scratch: 3.3 KB of spills, every phase 10-50 % slower).
struct F3Blk {
  long long k0;           // first step of the block is k0 + 1
  int nb, last;
  double* Acoef;
  double* Bcoef;
  const double* XG;
  const double* Aprev;    // nullptr: the previous block of the same launch left its coefficients in sA
};

__device__ __forceinline__ void f3_assemble_K(const BlockParams& b, const F3Blk& k, const F3Lds& L, const int r, const int tid) {
  const int nb = k.nb, lane = tid & 63, w = tid >> 6, lrow = lane >> 4, lcol = lane & 15;
  double* sK = L.sK;
  double* sA = L.sA;
  double* sX = L.sA + RB * F3_AS;
  const DevState* st = b.sp.st;
  // every global load of the assembly is issued before the first barrier (one L2 round trip, not two)
  double gv[2], yy[5];
#pragma unroll
  for (int u = 0; u < 2; ++u) gv[u] = st->f3_G[(w + 8 * u) * 64 + lane];     // tracked G of the previous block (T-layout dump:
                                                                               // an assembled block always follows a filter3 block)
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int idx = min(tid + u * F3_NT, nb * nb - 1), q = idx / nb, q2 = idx - q * nb;
    yy[u] = xg_load(k.XG + (size_t)(RB + q) * XGB + q2);
  }
  if (r + nb < RB)            // a full block writes every entry of K below
    for (int idx = tid; idx < RB * RB; idx += F3_NT) sK[idx] = 0.0;
  if (k.Aprev)                // (chained blocks: the previous block left its coefficients in sA)
    for (int idx = tid; idx < RB * 32; idx += F3_NT) {
      const int m = idx >> 5, c = idx & 31;
      const double v = k.Aprev[m * r + min(c, r - 1)];
      sA[m * F3_AS + c] = c < r ? v : 0.0;
    }
  for (int idx = tid; idx < RB * F3_AS; idx += F3_NT) {
    const int m = idx / F3_AS, q = idx - m * F3_AS;
    const double v = xg_load(k.XG + (size_t)m * XGB + min(q, nb - 1));
    sX[m * F3_AS + q] = q < nb ? v : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = w + 8 * u;
    const int row = 16 * (e >> 3) + lrow + 4 * (e & 3), col = 16 * ((e >> 2) & 1) + lcol;
    if (row < r && col < r) sK[row * RB + col] = gv[u];
  }
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int idx = tid + u * F3_NT;
    if (idx < nb * nb) { const int q = idx / nb, q2 = idx - q * nb; sK[(r + q) * RB + r + q2] = yy[u]; }
  }
  const int nct = (nb + 15) >> 4;
  if (w < 2 * nct) {
    const int ti = w & 1, ct = w >> 1;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < RB / 4; ++kk)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[(4 * kk + lrow) * F3_AS + 16 * ti + lcol], sX[(4 * kk + lrow) * F3_AS + 16 * ct + lcol], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 16 * ti + lrow + 4 * q, c = 16 * ct + lcol;
      if (i < r && c < nb) { sK[i * RB + r + c] = acc[q]; sK[(r + c) * RB + i] = acc[q]; }
    }
  }
  __syncthreads();
}

// Knock-out switches for tools/blk3_knock.hip (timing what each piece costs on the critical path; results are wrong
// with any of them set).  0 in the product: every `if` below folds away.
#ifndef F3_KNOCK
#define F3_KNOCK 0
#endif
// 0 compiles the start predictor out (tools/blk3_knock.hip times the fixed two-iteration schedule; with the predictor in,
// some knock-out masks run into a back-end error of this compiler: "Illegal instruction detected ... $src_shared_base")
#ifndef F3_PREDICT
#define F3_PREDICT 1
#endif

// One Newton-Schulz iteration of tile column C:  R = I - M X_c,  Xn = X_c + X^T R  (see the header).
// Mf: the step's matrix, T-layout, tile (kt, ti) at [(kt * 2 + ti) * 4 + kk]; Xc: own column (T-layout, float64).
// The correction X^T R is formed on the FLOAT32 matrix cores (v_mfma_f32_16x16x4_f32: 32 cycles against 64): R is small,
// so rounding X and R to float32 perturbs the new iterate by ~1e-7 ||X|| ||R|| -- 2e-9 after the first iteration, 1e-11
// or less after the last; the residual itself stays float64.  Xa: the A operands, float32, tile (kt, to) at
// [to * 8 + kt * 4 + kk] for output row tile to, in "P-layout" = T-layout with the 16 lanes of a row permuted by
// pi(4 a + v) = a + 4 v: the float32 MFMA returns row 4 (l >> 4) + v in register v where the float64 one returns
// (l >> 4) + 4 v, and feeding it the rows in that order makes its output land in T-layout.
// Returns this lane's share of ||R_c||_F^2.
template <int C, int FULL>
__device__ __forceinline__ double f3_ns_iter(const double (&Mf)[16], const double (&Xc)[8], const float (&Xa)[16], double (&Xn)[8],
                                             const F3Mask<FULL>& mk) {
  // the two 16 x 16 output tiles of a product are independent accumulator chains: alternate them, so that no MFMA
  // waits for the one before it
  f64x4 acc[2] = {f64x4{0.0, 0.0, 0.0, 0.0}, f64x4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
        acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Mf[(kt * 2 + ti) * 4 + kk], Xc[kt * 4 + kk], acc[ti], 0, 0, 0);
  double R[8];
  double nrm = 0.0;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double v = (F3_DIAG(mk, ti, C, q) ? 1.0 : 0.0) - acc[ti][q];
      R[ti * 4 + q] = v;
      nrm += v * v;
    }
  f32x4s d2[2] = {f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}};
  float Rf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) Rf[e] = (float)R[e];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int to = 0; to < 2; ++to)
        if (!(F3_KNOCK & 512)) d2[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xa[to * 8 + kt * 4 + kk], Rf[kt * 4 + kk], d2[to], 0, 0, 0);
#pragma unroll
  for (int to = 0; to < 2; ++to)
#pragma unroll
    for (int q = 0; q < 4; ++q) Xn[to * 4 + q] = Xc[to * 4 + q] + (double)d2[to][q];
  return nrm;
}

// Uniform bookkeeping of one step's inversion, identical in both programs (same LDS values, same decisions)
struct F3Ctl {
  bool have_prev;
  int ns_skip;
  int c_ns, c_sw, c_it, c_fail;
};

// residual norms of the iteration whose columns sit in dump parity `par` -> done / failed (uniform over the workgroup)
#define F3_DECIDE()                                                                                        \
  do {                                                                                                     \
    const double nx_ = L.nrm[par * 4 + 0] + L.nrm[par * 4 + 1];                                            \
    const double ny_ = L.nrm[par * 4 + 2] + L.nrm[par * 4 + 3];                                            \
    const double worst_ = fmax(nx_, ny_);                                                                  \
    ++ctl.c_it;                                                                                            \
    /* the thresholds are read from LDS beside the norms (L.nrm[8..10]): as kernel arguments they were kept in VGPRs,   */ \
    /* spilled, and reloaded from scratch in front of every one of these comparisons (scratch_load; s_waitcnt vmcnt(0)) */ \
    if (worst_ < L.nrm[8]) done = true;           /* ||R|| below the tolerance BEFORE the update just made */ \
    else if (!(worst_ < L.nrm[9]) || it == F3_MAXIT - 1) failed = true;   /* start too far or not converging */   \
    /* ||R_next||_F <= (||R||_F + ||M (Xc - Xa)||) ||R||_F: one more iteration is the last, no check needed */   \
    else last = worst_ * worst_ < L.nrm[10];                                                               \
    if (F3_KNOCK & 1) { done = false; failed = false; last = true; }      /* always exactly two iterations */ \
  } while (0)

// The direct symmetric sweep of both matrices on all 8 waves (half X: image X, half Y: image Y), in place in
// the images.  Called from both programs at the same point; the images were published before the barrier that
// precedes it.
__device__ __forceinline__ void f3_sweep_images(const F3Lds& L, const int r2, const int tid) {
  const int lt = tid & (WG - 1), c32 = lt & 31, rg = lt >> 5;
  const bool halfX = tid < WG;
  double* im = halfX ? L.img : L.img + 32 * F3_S;
  double A1[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) A1[m] = im[(rg + 8 * m) * F3_S + c32];
  sweep_all<32>(A1, r2, c32, rg, halfX ? L.rowbufX : L.rowbufY, L.errflag);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = rg + 8 * m;
    if (i < r2 && c32 < r2) im[i * F3_S + c32] = -A1[m];
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------
// Program of the four inversion waves.  INV 0 = X (P+ = M^-1), 1 = Y (W = (M / beta + I / q)^-1); C = own tile column.
// ------------------------------------------------------------------------------------------------------------
template <int C, int FULL>      // FULL: the F3Mask mode
__device__ __forceinline__ void f3_ns_program(const BlockParams& b, const F3Blk& k, const F3Lds& L, const int inv, const int role, const int lane,
                                              const bool carried) {
  const StepParams& p = b.sp;
  DevState* st = p.st;
  const int r = p.r, r2 = r + (r & 1), tid = 64 * role + lane;
  const int lrow = lane >> 4, lcol = lane & 15;
  const bool isX = inv == 0, isY = inv == 1;
  double* imgX = L.img;
  double* imgY = L.img + 32 * F3_S;
  double G[16], Wf[16], Xc[8];
  float Xa[16];                             // float32 A operands of the correction product (see f3_ns_iter)
  const int pcol = (lcol >> 2) + 4 * (lcol & 3);     // pi(lcol)       // Wf: W of the last step (zero outside r x r): Lbar = (I / q - W / q^2) / omega
  const double q0 = st->Q[0];
  F3Mask<FULL> mk;
  const int rt = FULL == 2 ? r : r - 16;               // extent of the partially filled tile (tile 0 when r <= 16, else tile 1)
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
        const bool in = row < r && col < r;
        const int e = (ti * 2 + tj) * 4 + qq;
        G[e] = in ? L.sK[row * RB + col] : 0.0;                              // G_0: exact Gram of the stored C / tracked G
        if (carried) {
          Wf[e] = st->f3_W[e * 64 + lane];
        } else {
          // the block starts from Lbar itself (just formed by the sweep): as a W, q I - q^2 Lbar
          const double l0 = imgX[row * F3_S + col];
          Wf[e] = in ? ((row == col ? q0 : 0.0) - q0 * q0 * l0) : 0.0;
        }
      }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = 16 * (e >> 2) + lrow + 4 * (e & 3);
    Xc[e] = carried ? st->f3_Xc[role][e * 64 + lane] : (row == 16 * C + lcol ? 1.0 : 0.0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = 16 * ((e >> 2) & 1) + lrow + 4 * (e & 3), cp = 16 * (e >> 3) + pcol;
    Xa[e] = carried ? st->f3_Xa[role][e * 64 + lane] : (row == cp ? 1.f : 0.f);
  }
  if (isX) {
    // <G_0, P> and tr G_0 of the own column, for eta of the first step (Pbar_1 = P + q I); carried: P = beta omega P+
    double g1 = 0.0, t1 = 0.0;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int row = 16 * ti + lrow + 4 * qq, col = 16 * C + lcol;
        const bool in = row < r && col < r;
        const double g = G[(ti * 2 + C) * 4 + qq];
        double pv;
        if (carried) pv = Xc[ti * 4 + qq];            // wave 4 applies beta omega
        else pv = st->P[in ? row * r + col : 0];
        g1 += in ? g * pv : 0.0;
        t1 += (row == col) ? g : 0.0;
      }
    g1 = wave_sum_f64_dpp(g1);
    t1 = wave_sum_f64_dpp(t1);
    if (lane == 0) { L.gp[C] = g1; L.tr[C] = t1; }
  }
  f3_barrier();                                                       // ---- init barrier

  // After a step's inversion: W (both columns) into Wf, and for the X waves <G_k, P+> and tr G_k of the own column
  // (eta of the next step).  Runs in the next step's phase 0, while wave 4 forms w, s and kappa, and once more after
  // the last step.  Lbar_{k+1} = (I / q - W / q^2) / omega_k is never formed: M is built from W directly (phase 1).
#define F3_W_AND_TRACES()                                                                                  \
  do {                                                                                                     \
    if (!(F3_KNOCK & 4))                                                                                   \
    _Pragma("unroll") for (int ti_ = 0; ti_ < 2; ++ti_)                                                    \
      _Pragma("unroll") for (int tj_ = 0; tj_ < 2; ++tj_) {                                                \
        double wv_[4];                                                                                     \
        if (isY && tj_ == C) {                                                                             \
          _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) wv_[q_] = Xc[ti_ * 4 + q_];                     \
        } else if (w_from_img) {                                                                           \
          _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) wv_[q_] = imgY[(16 * ti_ + lrow + 4 * q_) * F3_S + 16 * tj_ + lcol]; \
        } else {                                                                                           \
          const f64x2* d2_ = reinterpret_cast<const f64x2*>(L.dump) + (size_t)((w_par * 4 + 2 + tj_) * 4) * 64 + lane; \
          const f64x2 v0_ = d2_[(ti_ * 2) * 64], v1_ = d2_[(ti_ * 2 + 1) * 64];                            \
          wv_[0] = v0_[0]; wv_[1] = v0_[1]; wv_[2] = v1_[0]; wv_[3] = v1_[1];                               \
        }                                                                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                   \
          Wf[(ti_ * 2 + tj_) * 4 + q_] = F3_VALID(mk, ti_, tj_, q_) ? wv_[q_] : 0.0;                       \
      }                                                                                                    \
    BLK_T(6);                                                                                              \
    if (isX && !(F3_KNOCK & 2)) {                                                                          \
      double g1_ = 0.0;                                                                                    \
      _Pragma("unroll") for (int ti_ = 0; ti_ < 2; ++ti_)                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                   \
          g1_ += G[(ti_ * 2 + C) * 4 + q_] * Xc[ti_ * 4 + q_];     /* G is zero outside r x r */           \
      g1_ = wave_sum_f64_dpp(g1_);                                                                         \
      if (lane == 0) L.gp[C] = g1_;                                                                        \
    } else if (isY && !(F3_KNOCK & 2)) {                                                                   \
      /* tr G of the own column tile: every inversion wave holds all of G, the W waves have the shorter phase 0 */ \
      double t1_ = 0.0;                                                                                    \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) t1_ += F3_DIAG(mk, C, C, q_) ? G[(C * 2 + C) * 4 + q_] : 0.0; \
      t1_ = wave_sum_f64_dpp(t1_);                                                                         \
      if (lane == 0) L.tr[C] = t1_;                                                                        \
    }                                                                                                      \
  } while (0)

#define F3_ITERATE(parity_out)                                                                             \
  do {                                                                                                     \
    const double nr_ = f3_ns_iter<C, FULL>(Mf, Xc, Xa, Xn, mk);                                               \
    const float nw_ = wave_sum_f32_dpp((float)nr_);                                                        \
    f64x2* dp_ = reinterpret_cast<f64x2*>(L.dump) + (size_t)(((parity_out) * 4 + role) * 4) * 64 + lane;   \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) dp_[e_ * 64] = f64x2{Xn[2 * e_], Xn[2 * e_ + 1]};    \
    f32x4s* pp_ = reinterpret_cast<f32x4s*>(L.dumpP) + (size_t)(((parity_out) * 4 + role) * 2) * 64 + 16 * lrow + pcol; \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                                       \
      pp_[t_ * 64] = f32x4s{(float)Xn[t_ * 4], (float)Xn[t_ * 4 + 1], (float)Xn[t_ * 4 + 2], (float)Xn[t_ * 4 + 3]}; \
    if (lane == 0) L.nrm[(parity_out) * 4 + role] = (double)nw_;                                           \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) Xc[e_] = Xn[e_];                                      \
  } while (0)
#define F3_FETCH_PARTNER(parity_in)                                                                        \
  do {   /* both columns of the new iterate as float32 A operands: own tiles -> output tile C, partner's -> 1 - C */ \
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
  F3Ctl ctl = {carried, 0, 0, 0, 0, 0};
  int w_par = 0;
  bool w_from_img = false, fetch_late = false;
  double iq_w = carried ? st->f3_sc[0] : 1.0 / q0;          // 1 / q that Wf was formed with
  double kap_prev = 1.0;                                    // kappa of the step that just ended (start predictor)
  bool smw_ok = false;                                      // that step left a = Z h, b = Z w behind (not the first step of a block)
  BLK_T0();
  for (int jb = 0; jb < k.nb; ++jb) {
    // phase 0 (wave 4 forms w, s, kappa meanwhile): what the step that just ended left to do off the critical path
    if (fetch_late) F3_FETCH_PARTNER(w_par);     // (after BF: the partner published it in the last phase it ran)
    if (jb > 0) F3_W_AND_TRACES();
    BLK_T(7);
    f3_barrier();                                                     // ---- B1
    BLK_T(1);
    // =============================== phase 1: M, first iteration ===============================
    const bool try_ns = ctl.have_prev && p.use_ns && ctl.ns_skip == 0;
    if (!try_ns && ctl.ns_skip > 0) --ctl.ns_skip;
    double Mf[16], Xn[8];
    double hrow[8], wrow[8], h_j = 0.0, mub_j = 0.0, kap_k = 0.0;     // operands of phase F, loaded in phase 2
    int par = 0;
    {
      const double kap = L.sc[F3_KAPPA], iom = L.sc[F3_IOM], iq = L.sc[F3_IQ];
      kap_k = kap;
      const double ib = isY ? 1.0 / p.beta : 1.0, dq = isY ? iq : 0.0;
      // M = Lbar + kappa G with Lbar = (I / qw - W / qw^2) / omega, qw = the q that W was formed with; Y: M / beta + I / q
      const double c1 = iom * iq_w, c2 = c1 * iq_w;
      iq_w = iq;
      const double kb = kap * ib, cb = c2 * ib, dv = c1 * ib + dq;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int e = (ti * 2 + tj) * 4 + qq;
            const double m = kb * G[e] - cb * Wf[e];                 // zero outside r x r (G and Wf are)
            if (ti == tj) Mf[e] = m + (F3_DIAG(mk, ti, tj, qq) ? (F3_VALID(mk, ti, tj, qq) ? dv : 1.0) : 0.0);
            else Mf[e] = m;
          }
    }
    if (F3_PREDICT && try_ns && smw_ok && (p.ns_predict & 4)) {
      // Start of the iteration ("start predictor"): M_k differs from M_{k-1} by the rank-2 change of G
      // (h w^T + w h^T) / N + (ee / N^2) w w^T  -- downdated exactly, Sherman-Morrison-Woodbury with a = Z h, b = Z w left by
      // phase F and the 2 x 2 core folded into alpha, beta by wave 7 in phase 0 --  and, to first order, by the factor
      // kappa_k / kappa_{k-1} on everything (kappa G dominates M):  Z_0 = (kappa_{k-1} / kappa_k) (Z - alpha a^T - beta b^T).
      // Leaves ||I - M Z_0|| ~ 1e-4 .. 1e-3 where the plain start Z leaves 1e-2 .. 1e-1 (and > 1 through the first ~400 steps).
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
      // (the identity padding of r < 32 keeps its ones: alpha, a vanish there, only the scale has to be kept off it)
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
    if (try_ns) F3_ITERATE(0);
    BLK_T(2);
    f3_barrier();                                                     // ---- B2
    BLK_T(1);
    // =============================== phase 2: second iteration, G update ===============================
    bool done = false, failed = !try_ns, last = false;
    int it = 0;
    fetch_late = false;              // converged: the partner's final column is only needed by the NEXT step -> its phase 0
    if (try_ns) {
      F3_FETCH_PARTNER(0);           // issued with the norms' reads: one LDS round trip for both (harmless if the start failed:
      F3_DECIDE();                   //  the sweep path reloads the operands)
      if (!done && !failed) {
        if (!(F3_KNOCK & 256)) F3_ITERATE(1);
        par = 1;
        it = 1;
        if (last) { done = true; fetch_late = true; ++ctl.c_it; }
      }
    }
    {
      // G_k = G_{k-1} + (h w^T + w h^T) / N + ee w w^T / N^2   (tracked Gram)
      //     = G_{k-1} + u w^T + w hn^T,  u = h / N + (ee / N^2) w,  hn = h / N:  two FMAs per element
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
          for (int tj = 0; tj < 2; ++tj)
            if (!(F3_KNOCK & 16)) G[(ti * 2 + tj) * 4 + qq] += ui * wcol[tj] + wi * hn[tj];
          hrow[ti * 4 + qq] = hi;
          wrow[ti * 4 + qq] = wi;
        }
      h_j = hcol[C];
      mub_j = L.mub[16 * C + lcol];
    }
    BLK_T(3);
    // =============================== further iterations (only when the outcome is not known yet) ===============================
    if (!done) {
      f3_barrier();                                                   // ---- B3
      BLK_T(1);
      while (try_ns && !done && !failed) {
        F3_DECIDE();
        if (done) fetch_late = true;
        if (done || failed) break;
        F3_FETCH_PARTNER(par);
        F3_ITERATE(par ^ 1);
        par ^= 1;
        ++it;
        if (last) { done = true; fetch_late = true; ++ctl.c_it; break; }
        f3_barrier();
      }
    }
    BLK_T(4);
    // `par` = parity of the dump that holds the columns published by the last iteration that ran
    const bool from_img = !done;
    if (!done) {
      if (try_ns) { ++ctl.c_fail; ctl.ns_skip = p.ns_skip_n; }
      ++ctl.c_sw;
      double* im = isX ? imgX : imgY;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) im[(16 * ti + lrow + 4 * qq) * F3_S + 16 * C + lcol] = Mf[(ti * 2 + C) * 4 + qq];
      f3_barrier();
      f3_sweep_images(L, r2, tid);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int row = 16 * ti + lrow + 4 * qq;
          Xc[ti * 4 + qq] = im[row * F3_S + 16 * C + lcol];
#pragma unroll
          for (int to = 0; to < 2; ++to) Xa[to * 8 + ti * 4 + qq] = (float)im[row * F3_S + 16 * to + pcol];
        }
    } else {
      ++ctl.c_ns;
    }
    ctl.have_prev = true;
    // =============================== phase F: v = P+ h, mu (the only work between the inversion and the next step) ===============================
    w_par = par;
    w_from_img = from_img;
    if (!(F3_KNOCK & 8)) {
      // a = Z h, b = Z w of the own column (by symmetry) for the next step's start predictor; X waves: a is v = P+ h,
      // mu_k = mu_bar + kappa v (psmf.py:155-159); h.v for omega (rPSMF only)
      double vp0 = 0.0, vp1 = 0.0, vq0 = 0.0, vq1 = 0.0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        vp0 += Xc[qq] * hrow[qq]; vp1 += Xc[4 + qq] * hrow[4 + qq];
        vq0 += Xc[qq] * wrow[qq]; vq1 += Xc[4 + qq] * wrow[4 + qq];
      }
      const double vp = xor32_sum_f64(xor16_sum_f64(vp0 + vp1));
      // (the lane index made opaque here: the LDS / global addresses and the lane predicate of the stores below are then formed
      //  on the spot -- three integer instructions -- instead of being hoisted out of the step loop, spilled, and reloaded from
      //  scratch right in front of the barrier that ends the step: scratch_load; s_waitcnt vmcnt(0); ds_write, three times per step)
      int lane_f = lane;
      asm volatile("" : "+v"(lane_f));
      const int lcol = lane_f & 15, lrow = lane_f >> 4;
      const int j = 16 * C + lcol;
      if (isX) {
        const double mu_new = mub_j + kap_k * vp;
        if (lrow == 0 && j < r) {
          L.mub[j] = mu_new;                    // random walk: mu_bar_{k+1} = mu_k
          if (p.mu_hist) p.mu_hist[(size_t)(k.k0 + jb + 1 - p.series_t0) * r + j] = mu_new;
        }
        if (p.robust) {
          const double hvp = wave_sum_f64_dpp((lrow == 0) ? h_j * vp : 0.0);
          if (lane == 0) L.hv[C] = hvp;
        }
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
  if (fetch_late) F3_FETCH_PARTNER(w_par);
  if (k.nb > 0) F3_W_AND_TRACES();
  f3_barrier();                       // wave 4 has published pscale and 1 / omega of the last step
  const double ps = L.sc[F3_PSCALE], iom = L.sc[F3_IOM];
  // the state for the next block, as held: coalesced rows
#pragma unroll
  for (int e = 0; e < 8; ++e) st->f3_Xc[role][e * 64 + lane] = Xc[e];
#pragma unroll
  for (int e = 0; e < 16; ++e) st->f3_Xa[role][e * 64 + lane] = Xa[e];
  if (role == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) st->f3_G[e * 64 + lane] = G[e];
    if (lane == 0) st->f3_sc[0] = iq_w;
  }
  if (role == 2) {
#pragma unroll
    for (int e = 0; e < 16; ++e) st->f3_W[e * 64 + lane] = Wf[e];
  }
  if (k.last) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int row = 16 * ti + lrow + 4 * qq, col = 16 * C + lcol;
        if (row < r && col < r) {
          const int idx = row * r + col;
          if (isX) {
            st->XpX[idx] = Xc[ti * 4 + qq];
            st->P[idx] = ps * Xc[ti * 4 + qq];
            st->G[idx] = G[(ti * 2 + C) * 4 + qq];
          } else {
            st->XpY[idx] = Xc[ti * 4 + qq];
            st->Lbar[idx] = ((row == col ? iq_w : 0.0) - Wf[(ti * 2 + C) * 4 + qq] * iq_w * iq_w) * iom;
          }
        }
      }
  }
  if (role == 0 && lane == 0) { st->cnt[0] += ctl.c_ns; st->cnt[1] += ctl.c_sw; st->cnt[2] += ctl.c_it; st->cnt[3] += ctl.c_fail; }
#undef F3_ITERATE
#undef F3_FETCH_PARTNER
#undef F3_W_AND_TRACES
}

// ------------------------------------------------------------------------------------------------------------
// Program of the four vector waves: 4 = V and every scalar, 5 = A by rows, 6 = KA by rows, 7 = A^T by columns.
// ------------------------------------------------------------------------------------------------------------
template <int ROLE>
__device__ __forceinline__ void f3_v_program(const BlockParams& b, const F3Blk& k, const F3Lds& L, const int lane, const bool carried) {
  const StepParams& p = b.sp;
  DevState* st = p.st;
  constexpr int role = ROLE;
  const int r = p.r, r2 = r + (r & 1), tid = 64 * role + lane;
  const double dd = (double)p.d, idd = 1.0 / dd;
  constexpr bool isV0 = ROLE == 4, isV1 = ROLE == 5, isV2 = ROLE == 6, isV3 = ROLE == 7;
  // Each vector wave shares its SIMD with an inversion wave.  At equal priority its ~60 VALU instructions per phase were
  // issued about one per MFMA (64 cycles) and the barrier waited for THEM (3.7k cycles against 3.1k, stamps); with
  // priority they issue back to back and cost the MFMA stream a few hundred cycles instead.
  __builtin_amdgcn_s_setprio(3);
  // persistent registers:  V0: [0,16) V[16 hf + t][j]   V1: A[m][c]   V2: KA[m][c]   V3: A[32 hf + t][c]
  double pr[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) pr[i] = 0.0;
  // wave 4's running scalars (every lane holds the same values)
  double kappa = 0.0, Nk = 0.0, invN = 0.0, s_k = 0.0, eta_k = 0.0, ee_k = 0.0, phi = 1.0, omega = 1.0, pscale = 1.0, wj = 0.0;
  double q = st->Q[0];                  // Q = q I (checked by the host)
  double rho = st->rho, lam = st->lam;
  double iom0 = 1.0;                    // 1 / omega of the last step of the previous block (carried W is not yet divided by it)
  double kap7 = 1.0;                    // wave 7: kappa of the step that just ended
  if (isV0) {
    const int j = lane & 31, hf = lane >> 5;
    if (carried) {
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] = st->f3_V[t * 64 + lane];
      iom0 = st->f3_sc[1];
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

  // scalars of the step that just ended (omega needs v = P+ h): rpsmf.py:155-171
#define F3_V0_FINISH_PREV()                                               \
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

  F3Ctl ctl = {carried, 0, 0, 0, 0, 0};
  if (role == 4 && lane == 0) L.tick[0] = (long long)__builtin_amdgcn_s_memrealtime();
  BLK_T0();
  for (int jb = 0; jb < k.nb; ++jb) {
    // =============================== phase 0 ===============================
    double cm = 0.0;                       // V1: a_m, V2: (K a)_m  (kept for the rank-1 update of phase 2)
    if (isV0 && !(F3_KNOCK & 1024)) {
      const int j = lane & 31, hf = lane >> 5;
      double iom = iom0;
      if (jb > 0) { F3_V0_FINISH_PREV(); iom = fast_rcp(omega); }
      double part0 = 0.0, part1 = 0.0;
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        part0 += pr[t] * L.mub[16 * hf + t];
        part1 += pr[t + 1] * L.mub[16 * hf + t + 1];
      }
      const double part = part0 + part1;
      s_k = wave_sum_f64_dpp(part * L.mub[j]);                          // s = mu_bar^T V mu_bar: both halves hold their 16 rows' share
      kappa = fast_rcp(rho + s_k);
      if (lane == 0) { L.sc[F3_KAPPA] = kappa; L.sc[F3_IOM] = iom; L.sc[F3_Q] = q; L.sc[F3_IQ] = fast_rcp(q); }
      wj = part;                                                        // this half's share of w_j; completed in phase 1
    } else if ((isV1 || isV2) && !(F3_KNOCK & 32)) {
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
        cm = (lane == r + jb ? 1.0 : 0.0) - dot;                        // a_j = u_{r+j} - A mu_bar
        L.a[lane] = cm;
        coef_store(k.Bcoef + (size_t)jb * RB + lane, dot);              // y_hat_j = Z b_j
      } else {
        cm = L.sK[lane * RB + r + jb] - dot;                            // K a_j
        L.Ka[lane] = cm;
      }
    }
    else if (F3_PREDICT && isV3 && (p.ns_predict & 2)) {
      // Start predictor of the two inversions (f3_ns_program, phase 1): the 2 x 2 core of the rank-2 downdate.  With a = Z h,
      // b = Z w (left by the inversion waves in phase F), U = [h w] and G_k - G_{k-1} = U K U^T, K = [[0, 1/N], [1/N, ee/N^2]]:
      //   (M + kappa U K U^T)^-1 = Z - [a b] T [a b]^T,   T = ((kappa K)^-1 + U^T Z U)^-1,   (kappa K)^-1 = [[-ee, N], [N, 0]] / kappa
      // (W: M / beta, so kappa / beta).  Published as alpha = T11 a + T12 b, beta = T12 a + T22 b; lanes 0-31: P+, 32-63: W.
      const int j = lane & 31, hf = lane >> 5;
      double al = 0.0, be = 0.0;
      f64x2 ab7 = {0.0, 0.0};
      if (jb > 0) {
        ab7 = *reinterpret_cast<const f64x2*>(L.sab + 2 * lane);
        const double a = ab7[0], bb = ab7[1], hj = L.h[j], wv = L.w[j];
        const double ha = xor16_sum_f64(row_sum_f64_dpp(hj * a)), hb = xor16_sum_f64(row_sum_f64_dpp(hj * bb));
        const double wb = xor16_sum_f64(row_sum_f64_dpp(wv * bb));
        const double ik = (hf ? p.beta : 1.0) * fast_rcp(kap7);
        const double s11 = ha - L.sc[F3_EE] * ik, s12 = hb + L.sc[F3_N] * ik;
        const double idet = fast_rcp(s11 * wb - s12 * s12);
        const double t11 = wb * idet, t12 = -s12 * idet, t22 = s11 * idet;
        al = t11 * a + t12 * bb;
        be = t12 * a + t22 * bb;
      }
      const int pj = 32 * hf + 8 * (j & 3) + 4 * (j >> 4) + ((j >> 2) & 3);
      L.sal[pj] = al;
      L.sbe[pj] = be;
      float* f32b = L.s32 + 128 * hf;
      f32b[pj - 32 * hf] = (float)al;
      f32b[32 + pj - 32 * hf] = (float)be;
      *reinterpret_cast<f32x2s*>(f32b + 64 + 2 * j) = f32x2s{(float)ab7[0], (float)ab7[1]};
    }
    BLK_T(0);
    f3_barrier();                                                     // ---- B1
    BLK_T(1);
    // =============================== phase 1 ===============================
    const bool try_ns = ctl.have_prev && p.use_ns && ctl.ns_skip == 0;
    if (!try_ns && ctl.ns_skip > 0) --ctl.ns_skip;
    int par = 0;
    if (isV3) kap7 = L.sc[F3_KAPPA];                                    // (stable from B1 to the next step's phase 0)
    if (isV3 && !(F3_KNOCK & 64)) {
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
      // ee = a . K a: wave 5 is idle in this phase, wave 7's product above is the one the barrier waits for
      const double e1 = wave_sum_f64_dpp(cm * L.Ka[lane]);
      if (lane == 0) L.sc[F3_EE] = e1;
    } else if (isV0) {
      wj = xor32_sum_f64(wj);                                           // w = V mu_bar
      if (lane < 32) L.w[lane] = wj;
      // eta, N (psmf.py:121-128): <G, Pbar> and tr G were left by the X waves in phase 0
      const double gpv = pscale * (L.gp[0] + L.gp[1]) + q * (L.tr[0] + L.tr[1]);
      eta_k = rho + gpv * idd;
      Nk = s_k + eta_k;
      invN = fast_rcp(Nk);
      if (lane == 0) { L.sc[F3_N] = Nk; L.sc[F3_INVN] = invN; }
    }
    BLK_T(2);
    f3_barrier();                                                     // ---- B2
    BLK_T(1);
    // =============================== phase 2 ===============================
    bool done = false, failed = !try_ns, last = false;
    int it = 0;
    if (try_ns) {
      F3_DECIDE();
      if (!done && !failed) {
        par = 1;
        it = 1;
        if (last) done = true;
      }
    }
    if (F3_KNOCK & 128) {
    } else if (isV3) {
      const int c = lane & 31, hf = lane >> 5;
      const double wn = L.w[c] * L.sc[F3_INVN];
#pragma unroll
      for (int t = 0; t < 32; ++t) pr[t] += L.a[32 * hf + t] * wn;
    } else if (isV1 || isV2) {
      // rank-1 updates of A (rows) and KA (rows): psmf.py:130-133 in coefficient space
      const double cn = cm * L.sc[F3_INVN];
#pragma unroll
      for (int c = 0; c < 32; ++c) pr[c] += cn * L.w[c];
    } else if (isV0) {
      const int hf = lane >> 5;
      ee_k = L.sc[F3_EE];
      phi = 1.0;
      double vscale = 1.0;
      if (p.robust) {
        phi = (lam + ee_k * invN) * fast_rcp(lam + dd);                 // rpsmf.py:133-138
        vscale = p.alpha * phi;
      }
      const double wjn = wj * invN;
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] = vscale * (pr[t] - L.w[16 * hf + t] * wjn);   // psmf.py:135-138
    }
    BLK_T(3);
    if (!done) {
      f3_barrier();                                                   // ---- B3
      while (try_ns && !done && !failed) {
        F3_DECIDE();
        if (done || failed) break;
        par ^= 1;
        ++it;
        if (last) { done = true; break; }
        f3_barrier();
      }
    }
    if (!done) {
      if (try_ns) { ++ctl.c_fail; ctl.ns_skip = p.ns_skip_n; }
      ++ctl.c_sw;
      f3_barrier();
      f3_sweep_images(L, r2, tid);
    } else {
      ++ctl.c_ns;
    }
    ctl.have_prev = true;
    BLK_T(4);
    f3_barrier();                                                     // ---- BF
    BLK_T(1);
  }
  BLK_TOUT();
  if (role == 4 && lane == 0) L.tick[1] = (long long)__builtin_amdgcn_s_memrealtime();

  // ---- block end ----
  if (isV0) {
    if (k.nb > 0) F3_V0_FINISH_PREV();
    if (lane == 0) { L.sc[F3_PSCALE] = pscale; L.sc[F3_IOM] = fast_rcp(omega); L.sc[F3_Q] = q; }
  } else if (isV1) {
#pragma unroll
    for (int c = 0; c < 32; ++c) L.sA[lane * F3_AS + c] = pr[c];       // (a strided global store per column took ~4 us)
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
      st->Q[j * r + j] = q;             // Q = q I: the diagonal is what the next block reads (Q[0])
    }
    if (lane < r) st->mu[lane] = L.mub[lane];
    if (lane == 0) {
      st->f3_sc[1] = L.sc[F3_IOM]; st->f3_sc[2] = pscale;
      st->k = k.k0 + k.nb;
      st->rho = rho; st->lam = lam; st->phi = phi; st->omega = omega; st->ee = ee_k;
      st->s_done = s_k; st->eta_done = eta_k; st->N_done = Nk;
      if (*L.errflag && st->err == 0) st->err = (int)(k.k0 + 1);
      st->ns_valid = 3;
    }
  }
  // A_B (wave 5's rows) left through LDS before the barrier above: all four vector waves store it, coalesced
  for (int idx = tid - 256; idx < RB * r; idx += 256) { const int m = idx / r, c = idx - m * r; coef_store(k.Acoef + idx, L.sA[m * F3_AS + c]); }
#undef F3_V0_FINISH_PREV
}

#include "psmf_blk4.hip"      // filter4: the same skeleton for diagonal-Jacobian dynamics (sequential inversions); uses everything above

// SMALL = true: the r <= 16 instantiation, a kernel of its own -- compiled into the same kernel as the two r > 16 programs it
// cost the r = 32 path 2 % (register allocation over the larger kernel: 111 spilled registers against 96; measured A / B on one box)
// KIND 0: filter3 (random walk, two parallel inversions); KIND 1: filter4, KIND 2: filter5 (simplified hooks) (psmf_blk4.hip)
template <bool SMALL, int KIND = 0>
__device__ __forceinline__ void blk_filter3_body(const BlockParams& b0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const StepParams& p = b0.sp;
  DevState* st = p.st;
  const int r = p.r, tid0 = threadIdx.x;
  const int r2 = r + (r & 1);
  // ---- LDS carve ----
  // Everything a step touches sits in STATIC LDS: its addresses are compile-time constants that fold into the ds
  // instructions' immediate offsets.  (Off the dynamic-LDS base the compiler formed (lane part + constant) + base for
  // every row / column it reads and kept each sum in a VGPR of its own across the loop: 136 spilled registers.)
  __shared__ __attribute__((aligned(16))) double hot[2 * 4 * 8 * 64 + 3 * RM + 2 * RB + F3_NSC + 12 + 6 + 8 * 32];
  __shared__ __attribute__((aligned(16))) float hotP[2 * 4 * 2 * 64 * 4];
  __shared__ __attribute__((aligned(16))) float hotS[2 * 128];
  F3Lds L;
  L.sK = sm;
  L.sA = L.sK + RB * RB;
  L.sKA = L.sA + RB * F3_AS;
  L.img = L.sKA + RB * F3_AS;
  L.rowbufX = L.img + 2 * 32 * F3_S;
  L.rowbufY = L.rowbufX + 4 * RM;
  L.errflag = reinterpret_cast<int*>(L.rowbufY + 4 * RM);
  L.dump = hot;
  L.dumpP = hotP;
  L.mub = L.dump + 2 * 4 * 8 * 64;
  L.w = L.mub + RM;
  L.h = L.w + RM;
  L.a = L.h + RM;
  L.Ka = L.a + RB;
  L.sc = L.Ka + RB;
  L.nrm = L.sc + F3_NSC;
  L.hv = L.nrm + 12;
  L.gp = L.hv + 2;
  L.tr = L.gp + 2;
  L.sab = L.tr + 2;
  L.sal = L.sab + 128;
  L.sbe = L.sal + 64;
  L.s32 = hotS;
  __shared__ long long s_tick[2];
  L.tick = s_tick;
  __shared__ __attribute__((aligned(16))) double hot4[KIND >= 1 ? 5 * RM + 2 * 48 + F4_NKC : 2];
  F4Lds D;
  D.fd = hot4; D.mu = D.fd + RM; D.tp = D.mu + RM; D.th = D.tp + RM; D.rs = D.th + 2 * RM; D.qs = D.rs + 48; D.kc = D.qs + 48;
  if (tid0 == 0) { L.nrm[8] = p.ns_tol2; L.nrm[9] = p.ns_far2; L.nrm[10] = 0.25 * p.ns_tol2; }   // (read behind the barriers of the block set-up)

  // ---- one launch = `chain` consecutive blocks (1 when the blocks are launched one by one) ----
  // Chained, the blocks of a run pay the kernel launch, the cold instruction cache and the hand-off round trips once
  // instead of once per block; the state still travels from block to block through the f3_* dump (this CU's L1 / L2).
  const int nchain = b0.chain > 1 ? b0.chain : 1;
  for (int j = 0; j < nchain; ++j) {
  // (the thread index is made opaque per block: otherwise everything the programs' prologues derive from it -- lane masks,
  //  LDS addresses, layouts -- is loop-invariant, gets hoisted out of this loop and stays live across it: 3.3 KB of spills)
  int tid = tid0;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int role = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: the role branches are scalar branches
  const BlockParams& b = b0;
  F3Blk k;
  k.k0 = b0.k0; k.nb = b0.nb; k.last = b0.last; k.Acoef = b0.Acoef; k.Bcoef = b0.Bcoef; k.XG = b0.XG; k.Aprev = b0.Aprev;
  int assemble = b0.assemble;
  long long seq = b0.seq;
  if (b0.chain > 1) {
    const int slot = j & 1;
    k.k0 = b0.k0 + (long long)j * b0.chain_B;
    const long long left = b0.chain_kend - k.k0;
    k.nb = (int)(left < b0.chain_B ? left : b0.chain_B);
    k.Acoef = b0.Acoef0 + (size_t)slot * RB * RM;
    k.Bcoef = b0.Bcoef0 + (size_t)slot * RB * RB;
    seq = b0.seq + j;
    k.last = (j == nchain - 1) ? b0.last : 0;
    if (j > 0) {
      assemble = 1;
      k.XG = b0.XG0 + (size_t)slot * (RB + XGB) * XGB;
      k.Aprev = nullptr;                                               // left in sA by the block that just ended
    }
  }
  const long long t_begin = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz: in-situ duration / gap diagnostics
  if (j > 0) {
    if (!blk_chain_next(b0, seq)) return;
  } else {
    // Touch what the start-up will read -- the cross-Gram, the previous block's coefficients, the carried register dump --
    // while the hand-off flags are in flight: one memory round trip for the three instead of three in a row.  (A cross-Gram
    // that is not there yet is re-read after the poll's acquire fence.)
    double pf = 0.0;
    if (b.flags) {
      if (assemble) {
        pf = k.XG[(size_t)tid * 16];                                   // (RB + XGB) x XGB doubles = 512 lines of 128 bytes
        if (tid < RB * RM / 16) pf += k.Aprev[tid * 16];
      }
      constexpr int kDumpLines = (int)((sizeof(st->f3_G) + sizeof(st->f3_W) + sizeof(st->f3_Xc) + sizeof(st->f3_V) + sizeof(st->f3_Xa)) / 128);
      static_assert(kDumpLines <= F3_NT, "one line per thread");
      if (tid < kDumpLines) pf += st->f3_G[tid * 16];                  // the dump is contiguous from f3_G
    }
    if (!blk_handoff_begin(b)) return;
    if (pf == 1.2345e300) hot[0] = pf;                                 // (keeps the loads; never true)
  }
  const long long t_h = (long long)__builtin_amdgcn_s_memrealtime();
  if (!assemble) {
    for (int idx = tid; idx < RB * RB; idx += F3_NT) L.sK[idx] = b.K[idx];
  } else {
    f3_assemble_K(b, k, L, r, tid);
  }
  if (tid == 0) *L.errflag = 0;
  // filter4, a block that follows another one in the same launch: h, w, ee, N, kappa and (a, b) of that block's last step stay
  // where they are in LDS -- the first step's start predictor uses them
  const bool warm = KIND == 1 && j > 0;      // (filter5 has no start to predict)
  if (tid < RM) { L.mub[tid] = (tid < r && KIND == 0) ? st->mu[tid] : 0.0; if (!warm) { L.w[tid] = 0.0; L.h[tid] = 0.0; } }
  if (tid < F3_NSC && !warm) L.sc[tid] = 0.0;
  if (KIND >= 1) {
    // filter4 / filter5: mu_{k0}, theta and the block's share of the R_k / Q_k schedules into LDS (mu_bar, F of the first step: X pair's prologue)
    if (tid < RM) { D.mu[tid] = (tid < r) ? st->mu[tid] : 0.0; D.fd[tid] = 0.0; D.tp[tid] = 0.0; }
    if (tid >= 64 && tid < 64 + 2 * RM) {
      const int i = tid - 64, j = i & (RM - 1), hi = i >> 6;       // [0, RM): frequencies b (theta of cos-phase) | [RM, 2 RM): gains c
      const bool phased = p.dyn_kind == DYN_SINUSOID && (p.dyn_flags & 2);
      const bool have = p.n_theta > 0 && j < r && (hi == 0 || phased);
      const double tv = p.theta[have ? hi * r + j : 0];
      D.th[i] = have ? tv : 0.0;
    }
    if (tid >= 320 && tid < 320 + F4_NKC) f4_fill_trig_constants(D.kc, tid - 320);
    if (tid >= 256 && tid < 256 + 48) {
      const int jb_ = tid - 256;
      const long long ks = min((long long)(k.k0 + jb_ + 1), (long long)(k.k0 + k.nb)) - p.series_t0;
      if (p.rho_sched) D.rs[jb_] = p.rho_sched[ks];
      if (p.q_sched) D.qs[jb_] = p.q_sched[ks];
    }
  }
  const bool carried = st->ns_valid == (KIND == 2 ? 5 : (KIND == 1 ? 4 : 3));       // the previous block (or run) left the f3_* register dump behind
  __syncthreads();
  const long long t_a = (long long)__builtin_amdgcn_s_memrealtime();
  if (!carried && KIND == 0) {
    // Lbar_1 = (P + q I)^-1 by the direct sweep: both halves run it in lockstep on their own image
    const int lt = tid & (WG - 1), c32 = lt & 31, rg = lt >> 5;
    double* im = tid < WG ? L.img : L.img + 32 * F3_S;
    const double q0 = st->Q[0];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = rg + 8 * m;
      const bool in = (i < r) && (c32 < r);
      const double pv = st->P[in ? i * r + c32 : 0];
      im[i * F3_S + c32] = in ? pv + (i == c32 ? q0 : 0.0) : (i == c32 ? 1.0 : 0.0);
    }
    __syncthreads();
    f3_sweep_images(L, r2, tid);          // image X now holds Lbar_1 (ends with a barrier)
  }
  if (KIND == 2) {
    // filter5 (psmf_blk4.hip): the simplified hooks -- the vector program alone; waves 0-3 keep the barrier count
    if (role < 4) f5_idle_program(k);
    else if (role == 4) f5_v_program<4>(b, k, L, D, lane, carried);
    else if (role == 5) f5_v_program<5>(b, k, L, D, lane, carried);
    else if (role == 6) f5_v_program<6>(b, k, L, D, lane, carried);
    else f5_v_program<7>(b, k, L, D, lane, carried);
  } else if (KIND == 1) {
    // filter4 (psmf_blk4.hip): waves 0-1 the X pair (P+), 2-3 the Y pair (Lbar), 4-7 the vector waves
    const int md = SMALL ? 2 : (r == 32 ? 0 : 1);
    if (role < 2) {
      if (md == 2) { if (role & 1) f4_x_program<1, 2>(b, k, L, D, role, lane, carried, warm && carried); else f4_x_program<0, 2>(b, k, L, D, role, lane, carried, warm && carried); }
      else if (md == 0) { if (role & 1) f4_x_program<1, 0>(b, k, L, D, role, lane, carried, warm && carried); else f4_x_program<0, 0>(b, k, L, D, role, lane, carried, warm && carried); }
      else { if (role & 1) f4_x_program<1, 1>(b, k, L, D, role, lane, carried, warm && carried); else f4_x_program<0, 1>(b, k, L, D, role, lane, carried, warm && carried); }
    } else if (role < 4) {
      if (md == 2) { if (role & 1) f4_y_program<1, 2>(b, k, L, D, role, lane, carried, warm && carried); else f4_y_program<0, 2>(b, k, L, D, role, lane, carried, warm && carried); }
      else if (md == 0) { if (role & 1) f4_y_program<1, 0>(b, k, L, D, role, lane, carried, warm && carried); else f4_y_program<0, 0>(b, k, L, D, role, lane, carried, warm && carried); }
      else { if (role & 1) f4_y_program<1, 1>(b, k, L, D, role, lane, carried, warm && carried); else f4_y_program<0, 1>(b, k, L, D, role, lane, carried, warm && carried); }
    } else {
      if (role == 4) f4_v_program<4>(b, k, L, D, lane, carried, warm && carried);
      else if (role == 5) f4_v_program<5>(b, k, L, D, lane, carried, warm && carried);
      else if (role == 6) f4_v_program<6>(b, k, L, D, lane, carried, warm && carried);
      else f4_v_program<7>(b, k, L, D, lane, carried, warm && carried);
    }
  } else if (role < 4) {
    const int inv = role >> 1;
    if (SMALL) {
      if (role & 1) f3_ns_program<1, 2>(b, k, L, inv, role, lane, carried);
      else f3_ns_program<0, 2>(b, k, L, inv, role, lane, carried);
    } else if (r == 32) {
      if (role & 1) f3_ns_program<1, 0>(b, k, L, inv, role, lane, carried);
      else f3_ns_program<0, 0>(b, k, L, inv, role, lane, carried);
    } else if (r > 16) {       // (always true here; the test keeps the code placement of the build this kernel was tuned at: +-1.5 %)
      if (role & 1) f3_ns_program<1, 1>(b, k, L, inv, role, lane, carried);
      else f3_ns_program<0, 1>(b, k, L, inv, role, lane, carried);
    }
  } else {
    if (role == 4) f3_v_program<4>(b, k, L, lane, carried);
    else if (role == 5) f3_v_program<5>(b, k, L, lane, carried);
    else if (role == 6) f3_v_program<6>(b, k, L, lane, carried);
    else f3_v_program<7>(b, k, L, lane, carried);
  }
  if (tid == 0) {
    // cnt[4]: sum of in-kernel durations, cnt[5]: sum of the gaps to the previous filter kernel, cnt[7]: launches (10 ns ticks)
    const long long t_end = (long long)__builtin_amdgcn_s_memrealtime();
    st->dbg[0] += t_h - t_begin; st->dbg[1] += t_a - t_h; st->dbg[2] += L.tick[0] - t_a; st->dbg[3] += L.tick[1] - L.tick[0]; st->dbg[4] += t_end - L.tick[1];
    st->cnt[4] += t_end - t_begin;
    if (st->cnt[6] != 0) st->cnt[5] += t_begin - st->cnt[6];
    st->cnt[6] = t_end;
    st->cnt[7] += 1;
    if (j == 0) st->dbg[5] += 1;            // kernel launches
  }
  }   // chained blocks
  if (b0.chain > 1) {
    // the last block of the chain: complete, announced (the apply kernel of the bulk stream is waiting for it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid0 == 0) flag_store(b0.flags + 1, b0.seq + nchain);
  }
}

__global__ __launch_bounds__(F3_NT) void psmf_blk_filter3(BlockParams b0) { blk_filter3_body<false>(b0); }     // 16 < r <= 32
__global__ __launch_bounds__(F3_NT) void psmf_blk_filter3s(BlockParams b0) { blk_filter3_body<true>(b0); }     // r <= 16
// diagonal-Jacobian dynamics / per-step schedules (psmf_blk4.hip)
__global__ __launch_bounds__(F3_NT) void psmf_blk_filter4(BlockParams b0) { blk_filter3_body<false, 1>(b0); }   // 16 < r <= 32
__global__ __launch_bounds__(F3_NT) void psmf_blk_filter4s(BlockParams b0) { blk_filter3_body<true, 1>(b0); }   // r <= 16
// simplified hooks (ExperimentSynthetic), diagonal-Jacobian dynamics, any r <= 32 (psmf_blk4.hip)
__global__ __launch_bounds__(F3_NT) void psmf_blk_filter5(BlockParams b0) { blk_filter3_body<false, 2>(b0); }

}  // namespace psmf
