// Newton-Schulz refinement of an r x r inverse on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// The filter inverts a slowly changing SPD matrix every step; the previous step's inverse X0 is an
// excellent starting point.  With R = I - M X, the iteration X <- X + X R squares the residual
// (R <- R^2), is made of matrix products only (no pivot chain: the symmetric sweep of psmf_kernels.hip
// pays ~1000 cycles of LDS/barrier latency per 2 x 2 pivot, 16 times for r = 32), and converges to
// float64 round-off in 2-4 iterations when ||R0||_F < ~0.2.  The caller falls back to the sweep when
// the start is too far (first step, regime changes) or the final residual is not at round-off.
//
// One 256-thread half-workgroup (4 waves) per matrix; wave w owns output tile (w >> 1, w & 1) of the
// 32 x 32 product (only tile (0,0) when the padded size is 16).  Matrices live in LDS with row stride
// NS_S = 34 doubles (conflict-free A-operand reads, 2-way B-operand reads).  No barrier inside the
// tile routines; the driver's barriers are shared by both halves of a 512-thread workgroup.
#pragma once
#include "psmf_device.h"

namespace psmf {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int NS_S = 34;           // LDS row stride (doubles)
constexpr int NS_N = 32;           // padded matrix size

// sum over the 64 lanes of a wave in float32 with DPP row operations (~12 VALU instructions, no LDS
// round trips; __shfl_xor on a double costs two ds_bpermute per level).  Used for norms that only
// steer the iteration, never for filter arithmetic.  Result valid in every lane (readlane broadcast).
__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
  auto dpp = [](float x, const int ctrl_sel) -> float {
    int r;
    switch (ctrl_sel) {
      case 0: r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
      case 1: r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
      case 2: r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true); break;  // row_half_mirror
      default: r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return __int_as_float(r);
  };
  v += dpp(v, 0);
  v += dpp(v, 1);
  v += dpp(v, 2);
  v += dpp(v, 3);   // every lane of a 16-lane row holds its row's sum
  const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (s0 + s1) + (s2 + s3);
}

// D tile = C tile + A[16 ti .. , :] * B[:, 16 tj ..]   over k = 0 .. N (N = 16 or 32, compile time:
// every LDS operand read is issued before the first MFMA -- one LDS round trip, then the dependent
// accumulator chain).  lane l supplies A[16 ti + (l & 15)][k0 + (l >> 4)], B[k0 + (l >> 4)][16 tj + (l & 15)];
// result register q holds row (l >> 4) + 4 q, column l & 15 of the tile.
template <int N>
__device__ __forceinline__ f64x4 ns_tile_mm(const double* __restrict__ A, const double* __restrict__ B, f64x4 acc,
                                            const int ti, const int tj, const int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const double* ap = A + (16 * ti + lr) * NS_S + lk;
  const double* bp = B + lk * NS_S + 16 * tj + lr;
  double a[N / 4], b[N / 4];
#pragma unroll
  for (int q = 0; q < N / 4; ++q) {
    a[q] = ap[4 * q];
    b[q] = bp[4 * q * NS_S];
  }
#pragma unroll
  for (int q = 0; q < N / 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
  return acc;
}

// R = I - M X  -> sR ; returns this LANE's share of ||R||_F^2 (the caller reduces it)
template <int N>
__device__ __forceinline__ double ns_residual(const double* sM, const double* sX, double* sR, const int ti, const int tj,
                                              const int lane) {
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  acc = ns_tile_mm<N>(sM, sX, acc, ti, tj, lane);
  const int col = 16 * tj + (lane & 15);
  double nrm = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 16 * ti + (lane >> 4) + 4 * q;
    const double v = (row == col ? 1.0 : 0.0) - acc[q];
    sR[row * NS_S + col] = v;
    nrm += v * v;
  }
  return nrm;
}

// X <- X + X R  (tile (ti, tj) of X, in place in LDS: the products read the OLD X, so the caller
// keeps a barrier between this and anything that reads the new X)
template <int N>
__device__ __forceinline__ f64x4 ns_update_tile(const double* sX, const double* sR, const int ti, const int tj,
                                                const int lane) {
  const int col = 16 * tj + (lane & 15);
  f64x4 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = sX[(16 * ti + (lane >> 4) + 4 * q) * NS_S + col];
  return ns_tile_mm<N>(sX, sR, acc, ti, tj, lane);
}

__device__ __forceinline__ void ns_store_tile(double* sX, const f64x4 acc, const int ti, const int tj, const int lane) {
  const int col = 16 * tj + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; ++q) sX[(16 * ti + (lane >> 4) + 4 * q) * NS_S + col] = acc[q];
}

__device__ __forceinline__ double readlane_f64(double v, const int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// ------------------------------------------------------------------------------------------------------------
// Symmetric sweep of a matrix up to 32 x 32 held by ONE wave as NT x NT tiles of 16 x 16 in the MFMA output layout (the 16 x 16
// original: wave_sweep16 / wave_sweep16m, psmf_impute.hip / psmf_wave16.hip)
// (tile (ti, tj), lane l: column 16 tj + (l & 15), rows 16 ti + (l >> 4) + 4 q):  A <- -A^-1 of the leading r2 x r2 block (r2 even;
// identity padding beyond r), 2 x 2 SPD block pivots, no LDS, no barrier.  A round pivots on rows k, k + 1 of tile row tp = k >> 4:
// they sit in register kq = (k & 15) >> 2 of the lanes lk = k & 3, (k + 1) & 3 of the tiles (tp, 0 .. NT) -- which is where the A
// operand of the update of tile ROW ti wants the pivot columns' entries 16 ti .. 16 ti + 15 (by symmetry the pivot rows' entries in
// tile (tp, ti)); the B operand of tile COLUMN tj is -Ki [u; w]^T over the columns of tile tj (+Ki at the pivot columns, whose C
// input is zeroed), formed from u_j, w_j that one v_permlane16_swap pair brings into both rows.  NT^2 independent MFMAs per round
// (pipelined), then the pivot rows are overwritten in place.  Used where the LDS-and-barrier sweep of psmf_kernels.hip (one
// publish / barrier / read exchange per round, ~700 cycles alone and ~1400 on a CU shared with a streaming workgroup) sat on a
// step's critical path: the solve block of the per-step engine (r <= 32).
// ------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void wave_sweep_tiles(double (&A)[NT][NT][4], const int r2, const int lk, const int lr, bool& bad) {
#pragma unroll
  for (int k = 0; k < 16 * NT; k += 2) {
    if (k < r2) {                                  // uniform
      const int tp = k >> 4, kl = k & 15, kq = kl >> 2, b0 = (kl & 3) << 4, b1 = b0 + 16;
      const double rkd = A[tp][tp][kq];
      const double ka = readlane_f64(rkd, b0 | kl), kb = readlane_f64(rkd, b0 | (kl + 1)), ke = readlane_f64(rkd, b1 | (kl + 1));
      const double det = ka * ke - kb * kb;
      bad |= !(ka > 0.0) | !(det > 0.0);
      const double x0 = __builtin_amdgcn_rcp(det);
      const double e1 = fma(-det, x0, 1.0);
      const double dinv = fma(x0 * e1, 1.0 + e1, x0);       // 1 / det: v_rcp_f64 and one cubic step
      const double kp = ke * dinv, kq2 = -kb * dinv, ks = ka * dinv;       // Ki = [[kp, kq2], [kq2, ks]]
      const bool in_piv = (lk >> 1) == ((kl >> 1) & 1);     // the lanes that hold the two pivot rows
      const bool is_u = (lk & 1) == 0;                      // ... row k (else row k + 1)
      const double cu = is_u ? kp : kq2, cw = is_u ? kq2 : ks;
      double aop[NT], bop[NT], sv[NT];
      bool piv[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const double rk = A[tp][tj][kq];
        const unsigned lo = __double2loint(rk), hi = __double2hiint(rk);
        const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const double uj = __hiloint2double(h2[0], l2[0]), wj = __hiloint2double(h2[1], l2[1]);
        const bool c0 = (tj == tp) && (lr == kl), c1 = (tj == tp) && (lr == kl + 1);
        piv[tj] = c0 | c1;
        const double u1 = c0 ? 1.0 : (c1 ? 0.0 : uj), w1 = c0 ? 0.0 : (c1 ? 1.0 : wj);
        sv[tj] = cu * u1 + cw * w1;                         // t1_j / t2_j; at the pivot columns the entries of Ki
        aop[tj] = in_piv ? rk : 0.0;
        bop[tj] = in_piv ? (piv[tj] ? sv[tj] : -sv[tj]) : 0.0;
      }
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          f64x4 acc = {A[ti][tj][0], A[ti][tj][1], A[ti][tj][2], A[ti][tj][3]};
          if (tj == tp) {                  // (compile time) the pivot columns lie in tile column tp: their C input is zeroed
            const double keep = piv[tj] ? 0.0 : 1.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] *= keep;
          }
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tj], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) A[ti][tj][q] = acc[q];
        }
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) A[tp][tj][kq] = in_piv ? (piv[tj] ? -sv[tj] : sv[tj]) : A[tp][tj][kq];   // pivot rows: t1, t2; pivot block: -Ki
    }
  }
}

// Per-lane constants of wave_sweep16m as numbers (lane: column lr = l & 15, rows lk + 4 q, lk = l >> 4); round j pivots on
// rows / columns 2 j, 2 j + 1, which are register A[j >> 1] of the lane rows lk = 2 (j & 1), 2 (j & 1) + 1.
struct Sw16K {
  double pc0[8], pc1[8];   // lr == 2 j, lr == 2 j + 1
  double fnp[8];           // 1 - pc0 - pc1: not a pivot column
  double sg[8];            // lanes of the pivot rows: +1 at the pivot columns, -1 elsewhere; other lanes 0
  double fpiv[2], fnpiv[2];   // lanes that hold the pivot rows (lk >> 1 == j & 1), and 1 - that
  double fu, fw;           // lk even (row 2 j of the pair) / odd (row 2 j + 1)
};
__device__ __forceinline__ void sw16k_init(Sw16K& c, const int lk, const int lr) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    c.pc0[j] = lr == 2 * j ? 1.0 : 0.0;
    c.pc1[j] = lr == 2 * j + 1 ? 1.0 : 0.0;
    c.fnp[j] = 1.0 - c.pc0[j] - c.pc1[j];
    const double piv = (lk >> 1) == (j & 1) ? 1.0 : 0.0;
    c.sg[j] = piv * (2.0 * (c.pc0[j] + c.pc1[j]) - 1.0);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) { c.fpiv[h] = (lk >> 1) == h ? 1.0 : 0.0; c.fnpiv[h] = 1.0 - c.fpiv[h]; }
  c.fu = (lk & 1) == 0 ? 1.0 : 0.0;
  c.fw = 1.0 - c.fu;
}

// wave_sweep_tiles with the lane predicates as multipliers (the form of wave_sweep16m, psmf_wave16.hip): a select of a double is two
// v_cndmask and -- in kernels that have run out of SGPRs -- the reload of its lane mask from a spilled pair; a multiply by 0.0 / 1.0 is one
// instruction.  The select form spent 24 v_cndmask and 8 mask reloads of the 89 instructions of a round (NT = 2, ISA of
// psmf_blk_filter7); c: sw16k_init(lk, lr), tile-local (the pivot columns of a round lie in tile column tp only).
template <int NT>
__device__ __forceinline__ void wave_sweep_tiles_m(double (&A)[NT][NT][4], const int r2, const Sw16K& c, bool& bad) {
#pragma unroll
  for (int k = 0; k < 16 * NT; k += 2) {
    if (k < r2) {                                  // uniform
      const int tp = k >> 4, kl = k & 15, j = kl >> 1, h = j & 1, kq = j >> 1, b0 = h << 5, b1 = b0 + 16;
      const double rkd = A[tp][tp][kq];
      const double ka = readlane_f64(rkd, b0 | kl), kb = readlane_f64(rkd, b0 | (kl + 1)), ke = readlane_f64(rkd, b1 | (kl + 1));
      const double det = ka * ke - kb * kb;
      bad |= !(ka > 0.0) | !(det > 0.0);
      const double x0 = __builtin_amdgcn_rcp(det);
      const double e1 = fma(-det, x0, 1.0);
      const double dinv = fma(x0 * e1, 1.0 + e1, x0);
      // det * (row of Ki that belongs to this lane's pivot row): even row [ke, -kb], odd row [-kb, ka]
      const double cu = c.fu * ke - c.fw * kb, cw = c.fw * ka - c.fu * kb;
      double aop[NT], bop[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const double rk = A[tp][tj][kq];
        const unsigned lo = __double2loint(rk), hi = __double2hiint(rk);
        const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const double uj = __hiloint2double(h2[0], l2[0]), wj = __hiloint2double(h2[1], l2[1]);
        double pre;
        if (tj == tp) {            // (compile time) pivot columns: unit vectors -> the entries of Ki, with the opposite sign
          const double u1 = fma(uj, c.fnp[j], c.pc0[j]), w1 = fma(wj, c.fnp[j], c.pc1[j]);
          pre = fma(cu, u1, cw * w1) * c.sg[j];
        } else {
          pre = -fma(cu, uj, cw * wj) * c.fpiv[h];
        }
        aop[tj] = rk * c.fpiv[h];
        bop[tj] = pre * dinv;
      }
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          f64x4 acc = {A[ti][tj][0], A[ti][tj][1], A[ti][tj][2], A[ti][tj][3]};
          if (tj == tp) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] *= c.fnp[j];
          }
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ti], bop[tj], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) A[ti][tj][q] = acc[q];
        }
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) A[tp][tj][kq] = fma(A[tp][tj][kq], c.fnpiv[h], -bop[tj]);      // the pivot rows: t; pivot block: -Ki
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// A <- -A^-1 for 33 <= r2 <= 48 (3 x 3 tiles) by BLOCKS instead of 24 rounds over all nine tiles (round 5): with A = [[A11, B], [B^T, D]],
// A11 the leading 32 x 32,
//     A11^-1 (sweep on 2 x 2 tiles: 16 rounds of 4 MFMAs),  E = A11^-1 B,  S = D - B^T E,  S^-1 (sweep on one tile: (r2 - 32) / 2 rounds),
//     A^-1 = [[A11^-1 + F E^T, -F], [-F^T, S^-1]],  F = E S^-1.
// Every product is  C = X^T Y  over tiles in the MFMA output layout ("T-layout"), which is the B-operand layout of Y and the
// A-operand layout of X^T (psmf_blk3.hip) -- no operand is shuffled:  E = A11^-1^T B (A11^-1 symmetric),  E^T = B^T A11^-1,
// S = D - B^T E,  F^T = S^-1^T E^T,  F = (E^T)^T S^-1,  F E^T = (F^T)^T E^T.  72 MFMAs of products beside the two short sweeps: a
// round of the 3 x 3 sweep costs ~1 700 cycles of one wave's instruction stream, of the 2 x 2 one ~850, of the single tile ~500.
// Identity padding beyond r2 (as wave_sweep_tiles_m expects it) stays padding: the padded rows / columns of B are zero.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_xty(const double (&X)[4], const double (&Y)[4], f64x4& acc) {      // acc += X^T Y  (16 x 16 tiles, K = 16)
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[q], Y[q], acc, 0, 0, 0);
}

__device__ __forceinline__ void wave_inverse_blocked2(double (&A)[2][2][4], const int r2, const Sw16K& c, bool& bad);
__device__ __forceinline__ void wave_inverse_blocked3(double (&A)[3][3][4], const int r2, const Sw16K& c, bool& bad) {
  double T[2][2][4];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) T[ti][tj][q] = A[ti][tj][q];
  wave_inverse_blocked2(T, 32, c, bad);                       // T = -A11^-1
  // (every group of products below runs its independent accumulators interleaved, k-step by k-step: an MFMA that accumulates into the
  //  result of the one before it waits for it)
#define MF(acc_, x_, y_) acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(x_, y_, acc_, 0, 0, 0)
  // E = A11^-1 B = -(T^T B)  (two row tiles);  E^T = -(B^T T)  (two column tiles)
  double E[2][4], Et[2][4];
  {
    f64x4 e0 = {0.0, 0.0, 0.0, 0.0}, e1 = e0, t0 = e0, t1 = e0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        MF(e0, T[m][0][q], A[m][2][q]); MF(e1, T[m][1][q], A[m][2][q]);
        MF(t0, A[m][2][q], T[m][0][q]); MF(t1, A[m][2][q], T[m][1][q]);
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) { E[0][q] = -e0[q]; E[1][q] = -e1[q]; Et[0][q] = -t0[q]; Et[1][q] = -t1[q]; }
  }
  // S = D - B^T E
  double S1[1][1][4];
  {
    f64x4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { MF(s0, A[0][2][q], E[0][q]); MF(s1, A[1][2][q], E[1][q]); }
#pragma unroll
    for (int q = 0; q < 4; ++q) S1[0][0][q] = A[2][2][q] - (s0[q] + s1[q]);
  }
  wave_sweep_tiles_m<1>(S1, r2 - 32, c, bad);                 // S1 = -S^-1 on its leading (r2 - 32) block; the padding keeps its ones
  // F^T = S^-1 E^T = -(S1^T E^T),  F = E S^-1 = -((E^T)^T S1)
  double Ft[2][4], F[2][4];
  {
    f64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      MF(a0, S1[0][0][q], Et[0][q]); MF(a1, S1[0][0][q], Et[1][q]);
      MF(b0, Et[0][q], S1[0][0][q]); MF(b1, Et[1][q], S1[0][0][q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { Ft[0][q] = -a0[q]; Ft[1][q] = -a1[q]; F[0][q] = -b0[q]; F[1][q] = -b1[q]; }
  }
  // -A^-1: top left -(A11^-1 + F E^T) = T - (F^T)^T E^T, top right +F, bottom left +F^T, bottom right -S^-1 = S1
  {
    f64x4 c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      MF(c00, Ft[0][q], Et[0][q]); MF(c01, Ft[0][q], Et[1][q]);
      MF(c10, Ft[1][q], Et[0][q]); MF(c11, Ft[1][q], Et[1][q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      A[0][0][q] = T[0][0][q] - c00[q]; A[0][1][q] = T[0][1][q] - c01[q];
      A[1][0][q] = T[1][0][q] - c10[q]; A[1][1][q] = T[1][1][q] - c11[q];
    }
  }
#undef MF
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) { A[t][2][q] = F[t][q]; A[2][t][q] = Ft[t][q]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) A[2][2][q] = S1[0][0][q];
}

// The same by blocks for 17 <= r2 <= 32 (2 x 2 tiles): two single-tile sweeps (8 + (r2 - 16) / 2 rounds of ~500 cycles) and 24 MFMAs of
// products against 16 rounds of ~850 on four tiles.  Each 16 x 16 x 16 product is split over two accumulators (k-steps 0, 1 | 2, 3): half
// the chain of dependent MFMAs.
__device__ __forceinline__ void wave_inverse_blocked2(double (&A)[2][2][4], const int r2, const Sw16K& c, bool& bad) {
#define MF(acc_, x_, y_) acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(x_, y_, acc_, 0, 0, 0)
  double T[1][1][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) T[0][0][q] = A[0][0][q];
  wave_sweep_tiles_m<1>(T, 16, c, bad);                       // T = -A11^-1
  double E[4], Et[4];
  {
    f64x4 e0 = {0.0, 0.0, 0.0, 0.0}, e1 = e0, t0 = e0, t1 = e0;
    MF(e0, T[0][0][0], A[0][1][0]); MF(e1, T[0][0][2], A[0][1][2]); MF(t0, A[0][1][0], T[0][0][0]); MF(t1, A[0][1][2], T[0][0][2]);
    MF(e0, T[0][0][1], A[0][1][1]); MF(e1, T[0][0][3], A[0][1][3]); MF(t0, A[0][1][1], T[0][0][1]); MF(t1, A[0][1][3], T[0][0][3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) { E[q] = -(e0[q] + e1[q]); Et[q] = -(t0[q] + t1[q]); }
  }
  double S1[1][1][4];
  {
    f64x4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0;
    MF(s0, A[0][1][0], E[0]); MF(s1, A[0][1][2], E[2]);
    MF(s0, A[0][1][1], E[1]); MF(s1, A[0][1][3], E[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) S1[0][0][q] = A[1][1][q] - (s0[q] + s1[q]);
  }
  wave_sweep_tiles_m<1>(S1, r2 - 16, c, bad);                 // S1 = -S^-1 (padding keeps its ones)
  double Ft[4], F[4];
  {
    f64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, b0 = a0, b1 = a0;
    MF(a0, S1[0][0][0], Et[0]); MF(a1, S1[0][0][2], Et[2]); MF(b0, Et[0], S1[0][0][0]); MF(b1, Et[2], S1[0][0][2]);
    MF(a0, S1[0][0][1], Et[1]); MF(a1, S1[0][0][3], Et[3]); MF(b0, Et[1], S1[0][0][1]); MF(b1, Et[3], S1[0][0][3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) { Ft[q] = -(a0[q] + a1[q]); F[q] = -(b0[q] + b1[q]); }
  }
  {
    f64x4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0;
    MF(c0, Ft[0], Et[0]); MF(c1, Ft[2], Et[2]);
    MF(c0, Ft[1], Et[1]); MF(c1, Ft[3], Et[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) A[0][0][q] = T[0][0][q] - (c0[q] + c1[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { A[0][1][q] = F[q]; A[1][0][q] = Ft[q]; A[1][1][q] = S1[0][0][q]; }
#undef MF
}

// the sweep a solve uses for NTL x NTL tiles: by blocks where that is the shorter instruction stream (NTL = 2, 3)
template <int NTL>
__device__ __forceinline__ void wave_invert_tiles(double (&A)[NTL][NTL][4], const int r2, const Sw16K& c, bool& bad) {
  if constexpr (NTL == 3) {
    if (r2 > 32) { wave_inverse_blocked3(A, r2, c, bad); return; }
  }
  if constexpr (NTL == 2) {
    if (r2 > 16) { wave_inverse_blocked2(A, r2, c, bad); return; }
  }
  wave_sweep_tiles_m<NTL>(A, r2, c, bad);
}

}  // namespace psmf
