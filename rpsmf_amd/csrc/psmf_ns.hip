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

}  // namespace psmf
