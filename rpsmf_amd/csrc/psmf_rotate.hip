// A non-diagonal observation covariance R (psmf.py:150-152, rpsmf.py:150-152: the reference's dense d x d branch of
// _compute_inverse_coefficient_innovation).  With R = U diag(lam) U^T every update of the recursion is equivariant under
// y -> U^T y, C -> U^T C (eta = tr(R + C Pbar C^T) / d and the residual norms are invariant), so the handle runs the
// non-uniform-diagonal step (rho_rows = lam, psmf_set_row_noise's path) on the ROTATED series and dictionary and rotates at its
// boundary: the series and C on the way in, C, y_hat and the roll-out on the way out.  No d x d inverse is formed; the cost per
// step stays O(d r^2), the rotations are four kinds of plain float64 GEMMs against the resident U (psmf_set_noise_rotation).
//
// psmf_rot_gemm: O[M x N] = A[M x K] B[K x N], any strides on A and B (so that U and U^T, row-major C with padded rows and the
// time-major series all go through one kernel), float64 accumulation on the matrix cores (v_mfma_f64_16x16x4), 64 x 64 tile per
// workgroup, K in slabs of 16 through LDS.  One-off work per upload / download, not the per-step path: written for correctness
// and a fair fraction of the float64 MFMA rate, not tuned further.
#pragma once
#include "psmf_blk3.hip"      // f64x4

namespace psmf {

constexpr int ROT_T = 64;     // output tile
constexpr int ROT_K = 16;     // K slab

template <typename TA, typename TB, typename TO>
__global__ __launch_bounds__(256) void psmf_rot_gemm(const TA* __restrict__ A, const long long a_i, const long long a_k,
                                                      const TB* __restrict__ B, const long long b_k, const long long b_j,
                                                      TO* __restrict__ O, const long long o_i, const int M, const int N, const int K) {
  __shared__ double As[ROT_K][ROT_T + 1];      // As[k][i]
  __shared__ double Bs[ROT_K][ROT_T + 1];      // Bs[k][j]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int i0 = blockIdx.y * ROT_T, j0 = blockIdx.x * ROT_T;
  f64x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  const bool a_kfast = a_k == 1, b_jfast = b_j == 1;
  for (int k0 = 0; k0 < K; k0 += ROT_K) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;
      {
        const int i = a_kfast ? idx >> 4 : idx & 63, k = a_kfast ? idx & 15 : idx >> 6;
        const long long gi = i0 + i, gk = k0 + k;
        As[k][i] = (gi < M && gk < K) ? (double)A[gi * a_i + gk * a_k] : 0.0;
      }
      {
        const int j = b_jfast ? idx & 63 : idx >> 4, k = b_jfast ? idx >> 6 : idx & 15;
        const long long gj = j0 + j, gk = k0 + k;
        Bs[k][j] = (gj < N && gk < K) ? (double)B[gk * b_k + gj * b_j] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < ROT_K; kk += 4) {
      const double a = As[kk + lk][16 * w + lr];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[kk + lk][16 * t + lr], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long i = i0 + 16 * w + lk + 4 * q, j = j0 + 16 * t + lr;
      if (i < M && j < N) O[i * o_i + j] = (TO)acc[t][q];
    }
}

}  // namespace psmf
