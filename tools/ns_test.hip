// GPU-box check of the f64-MFMA Newton-Schulz tiles against a CPU inverse (layout + convergence).
#include "../rpsmf_amd/csrc/psmf_kernels.hip"
#include "../rpsmf_amd/csrc/psmf_ns.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace psmf;
__global__ __launch_bounds__(256) void ns_kernel(const double* M, const double* X0, double* Xout, double* norms, int n, int iters) {
  __shared__ double sM[NS_N * NS_S], sX[NS_N * NS_S], sR[NS_N * NS_S];
  __shared__ double snrm[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ti = w >> 1, tj = w & 1;
  for (int idx = tid; idx < NS_N * NS_N; idx += 256) { const int i = idx / NS_N, j = idx % NS_N; sM[i * NS_S + j] = M[idx]; sX[i * NS_S + j] = X0[idx]; }
  __syncthreads();
  const bool active = n == 32 || w == 0;
  unsigned long long t0 = clock64();
  for (int it = 0; it <= iters; ++it) {
    double nr = 0.0;
    if (active) nr = n == 32 ? ns_residual<32>(sM, sX, sR, ti, tj, lane) : ns_residual<16>(sM, sX, sR, ti, tj, lane);
    nr = (double)wave_sum_f32_dpp((float)nr);
    if (lane == 0) snrm[w] = nr;
    __syncthreads();
    if (tid == 0) norms[it] = sqrt((snrm[0] + snrm[1]) + (snrm[2] + snrm[3]));
    if (it == iters) break;
    f64x4 acc = {0, 0, 0, 0};
    if (active) acc = n == 32 ? ns_update_tile<32>(sX, sR, ti, tj, lane) : ns_update_tile<16>(sX, sR, ti, tj, lane);
    __syncthreads();
    if (active) ns_store_tile(sX, acc, ti, tj, lane);
    __syncthreads();
  }
  unsigned long long t1 = clock64();
  if (tid == 0) norms[15] = (double)(t1 - t0);
  for (int idx = tid; idx < NS_N * NS_N; idx += 256) { const int i = idx / NS_N, j = idx % NS_N; Xout[idx] = sX[i * NS_S + j]; }
}
static void cpu_inv(std::vector<double> A, std::vector<double>& Ai, int n) {
  Ai.assign(n * n, 0.0); for (int i = 0; i < n; ++i) Ai[i * n + i] = 1.0;
  for (int k = 0; k < n; ++k) { double p = A[k * n + k]; for (int j = 0; j < n; ++j) { A[k * n + j] /= p; Ai[k * n + j] /= p; }
    for (int i = 0; i < n; ++i) if (i != k) { double f = A[i * n + k]; for (int j = 0; j < n; ++j) { A[i * n + j] -= f * A[k * n + j]; Ai[i * n + j] -= f * Ai[k * n + j]; } } }
}
int main() {
  const int N = NS_N;
  for (int n : {32, 16}) for (double pert : {0.02, 0.1, 0.3}) {
    std::vector<double> M(N * N, 0.0), Mp(N * N, 0.0), X0, Xref;
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
      const bool in = i < n && j < n;
      M[i * N + j] = in ? ((i == j ? 40.0 + i : 0.0) + 10.0 * std::cos(0.3 * (i - j)) * std::exp(-0.05 * std::abs(i - j))) : (i == j ? 1.0 : 0.0);
      Mp[i * N + j] = in ? M[i * N + j] * (1.0 + pert * std::sin(1.0 + i + j)) : M[i * N + j];   // symmetric perturbation
    }
    cpu_inv(Mp, X0, N); cpu_inv(M, Xref, N);       // X0 = inverse of a perturbed M (a "previous step")
    double *dM, *dX0, *dX, *dn; hipMalloc((void**)&dM, N * N * 8); hipMalloc((void**)&dX0, N * N * 8); hipMalloc((void**)&dX, N * N * 8); hipMalloc((void**)&dn, 16 * 8);
    hipMemcpy(dM, M.data(), N * N * 8, hipMemcpyHostToDevice); hipMemcpy(dX0, X0.data(), N * N * 8, hipMemcpyHostToDevice);
    ns_kernel<<<1, 256>>>(dM, dX0, dX, dn, n, 4); hipDeviceSynchronize();
    std::vector<double> X(N * N); double nr[16]; hipMemcpy(X.data(), dX, N * N * 8, hipMemcpyDeviceToHost); hipMemcpy(nr, dn, 128, hipMemcpyDeviceToHost);
    double err = 0, ref = 0; for (int i = 0; i < N * N; ++i) { err = std::fmax(err, std::fabs(X[i] - Xref[i])); ref = std::fmax(ref, std::fabs(Xref[i])); }
    printf("n=%d pert=%.2f  ||R||: %.2e %.2e %.2e %.2e %.2e   max|X - inv(M)|/max|inv| = %.2e   %.0f cycles for 4 iterations (9 products)\n", n, pert, nr[0], nr[1], nr[2], nr[3], nr[4], err / ref, nr[15]);
  }
  return 0;
}
