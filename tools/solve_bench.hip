// GPU-box microbenchmark (not part of the product): latency of the r x r solve block alone.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I rpsmf_amd/csrc tools/solve_bench.hip -o /tmp/sb && /tmp/sb
#include "../rpsmf_amd/csrc/psmf_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace psmf;

__global__ __launch_bounds__(WG) void solve_many(StepParams p, int nrep, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < nrep; ++i) {
    solve_block(p, sm);
    __syncthreads();
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
  for (int r : {8, 16, 20, 32, 64}) {
    DevState* st; hipMalloc((void**)&st, sizeof(DevState)); hipMemset(st, 0, sizeof(DevState));
    std::vector<double> P(r * r), G(r * r);
    // SPD test matrices
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) {
      P[i * r + j] = (i == j ? 1.0 : 0.0) + 0.3 / (1.0 + std::abs(i - j));
      G[i * r + j] = (i == j ? 50.0 : 0.0) + 10.0 * std::cos(0.1 * (i - j));
    }
    hipMemcpy(st->Pbar, P.data(), r * r * 8, hipMemcpyHostToDevice);
    hipMemcpy(st->G, G.data(), r * r * 8, hipMemcpyHostToDevice);
    double kappa = 0.5; hipMemcpy(&st->kappa, &kappa, 8, hipMemcpyHostToDevice);
    StepParams p{}; p.st = st; p.r = r; p.coef_update = 1;
    unsigned long long* clk; hipMalloc((void**)&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nrep = 200; const size_t lds = 64 * 1024;
    solve_many<<<1, WG, lds>>>(p, 10, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    solve_many<<<1, WG, lds>>>(p, nrep, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    // check: Pplus * (Pbar^-1 + kappa G) = I  <=>  (I + kappa Pbar G) Pplus = Pbar
    std::vector<double> X(r * r); hipMemcpy(X.data(), st->Pplus, r * r * 8, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) {
      double acc = X[i * r + j];
      for (int l = 0; l < r; ++l) { double pg = 0; for (int m = 0; m < r; ++m) pg += P[i * r + m] * G[m * r + l]; acc += kappa * pg * X[l * r + j]; }
      maxerr = std::fmax(maxerr, std::fabs(acc - P[i * r + j])); maxref = std::fmax(maxref, std::fabs(P[i * r + j]));
    }
    int err; hipMemcpy(&err, &st->err, 4, hipMemcpyDeviceToHost);
    printf("r=%d  solve %.2f us  (%.0f shader cycles, shader clock %.2f GHz)  residual %.2e err=%d\n", r, ms * 1e3 / nrep,
           (double)h[0] / nrep, (double)h[0] / ((double)h[1] * 10.0) , maxerr / maxref, err);
    hipFree(st); hipFree(clk);
  }
  return 0;
}
