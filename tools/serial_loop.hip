// GPU-box diagnostic: the serial stage executed N times inside ONE launch (warm caches) vs N launches.
#include "../rpsmf_amd/csrc/psmf_kernels.hip"
#include <cstdio>
#include <vector>
using namespace psmf;
__global__ __launch_bounds__(512) void serial_many(StepParams p, int n) {
  for (int i = 0; i < n; ++i) { serial_body<32>(p, 0); __syncthreads(); }
}
int main() {
  const int r = 32, nwg = 447, ps = r + 1;
  DevState* st; hipMalloc((void**)&st, sizeof(DevState)); hipMemset(st, 0, sizeof(DevState));
  std::vector<double> I(r * r, 0.0); for (int i = 0; i < r; ++i) I[i * r + i] = 1.0;
  for (double* dst : {st->V, st->P, st->Pplus, st->Pbar, st->Q, st->G}) hipMemcpy(dst, I.data(), r * r * 8, hipMemcpyHostToDevice);
  double one = 1.0; hipMemcpy(&st->rho, &one, 8, hipMemcpyHostToDevice); hipMemcpy(&st->N, &one, 8, hipMemcpyHostToDevice); hipMemcpy(&st->kappa, &one, 8, hipMemcpyHostToDevice);
  double* part; hipMalloc((void**)&part, (size_t)nwg * ps * 8 + 65536); hipMemset(part, 0, (size_t)nwg * ps * 8 + 65536);
  StepParams p{}; p.st = st; p.partials = part; p.r = r; p.d = 100000; p.d_local = 100000; p.n_sweep_wg = nwg; p.ps = ps;
  p.coef_update = 1; p.eta_full = 1; p.pbar_predict = 1; p.track_g = 1; p.alpha = p.beta = 1.0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  const int n = 200;
  // workers only (256 threads): the helper waves of the product kernel retire after the reduction
  serial_many<<<1, 256>>>(p, 5); hipDeviceSynchronize();
  hipEventRecord(e0); serial_many<<<1, 256>>>(p, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("in one launch (256 threads):  %.2f us per serial stage\n", ms * 1e3 / n);
  for (int i = 0; i < 5; ++i) psmf_serial<32><<<1, 512>>>(p, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < n; ++i) psmf_serial<32><<<1, 512>>>(p, 0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("one launch each (512 threads): %.2f us per serial stage\n", ms * 1e3 / n);
  hipEventRecord(e0); for (int i = 0; i < n; ++i) psmf_serial<32><<<1, 512>>>(p, 1); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("one launch each, first=1:      %.2f us per serial stage\n", ms * 1e3 / n);
  return 0;
}
