// GPU-box diagnostic: phase timing of psmf_serial<RPAD> (stamps compiled in with PSMF_SERIAL_STAMPS).  -DPROF_R=40 -DPROF_NWG=248: the r > 32 instance.
#define PSMF_SERIAL_STAMPS 1
#include "../rpsmf_amd/csrc/psmf_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace psmf;
#ifndef PROF_R
#define PROF_R 32
#endif
#ifndef PROF_NWG
#define PROF_NWG 224
#endif
int main() {
  constexpr int r = PROF_R, RPAD = PROF_R > 32 ? 64 : 32;
  const int nwg = PROF_NWG, ps = r + 1;      // 224 sweep workgroups at d = 1e5 (512-thread sweep); r > 32: 248
  DevState* st; hipMalloc((void**)&st, sizeof(DevState)); hipMemset(st, 0, sizeof(DevState));
  std::vector<double> I(r * r, 0.0); for (int i = 0; i < r; ++i) I[i * r + i] = 1.0;
  hipMemcpy(st->V, I.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(st->P, I.data(), r * r * 8, hipMemcpyHostToDevice);
  hipMemcpy(st->Pplus, I.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(st->Pbar, I.data(), r * r * 8, hipMemcpyHostToDevice);
  hipMemcpy(st->Q, I.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(st->G, I.data(), r * r * 8, hipMemcpyHostToDevice);
  double one = 1.0; hipMemcpy(&st->rho, &one, 8, hipMemcpyHostToDevice); hipMemcpy(&st->N, &one, 8, hipMemcpyHostToDevice); hipMemcpy(&st->kappa, &one, 8, hipMemcpyHostToDevice);
  double* part; hipMalloc((void**)&part, (4096 + 64) * 8 + (size_t)nwg * ps * 8 + 65536); hipMemset(part, 0, (4096 + 64) * 8 + (size_t)nwg * ps * 8 + 65536);
  StepParams p{}; p.st = st; p.partials = part; p.r = r; p.d = 100000; p.d_local = 100000; p.n_sweep_wg = nwg; p.ps = ps;
  p.coef_update = 1; p.eta_full = 1; p.pbar_predict = 1; p.track_g = 1; p.alpha = p.beta = 1.0; p.rho_mean = 1.0;
  // the stage loads theta / gradsum / Adam moments unconditionally (RM entries each, masked afterwards): they must point at memory
  double* th; hipMalloc((void**)&th, 4 * RM * 8); hipMemset(th, 0, 4 * RM * 8);
  p.theta = th; p.gradsum = th + RM; p.adam_m = th + 2 * RM; p.adam_v = th + 3 * RM; p.rp = (r + 3) / 4 * 4; p.nv = (r + 3) / 4; p.rows_per_wg = 224; p.solve_dual = PROF_R > 32 ? 1 : 0;
  for (int it = 0; it < 3; ++it) { if (RPAD == 64 && !getenv("NARROW")) psmf_serial_wide<<<1, SERIAL_WIDE_NT>>>(p, 0); else psmf_serial<RPAD><<<1, serial_threads(RPAD)>>>(p, 0); hipDeviceSynchronize(); }
  unsigned long long h[16]; hipMemcpy(h, reinterpret_cast<unsigned long long*>(part) + 4096, 16 * 8, hipMemcpyDeviceToHost);
  const char* nm[8] = {"issue loads", "partials->LDS reduce (+wait for loads)", "P+h col-reduce, mu", "gradient/robust scalars", "elementwise V,P,G + stores", "prep: mu_bar, Pbar, partials", "col-reduce V mu_bar", "s, <G,Pbar>, N, stores"};
  for (int q = 0; q < 8; ++q) printf("%-44s %6llu cycles\n", nm[q], h[q + 1] - h[q]);
  printf("total %llu cycles (stamps ~40 each; 100 MHz s_memtime? no: shader clock)\n", h[8] - h[0]);
  return 0;
}
