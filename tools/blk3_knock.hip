// GPU-box diagnostic: time of a steady block of psmf_blk_filter3 with pieces knocked out (-DF3_KNOCK=mask, see psmf_blk3.hip):
// what each piece costs on the critical path.  tools/knock.sh runs the set.
#define F3_PREDICT 0
#include "../rpsmf_amd/csrc/psmf_blk3.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace psmf;
int main() {
  const int r = 32, nb = 32;
  DevState* st; hipMalloc((void**)&st, sizeof(DevState)); hipMemset(st, 0, sizeof(DevState));
  std::vector<double> I(r * r, 0.0), Q(r * r, 0.0); for (int i = 0; i < r; ++i) { I[i * r + i] = 1.0; Q[i * r + i] = 0.1; }
  hipMemcpy(st->V, I.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(st->P, I.data(), r * r * 8, hipMemcpyHostToDevice);
  hipMemcpy(st->Q, Q.data(), r * r * 8, hipMemcpyHostToDevice);
  double one = 1.0; hipMemcpy(&st->rho, &one, 8, hipMemcpyHostToDevice);
  // a filter-like Gram: Z = [C0 | y_1 .. y_32], y_t = Ctrue x_t + noise, x_t a smooth trajectory
  const int dd = 4096;
  std::vector<double> Z((size_t)dd * RB), K(RB * RB, 0.0), Ct((size_t)dd * r);
  unsigned s = 1; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
  for (auto& c : Ct) c = 2.0 * rnd();
  for (int i = 0; i < dd; ++i) for (int c = 0; c < r; ++c) Z[(size_t)i * RB + c] = 0.2 * rnd();
  for (int t = 0; t < nb; ++t) {
    std::vector<double> x(r); for (int c = 0; c < r; ++c) x[c] = std::cos(0.01 * (c + 1) * (t + 1) + c);
    for (int i = 0; i < dd; ++i) { double acc = 0; for (int c = 0; c < r; ++c) acc += Ct[(size_t)i * r + c] * x[c]; Z[(size_t)i * RB + r + t] = acc + 0.6 * rnd(); }
  }
  for (int a = 0; a < RB; ++a) for (int c = 0; c < RB; ++c) { double acc = 0; for (int i = 0; i < dd; ++i) acc += Z[(size_t)i * RB + a] * Z[(size_t)i * RB + c]; K[a * RB + c] = acc; }
  double *dK, *dA, *dB, *dKp; hipMalloc((void**)&dK, RB * RB * 8); hipMalloc((void**)&dA, RB * RM * 8); hipMalloc((void**)&dB, RB * RB * 8); hipMalloc((void**)&dKp, 1 << 20);
  hipMemcpy(dK, K.data(), RB * RB * 8, hipMemcpyHostToDevice);
  BlockParams b{}; b.sp.st = st; b.sp.r = r; b.sp.d = 4096; b.sp.d_local = 4096; b.sp.use_ns = 1; b.sp.coef_update = 1; b.sp.eta_full = 1; b.sp.pbar_predict = 1;
  b.sp.alpha = b.sp.beta = 1.0; b.sp.ns_far2 = 0.09; b.sp.ns_tol2 = getenv("TOL") ? atof(getenv("TOL")) * atof(getenv("TOL")) : 9e-14; b.K = dK; b.Acoef = dA; b.Bcoef = dB; b.Kpart = dKp; b.k0 = 0; b.nb = nb;
  const size_t lds = blk_filter3_lds_bytes();
  hipFuncSetAttribute((const void*)psmf_blk_filter3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 2; ++it) { psmf_blk_filter3<<<1, F3_NT, lds>>>(b); hipDeviceSynchronize(); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); psmf_blk_filter3<<<1, F3_NT, lds>>>(b); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  for (int it = 0; it < 40; ++it) { psmf_blk_filter3<<<1, F3_NT, lds>>>(b); }   // carry the state: steady regime
  hipDeviceSynchronize();
  hipEventRecord(e0); psmf_blk_filter3<<<1, F3_NT, lds>>>(b); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("F3_KNOCK=%d: block of %d steps: %.1f us = %.3f us/step\n", F3_KNOCK, nb, ms * 1e3, ms * 1e3 / nb);
  return 0;
}
