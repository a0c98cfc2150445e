// GPU-box diagnostic: per-stage cycles per wave of the small-rank block filter psmf_blk_filter7 (17 <= r <= 32; stamps via PSMF_BLK_STAMPS).
//   hipcc -O3 --offload-arch=gfx950 -I include -o tools/bin/blk32_prof tools/blk32_prof.hip
//   tools/bin/blk32_prof [r] [kind: 0 rw, 1 cos-phase, 4 fourier] [terms] [recursive] [dual: 1 = the two inversions side by side (random walk)]
#define PSMF_BLK_STAMPS 1
#include "../rpsmf_amd/csrc/psmf_blk32.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace psmf;

int main(int argc, char** argv) {
  const int r = argc > 1 ? atoi(argv[1]) : 20, kind = argc > 2 ? atoi(argv[2]) : 1, terms = argc > 3 ? atoi(argv[3]) : 1, rec = argc > 4 ? atoi(argv[4]) : 0, dual = argc > 5 ? atoi(argv[5]) : 0;
  const int nb = (64 - r < 48) ? 64 - r : 48;
  DevState* st; hipMalloc((void**)&st, sizeof(DevState)); hipMemset(st, 0, sizeof(DevState));
  std::vector<double> I(r * r, 0.0), Q(r * r, 0.0); for (int i = 0; i < r; ++i) { I[i * r + i] = 1.0; Q[i * r + i] = 0.1; }
  hipMemcpy(st->V, I.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(st->P, I.data(), r * r * 8, hipMemcpyHostToDevice);
  hipMemcpy(st->Q, Q.data(), r * r * 8, hipMemcpyHostToDevice);
  double one = 1.0; hipMemcpy(&st->rho, &one, 8, hipMemcpyHostToDevice);
  const int dd = 4096;
  std::vector<double> Z((size_t)dd * RB, 0.0), K(RB * RB, 0.0), Ct((size_t)dd * r);
  unsigned s = 1; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
  for (auto& c : Ct) c = 2.0 * rnd();
  for (int i = 0; i < dd; ++i) for (int c = 0; c < r; ++c) Z[(size_t)i * RB + c] = 0.2 * rnd();
  for (int t = 0; t < nb; ++t) {
    std::vector<double> x(r); for (int c = 0; c < r; ++c) x[c] = std::cos(0.01 * (c + 1) * (t + 1) + c);
    for (int i = 0; i < dd; ++i) { double acc = 0; for (int c = 0; c < r; ++c) acc += Ct[(size_t)i * r + c] * x[c]; Z[(size_t)i * RB + r + t] = acc + 0.6 * rnd(); }
  }
  for (int a = 0; a < RB; ++a) for (int c = 0; c < RB; ++c) { double acc = 0; for (int i = 0; i < dd; ++i) acc += Z[(size_t)i * RB + a] * Z[(size_t)i * RB + c]; K[a * RB + c] = acc; }
  double *dK, *dA, *dB, *dKp, *th; hipMalloc((void**)&dK, RB * RB * 8); hipMalloc((void**)&dA, RB * RM * 8); hipMalloc((void**)&dB, RB * RB * 8); hipMalloc((void**)&dKp, 1 << 20);
  const int nth = dyn_n_theta(kind, 3, terms, r);
  const size_t cap = nth > 64 ? nth : 64;
  hipMalloc((void**)&th, 4 * cap * 8); hipMemset(th, 0, 4 * cap * 8);
  std::vector<double> thh(cap, 0.0); for (int i = 0; i < nth; ++i) thh[i] = 0.05 + 0.1 * (rnd() + 0.5);
  hipMemcpy(th, thh.data(), cap * 8, hipMemcpyHostToDevice);
  hipMemcpy(dK, K.data(), RB * RB * 8, hipMemcpyHostToDevice);
  BlockParams b{}; b.sp.st = st; b.sp.r = r; b.sp.d = dd; b.sp.d_local = dd; b.sp.use_ns = 1; b.sp.coef_update = 1; b.sp.eta_full = 1; b.sp.pbar_predict = 1;
  b.sp.alpha = b.sp.beta = 1.0; b.sp.dyn_kind = kind; b.sp.dyn_flags = 3; b.sp.dyn_terms = terms; b.sp.n_theta = nth; b.sp.theta = th; b.sp.gradsum = th + cap; b.sp.adam_m = th + 2 * cap; b.sp.adam_v = th + 3 * cap;
  b.sp.recursive = rec; b.sp.update_every = 1; b.sp.lr = 1e-3; b.sp.b1 = 0.9; b.sp.b2 = 0.999;
  b.dual6 = dual; b.K = dK; b.Acoef = dA; b.Bcoef = dB; b.Kpart = dKp; b.k0 = 0; b.nb = nb;
  const size_t lds = blk_filter_lds_bytes();
  hipFuncSetAttribute((const void*)psmf_blk_filter7, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)psmf_blk_filter6d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  auto go = [&]() { if (dual) psmf_blk_filter7<<<1, WG, lds>>>(b); else psmf_blk_filter7<<<1, WG, lds>>>(b); };
  for (int it = 0; it < 20; ++it) go();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemset(dKp, 0, 4096); go(); hipDeviceSynchronize();
  unsigned long long h[4 * 12]; hipMemcpy(h, dKp, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[8] = {"dynamics forward", "A: w, s, Pbar, eta (0) / b, Ka, h, g_f (1)", "A: first sweep (0)", "barrier A|B", "B: second sweep (0)", "B: updates (0)", "B: dyn_backward (1-3) / barrier wait (0)", "barriers, mu, adam"};
  printf("r=%d kind=%d terms=%d recursive=%d n_theta=%d: block of %d steps: %.1f us = %.2f us/step (%s)\n", r, kind, terms, rec, nth, nb, ms * 1e3, ms * 1e3 / nb, hipGetErrorString(hipGetLastError()));
  for (int w = 0; w < 4; ++w) {
    double tot = 0; for (int q = 0; q < 8; ++q) tot += (double)h[w * 12 + q] / nb;
    printf(" wave %d (total %.0f cycles/step)\n", w, tot);
    for (int q = 0; q < 8; ++q) printf("   %-50s %7.0f\n", nm[q], (double)h[w * 12 + q] / nb);
  }
  return 0;
}
