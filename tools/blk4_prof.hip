// GPU-box diagnostic: per-phase cycles of psmf_blk_filter4 (stamps via PSMF_BLK_STAMPS) on a synthetic block, carried state.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -o tools/bin/blk4_prof tools/blk4_prof.hip
//   [NS=0|1] [TOL=x] [Q=q] tools/bin/blk4_prof [r]
#define PSMF_BLK_STAMPS 1
#include "../rpsmf_amd/csrc/psmf_blk3.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace psmf;
int main(int argc, char** argv) {
  const int r = argc > 1 ? atoi(argv[1]) : 20, nb = (64 - r < 48) ? 64 - r : 48;
  const double q = getenv("Q") ? atof(getenv("Q")) : 0.1;
  DevState* st; hipMalloc((void**)&st, sizeof(DevState)); hipMemset(st, 0, sizeof(DevState));
  std::vector<double> I(r * r, 0.0), Q(r * r, 0.0), V(r * r, 0.0); for (int i = 0; i < r; ++i) { I[i * r + i] = 1.0; Q[i * r + i] = q; V[i * r + i] = 0.1; }
  hipMemcpy(st->V, V.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(st->P, I.data(), r * r * 8, hipMemcpyHostToDevice);
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
  hipMalloc((void**)&th, 4 * 64 * 8); hipMemset(th, 0, 4 * 64 * 8);
  std::vector<double> thh(64, 0.0); for (int i = 0; i < r; ++i) thh[i] = 0.05 + 0.1 * (rnd() + 0.5);
  hipMemcpy(th, thh.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(dK, K.data(), RB * RB * 8, hipMemcpyHostToDevice);
  BlockParams b{}; b.sp.st = st; b.sp.r = r; b.sp.d = dd; b.sp.d_local = dd; b.sp.use_ns = getenv("NS") ? atoi(getenv("NS")) : 1; b.sp.coef_update = 1; b.sp.eta_full = 1; b.sp.pbar_predict = 1;
  const int mode5 = getenv("MODE") && atoi(getenv("MODE")) == 5;
  if (mode5) { b.sp.coef_update = 0; b.sp.eta_full = 0; b.sp.pbar_predict = 0; }
  b.sp.alpha = b.sp.beta = 1.0; b.sp.ns_predict = 7; b.sp.ns_far2 = 0.09; b.sp.ns_tol2 = getenv("TOL") ? atof(getenv("TOL")) * atof(getenv("TOL")) : 9e-8;
  b.sp.dyn_kind = DYN_COS_PHASE; b.sp.n_theta = r; b.sp.theta = th; b.sp.gradsum = th + 64; b.sp.adam_m = th + 128; b.sp.adam_v = th + 192; b.sp.update_every = 1;
  b.K = dK; b.Acoef = dA; b.Bcoef = dB; b.Kpart = dKp; b.k0 = 0; b.nb = nb; b.last = 1;
  const size_t lds = blk_filter3_lds_bytes();
  hipFuncSetAttribute((const void*)psmf_blk_filter4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)psmf_blk_filter4s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)psmf_blk_filter5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  auto go = [&]() { if (mode5) psmf_blk_filter5<<<1, F3_NT, lds>>>(b); else if (r > 16) psmf_blk_filter4<<<1, F3_NT, lds>>>(b); else psmf_blk_filter4s<<<1, F3_NT, lds>>>(b); };
  for (int it = 0; it < 40; ++it) go();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c0[8], c1[8]; hipMemcpy(c0, st->cnt, sizeof(c0), hipMemcpyDeviceToHost);
  hipMemset(dKp, 0, 4096); go(); hipDeviceSynchronize();
  hipMemcpy(c1, st->cnt, sizeof(c1), hipMemcpyDeviceToHost);
  unsigned long long h[8 * 12]; hipMemcpy(h, dKp, sizeof(h), hipMemcpyDeviceToHost);
  printf("r=%d q=%g use_ns=%d: block of %d steps: %.1f us = %.2f us/step (%s); last block: ns %lld sweep %lld decides %lld failed %lld\n", r, q, b.sp.use_ns, nb, ms * 1e3, ms * 1e3 / nb,
         hipGetErrorString(hipGetLastError()), c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2], c1[3] - c0[3]);
  const char* nx[8] = {"-", "barrier waits", "phase 1 (M, it 0)", "phase 2 (it 1, G)", "more iterations + sweep", "phase F", "Y control", "phase 0"};
  for (int w : {0, 2, 4, 5, 6, 7}) {
    if (mode5 && w < 4) continue;
    printf("wave %d (%s):", w, w < 2 ? "X" : (w < 4 ? "Y" : "V"));
    if (w < 4) { for (int qq = 1; qq < 8; ++qq) printf("  %s %.0f", nx[qq], (double)h[w * 12 + qq] / nb); }
    else { for (int qq = 0; qq < 5; ++qq) printf("  [%d] %.0f", qq, (double)h[w * 12 + qq] / nb); }
    printf("\n");
  }
  return 0;
}
