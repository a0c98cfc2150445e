// GPU-box diagnostic: per-phase s_memtime cycles of one column of the masked batched engine (config D shape:
// d = 19, r = 10, 40 % missing), averaged over the columns of a run; 50 replicas in one launch as in bench.py.
#define PSMF_IMPUTE_STAMPS 1
#define PSMF_IMPUTE_KERNEL_ONLY 1
#include "../rpsmf_amd/csrc/psmf_impute.hip"
#include <cstdio>
#include <vector>
using namespace psmf;
int main(int argc, char** argv) {
  const int d = 19, r = 10, B = argc > 1 ? atoi(argv[1]) : 50, n = argc > 2 ? atoi(argv[2]) : 20000, n_iter = 2, robust = argc > 3 ? atoi(argv[3]) : 0, ver = argc > 4 ? atoi(argv[4]) : 2;
  unsigned s = 7; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0; };
  std::vector<double> Y((size_t)n * d), C((size_t)B * d * r), X((size_t)B * n * r), V(r * r, 0.0), P(r * r, 0.0), Q(r * r, 0.0);
  std::vector<uint8_t> M((size_t)B * n * d), Mm((size_t)B * n * d);
  for (int i = 0; i < d; ++i) { double v = 10 * rnd(); for (int t = 0; t < n; ++t) { v += 0.05 * (rnd() - 0.5); Y[(size_t)t * d + i] = v; } }
  for (auto& c : C) c = rnd();
  for (auto& x : X) x = rnd();
  for (size_t i = 0; i < M.size(); ++i) { M[i] = rnd() > 0.4; Mm[i] = !M[i]; }
  for (int i = 0; i < r; ++i) { V[i * r + i] = 2.0; P[i * r + i] = 1.0; Q[i * r + i] = 0.1; }
  ImputeParams p{};
  p.d = d; p.n = n; p.r = r; p.n_iter = n_iter; p.robust = robust; p.method = robust; p.want_bands = 0; p.sig = 2.0; p.lambda0 = 1.8; p.rho0 = 10.0;
  double *dY, *dC, *dX, *dV, *dP, *dQ, *dE, *dF, *dI; uint8_t *dM, *dMm; int* dErr; unsigned long long* prof;
  hipMalloc(&dY, Y.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dX, X.size() * 8); hipMalloc(&dV, 800); hipMalloc(&dP, 800); hipMalloc(&dQ, 800);
  hipMalloc(&dE, B * n_iter * 8); hipMalloc(&dF, B * n_iter * 8); hipMalloc(&dI, B * 8); hipMalloc(&dM, M.size()); hipMalloc(&dMm, Mm.size()); hipMalloc(&dErr, B * 4);
  hipMalloc(&prof, (size_t)B * 4 * 8 * 8); hipMemset(prof, 0, (size_t)B * 4 * 8 * 8);
  hipMemcpy(dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dV, V.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(dP, P.data(), r * r * 8, hipMemcpyHostToDevice); hipMemcpy(dQ, Q.data(), r * r * 8, hipMemcpyHostToDevice);
  hipMemcpy(dM, M.data(), M.size(), hipMemcpyHostToDevice); hipMemcpy(dMm, Mm.data(), Mm.size(), hipMemcpyHostToDevice);
  p.Yorg = dY; p.M = dM; p.Mmiss = dMm; p.C = dC; p.X = dX; p.V0 = dV; p.P0 = dP; p.Q0 = dQ; p.Epred = dE; p.Efull = dF; p.inside = dI; p.err = dErr; p.prof = prof; p.q_iso = argc > 5 ? atoi(argv[5]) : 1;
  const size_t lds = ver == 3 ? impute3_lds_bytes(d, r) : impute2_lds_bytes(d, r);      // (version 1, round 1's loop, was removed in round 5)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  if (ver == 3) { void* args[] = {&p}; hipLaunchKernel(impute3_kernel(d), dim3(B), dim3(WG), args, lds, 0); } else psmf_impute_kernel2<<<B, WG, lds>>>(p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)B * 4 * 8); hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
  const double cols = (double)n * n_iter;
  printf("%d replicas x %d columns x %d passes (%s): %.2f ms, %.2f us per column per replica (with stamps)\n", B, n, n_iter, robust ? "rPSMF" : "PSMF", ms, 1e3 * ms / cols);
  const char* nm1[8] = {"A residual rows, w = V x (+barrier)", "B augmented masked Gram, reduce (2 barriers)", "C two sweep inversions (LDS, barriers)", "D x update, omega / phi",
                        "E C update, bands, metrics (+barrier)", "F V, P, Q update (+barrier)", "-", "loop back-edge, prefetch issue"};
  const char* nm2[8] = {"P1 work (rows / w, s)", "   wait at barrier 1", "P2 work (MFMA Gram / wave 0: P + Q, kappa)", "   wait at barrier 2",
                        "P3a work (wave 0: G, eta, N, phi)", "   wait at barrier 3", "P3b sweeps, x, omega (wave 0) / P4 C, V, bands (waves 1-3)", "   wait at barrier 4"};
  const char* nm3[8] = {"P1 work (wave 1: rows / 2: X store / 3: w, s)", "   wait at barrier 1", "operands -> registers", "own Gram, scalars",
                        "sweep (waves 0, 1)", "rest: x, omega, P (0) / bands (1) / C, V (2, 3)", "   wait at barrier 2", "-"};
  const char** nm = ver == 1 ? nm1 : (ver == 3 ? nm3 : nm2);
  for (int w : {0, 1, 2, 3}) { printf("replica 0, wave %d: shader-clock cycles per column (s_memtime)\n", w); for (int q = 0; q < 8; ++q) if (nm[q][0] != '-') printf("   %-56s %8.0f\n", nm[q], (double)h[(size_t)w * 8 + q] / cols); }
  return 0;
}
