// GPU-box diagnostic: where a pivot step of the symmetric sweep spends its cycles (in-kernel stamps).
#include "../rpsmf_amd/csrc/psmf_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace psmf;
#define STAMP(x) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); __builtin_amdgcn_sched_barrier(0); }

template <int RPAD>
__global__ __launch_bounds__(WG) void prof(double* Ain, int r, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* rowbuf = reinterpret_cast<double*>(smem_raw);
  constexpr int RG = WG / RPAD; constexpr int RGW = RG / 4 > 0 ? RG / 4 : 1;
  constexpr int M = (RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1;
  const int tid = threadIdx.x, c = tid % RPAD, rg = tid / RPAD, wv = tid >> 6;
  double A[M];
  for (int m = 0; m < M; ++m) { int i = rg + m * RG; A[m] = (i < r && c < r) ? Ain[i * r + c] : 0.0; }
  const bool con = c < r; const int cc = con ? c : r - 1;
  int ic[M]; for (int m = 0; m < M; ++m) ic[m] = min(rg + m * RG, r - 1);
  if (rg == 0 && con) rowbuf[c] = A[0];
  __syncthreads();
  unsigned long long t0, t1, t2, t3, t4, t5, acc[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < r; ++k) {
    STAMP(t0);
    const double* rb = rowbuf + (k & 1) * RM; double* rbn = rowbuf + ((k + 1) & 1) * RM;
    const double d = rb[k]; const double vc = rb[cc];
    double vi[M]; for (int m = 0; m < M; ++m) vi[m] = rb[ic[m]];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t1);
    const double dinv = fast_rcp(d);
    asm volatile("" :: "v"(dinv));
    STAMP(t2);
    const double vcd = vc * dinv; const bool ck = (c == k);
    for (int m = 0; m < M; ++m) { const double t = fma(-vi[m], vcd, A[m]); A[m] = ck ? vi[m] * dinv : t; }
    if (((k % RG) / RGW) == wv) { const double rowk = ck ? -dinv : vcd; for (int m = 0; m < M; ++m) A[m] = (rg + m * RG == k) ? rowk : A[m]; }
    for (int m = 0; m < M; ++m) asm volatile("" :: "v"(A[m]));
    STAMP(t3);
    if (k + 1 < r && (((k + 1) % RG) / RGW) == wv) { double nxt = 0.0; for (int m = 0; m < M; ++m) nxt = (rg + m * RG == k + 1) ? A[m] : nxt; if (con && ((k + 1) % RG) == rg) rbn[c] = nxt; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t4);
    __builtin_amdgcn_s_barrier();
    STAMP(t5);
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4;
  }
  if ((tid & 63) == 0) for (int q = 0; q < 5; ++q) out[wv * 5 + q] = acc[q];
  for (int m = 0; m < M; ++m) { int i = rg + m * RG; if (i < r && c < r) Ain[i * r + c] = A[m]; }
}

int main() {
  const int r = 32;
  std::vector<double> P(r * r);
  for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) P[i * r + j] = (i == j ? 1.0 : 0.0) + 0.3 / (1.0 + std::abs(i - j));
  double* dA; hipMalloc((void**)&dA, r * r * 8); hipMemcpy(dA, P.data(), r * r * 8, hipMemcpyHostToDevice);
  unsigned long long* out; hipMalloc((void**)&out, 20 * 8);
  prof<32><<<1, WG, 4096>>>(dA, r, out); hipDeviceSynchronize();
  hipMemcpy(dA, P.data(), r * r * 8, hipMemcpyHostToDevice);
  prof<32><<<1, WG, 4096>>>(dA, r, out); hipDeviceSynchronize();
  unsigned long long h[20]; hipMemcpy(h, out, 160, hipMemcpyDeviceToHost);
  const char* names[5] = {"lds-read", "rcp", "update", "publish+wait", "barrier"};
  for (int w = 0; w < 4; ++w) { printf("wave %d:", w); for (int q = 0; q < 5; ++q) printf("  %s %.0f", names[q], (double)h[w * 5 + q] / r); printf("  (cycles per pivot, stamps included ~40 each)\n"); }
  return 0;
}
