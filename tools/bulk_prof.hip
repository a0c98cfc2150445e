// GPU-box diagnostic: in-kernel time stamps of the streaming bulk kernels (define BK_STAMPS before including).
#define BK_STAMPS 1
#include "../rpsmf_amd/csrc/psmf_bulk.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace psmf;
int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 100000, r = 32, nb = 32;
  float *C, *Y, *YP; double *XGpart, *A, *B; long long* stamps;
  hipMalloc(&C, (size_t)d * r * 4); hipMalloc(&Y, (size_t)3 * nb * d * 4); hipMalloc(&YP, (size_t)3 * nb * d * 4);
  hipMalloc(&XGpart, (size_t)256 * 8192 * 8); hipMalloc(&A, 64 * 64 * 8); hipMalloc(&B, 64 * 64 * 8); hipMalloc(&stamps, 256 * 8 * 8 * 8);
  hipMemset(C, 0, (size_t)d * r * 4); hipMemset(Y, 0, (size_t)3 * nb * d * 4); hipMemset(A, 0, 64 * 64 * 8); hipMemset(B, 0, 64 * 64 * 8);
  BlockParams b{}; b.sp.C = C; b.sp.Y = Y; b.sp.YP = YP; b.sp.store_yp = 1; b.sp.r = r; b.sp.rp = 32; b.sp.d = d; b.sp.d_local = d; b.sp.series_t0 = 0;
  b.k0 = 0; b.nb = nb; b.k1 = nb; b.nb1 = nb; b.XGpart = XGpart; b.Acoef = A; b.Bcoef = B; b.Kpart = (double*)stamps;
  hipFuncSetAttribute((const void*)psmf_blk_xgram2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blk_xgram2_lds_bytes());
  hipFuncSetAttribute((const void*)psmf_blk_apply2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blk_apply2_lds_bytes());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0);
      if (which == 0) psmf_blk_xgram2<2><<<BK_XG_WG, BK_NT, blk_xgram2_lds_bytes()>>>(b);
      else psmf_blk_apply2<2><<<256, BK_NT, blk_apply2_lds_bytes()>>>(b);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<long long> h(256 * 64); hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    // per wave: [0] start, [1] first tile loaded/zeroed, [2] loop end, [3] after reduce / end
    double s1 = 0, s2 = 0, s3 = 0, mx = 0; long long tmin = 1LL << 62, tmax = 0;
    for (int w = 0; w < 256 * 8; ++w) { const long long* p = &h[w * 8]; s1 += p[1] - p[0]; s2 += p[2] - p[1]; s3 += p[3] - p[2]; if (p[3] - p[0] > mx) mx = p[3] - p[0]; if (p[0] < tmin) tmin = p[0]; if (p[3] > tmax) tmax = p[3]; }
    double clk = 0;
    for (int w = 0; w < 256 * 8; ++w) { const long long* p = &h[w * 8]; clk += (double)(p[6] - p[5]) / (double)(p[2] - p[1] > 0 ? p[2] - p[1] : 1); }
    printf("   shader clock in the tile loop (s_memtime / s_memrealtime): %.0f MHz\n", 100.0 * clk / (256 * 8));
    const double n = 256 * 8;
    printf("%s: %.1f us by events; per wave (10 ns ticks -> us): prologue %.2f, tile loop %.2f, epilogue %.2f; slowest wave %.2f; first start -> last end %.2f\n",
           which == 0 ? "xgram2" : "apply2", best * 1e3, 0.01 * s1 / n, 0.01 * s2 / n, 0.01 * s3 / n, 0.01 * mx, 0.01 * (tmax - tmin));
  }
  return 0;
}
