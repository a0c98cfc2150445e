// GPU-box diagnostic: the two streaming bulk kernels of the blocked engine on data that comes from HBM -- consecutive
// launches walk consecutive blocks of a long series, as in the pipeline (tools/bulk_prof.hip repeats ONE block, which at
// d <= 1e6 stays in the 256 MB Infinity Cache).   usage: bulk_stream [d] [blocks] [xgram workgroups] [apply workgroups] [reserved CU-mask bits: the library's bulk stream = 8]
#include "../rpsmf_amd/csrc/psmf_bulk.hip"
#include <cstdio>
#include <cstdlib>
using namespace psmf;
int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 1000000, nrot = argc > 2 ? atoi(argv[2]) : 8, r = 32, nb = 32;
  const int gx = argc > 3 ? atoi(argv[3]) : BK_XG_WG, ga = argc > 4 ? atoi(argv[4]) : 256;
  float *C, *Y, *YP; double *XGpart, *A, *B;
  const size_t ysz = (size_t)(nrot + 1) * nb * d * 4;
  hipMalloc(&C, (size_t)d * r * 4); hipMalloc(&Y, ysz); hipMalloc(&YP, ysz);
  hipMalloc(&XGpart, (size_t)1024 * 8192 * 8); hipMalloc(&A, 64 * 64 * 8); hipMalloc(&B, 64 * 64 * 8);
  hipMemset(C, 0, (size_t)d * r * 4); hipMemset(Y, 0, ysz); hipMemset(A, 0, 64 * 64 * 8); hipMemset(B, 0, 64 * 64 * 8);
  BlockParams b{}; b.sp.C = C; b.sp.Y = Y; b.sp.YP = YP; b.sp.store_yp = 1; b.sp.r = r; b.sp.rp = 32; b.sp.d = d; b.sp.d_local = d; b.sp.series_t0 = 0;
  b.nb = nb; b.nb1 = nb; b.XGpart = XGpart; b.Acoef = A; b.Bcoef = B;
  hipFuncSetAttribute((const void*)psmf_blk_xgram2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blk_xgram2_lds_bytes());
  hipFuncSetAttribute((const void*)psmf_blk_apply2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blk_apply2_lds_bytes());
  const int nres = argc > 5 ? atoi(argv[5]) : 0;
  hipStream_t st = nullptr;
  if (nres > 0) {
    uint32_t mb[8] = {0};
    for (int i = nres; i < 256; ++i) mb[i >> 5] |= 1u << (i & 31);
    if (hipExtStreamCreateWithCUMask(&st, 8, mb) != hipSuccess) { printf("no masked stream\n"); return 1; }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes_x = 3.0 * d * 32 * 4, bytes_a = 4.0 * d * 32 * 4;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, st);
      for (int j = 0; j < nrot; ++j) {
        b.k0 = (long long)j * nb; b.k1 = b.k0 + nb;
        if (which == 0) psmf_blk_xgram2<2><<<gx, BK_NT, blk_xgram2_lds_bytes(), st>>>(b);
        else psmf_blk_apply2<2><<<ga, BK_NT, blk_apply2_lds_bytes(), st>>>(b);
      }
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s d=%d: %.1f us per block, %.2f TB/s (%d workgroups, %d mask bits reserved)\n", which == 0 ? "xgram2" : "apply2", d, 1e3 * ms / nrot,
                      (which == 0 ? bytes_x : bytes_a) / (1e-3 * ms / nrot) * 1e-12, which == 0 ? gx : ga, nres);
    }
  }
  return 0;
}
