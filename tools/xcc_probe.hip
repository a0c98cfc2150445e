// GPU-box diagnostic: which XCD / SE / CU a one-workgroup kernel lands on for CU-masked streams (bits [lo, hi) set).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void where(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[0] = xcc; out[1] = hw; }
}
int main(int argc, char** argv) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
  unsigned* d; hipMalloc(&d, 8);
  auto probe = [&](int lo, int hi) {
    uint32_t m[16] = {0};
    for (int i = lo; i < hi; ++i) m[i >> 5] |= 1u << (i & 31);
    hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, words, m) != hipSuccess) { printf("mask [%d,%d): stream creation failed\n", lo, hi); return; }
    printf("mask [%3d,%3d):", lo, hi);
    for (int it = 0; it < 6; ++it) {
      where<<<1, 512, 0, s>>>(d); hipStreamSynchronize(s);
      unsigned h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
      printf("  xcc %u se %u sh %u cu %u", h[0] & 0xf, (h[1] >> 13) & 7, (h[1] >> 12) & 1, (h[1] >> 8) & 0xf);
    }
    printf("\n");
    hipStreamDestroy(s);
  };
  printf("%d CUs\n", ncu);
  probe(0, 8); probe(8, 16); probe(0, 1); probe(1, 2); probe(7, 8); probe(8, 9); probe(9, 10); probe(16, 17); probe(32, 33); probe(64, 65); probe(16, ncu);
  return 0;
}
