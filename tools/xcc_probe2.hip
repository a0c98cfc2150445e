// GPU-box diagnostic: XCD of one-workgroup kernels on several CU-masked streams alive at once (stability per stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void where(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[0] = xcc & 0xf; out[1] = hw; }
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
  const int NS = 8, NL = 8;
  hipStream_t plain[3]; for (auto& p : plain) hipStreamCreateWithFlags(&p, hipStreamNonBlocking);   // like the library's other streams
  hipStream_t s[NS];
  unsigned* d; hipMalloc(&d, NS * NL * 8);
  for (int i = 0; i < NS; ++i) {
    uint32_t m[16] = {0};
    const int lo = (i & 1) ? 8 : 0;
    for (int b = lo; b < lo + 8; ++b) m[b >> 5] |= 1u << (b & 31);
    hipExtStreamCreateWithCUMask(&s[i], words, m);
  }
  for (int l = 0; l < NL; ++l)
    for (int i = 0; i < NS; ++i) where<<<1, 512, 0, s[i]>>>(d + (i * NL + l) * 2);
  hipDeviceSynchronize();
  unsigned h[NS * NL * 2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < NS; ++i) {
    printf("stream %d mask [%d,%d):", i, (i & 1) ? 8 : 0, (i & 1) ? 16 : 8);
    for (int l = 0; l < NL; ++l) printf("  x%u s%u c%u", h[(i * NL + l) * 2], (h[(i * NL + l) * 2 + 1] >> 13) & 7, (h[(i * NL + l) * 2 + 1] >> 8) & 0xf);
    printf("\n");
  }
  // second round after a pause
  for (int l = 0; l < NL; ++l)
    for (int i = 0; i < NS; ++i) { where<<<1, 512, 0, s[i]>>>(d + (i * NL + l) * 2); hipStreamSynchronize(s[i]); }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < NS; ++i) {
    printf("stream %d (synchronised launches):", i);
    for (int l = 0; l < NL; ++l) printf("  x%u s%u c%u", h[(i * NL + l) * 2], (h[(i * NL + l) * 2 + 1] >> 13) & 7, (h[(i * NL + l) * 2 + 1] >> 8) & 0xf);
    printf("\n");
  }
  return 0;
}
