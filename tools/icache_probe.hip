// Cost of a loop back-edge as a function of the loop body's code size (instruction-cache reach), gfx950.
//   icache_probe [blocks] [iters]
// Each kernel: `iters` trips of a loop whose body is NB bytes of s_nop; s_memtime at the loop top and at the loop bottom.
// Prints cycles per trip inside the body and across the back-edge, wave 0 of block 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define XSTR(x) #x
#define STR(x) XSTR(x)
#define KERNEL(NAME, NINSTR, BAR)                                                                       \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters) {                     \
    unsigned long long body = 0, edge = 0, t0, t1, tp;                                                  \
    STAMP(tp);                                                                                          \
    for (int i = 0; i < iters; ++i) {                                                                   \
      STAMP(t0);                                                                                        \
      asm volatile(".rept " STR(NINSTR) "\n\ts_nop 0\n\t.endr" ::: "memory");                          \
      if (BAR) __syncthreads();                                                                         \
      STAMP(t1);                                                                                        \
      body += t1 - t0;                                                                                  \
      if (i) edge += t0 - tp;                                                                           \
      tp = t1;                                                                                          \
    }                                                                                                   \
    if ((threadIdx.x & 63) == 0) {                                                                      \
      out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = body;                                            \
      out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = edge;                                        \
    }                                                                                                   \
  }

KERNEL(k_256, 64, 0)
KERNEL(k_4k, 1024, 0)
KERNEL(k_8k, 2048, 0)
KERNEL(k_16k, 4096, 0)
KERNEL(k_24k, 6144, 0)
KERNEL(k_32k, 8192, 0)
KERNEL(k_48k, 12288, 0)
KERNEL(b_256, 64, 1)
KERNEL(b_8k, 2048, 1)
KERNEL(b_32k, 8192, 1)


// ---- second family: 64-instruction body, something extra in the body (B) or between the bottom stamp and the back-edge (L)
#define KERNEL2(NAME, BODYX, LATCHX)                                                                    \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters) {                     \
    __shared__ double sh[512];                                                                          \
    unsigned long long body = 0, edge = 0, t0, t1, tp;                                                  \
    double x = threadIdx.x, y = 1.0;                                                                    \
    sh[threadIdx.x] = x; sh[threadIdx.x + 256] = x;                                                     \
    __syncthreads();                                                                                    \
    STAMP(tp);                                                                                          \
    for (int i = 0; i < iters; ++i) {                                                                   \
      STAMP(t0);                                                                                        \
      asm volatile(".rept 64\n\ts_nop 0\n\t.endr" ::: "memory");                                     \
      BODYX;                                                                                            \
      __syncthreads();                                                                                  \
      STAMP(t1);                                                                                        \
      body += t1 - t0;                                                                                  \
      if (i) edge += t0 - tp;                                                                           \
      tp = t1;                                                                                          \
      LATCHX;                                                                                           \
    }                                                                                                   \
    if (x == -1.0 || y == -1.0) out[0] = 1;                                                             \
    if ((threadIdx.x & 63) == 0) {                                                                      \
      out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = body;                                            \
      out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = edge;                                        \
    }                                                                                                   \
  }
#define NOTHING
KERNEL2(x_plain, NOTHING, NOTHING)
KERNEL2(x_valu_latch, NOTHING, asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x)))
KERNEL2(x_valu_f64_body, x = fma(x, 1.0000001, 1e-9), asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x)))
KERNEL2(x_lds_body, { x += sh[(threadIdx.x + i) & 511]; sh[threadIdx.x] = x; }, NOTHING)
KERNEL2(x_lds_valu, { x += sh[(threadIdx.x + i) & 511]; sh[threadIdx.x] = x; }, asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x)))
KERNEL2(x_two_barriers, { __syncthreads(); x += sh[(threadIdx.x + i) & 511]; }, NOTHING)
KERNEL2(x_div_branch, { if ((threadIdx.x >> 6) == (i & 3)) x += sh[(threadIdx.x + i) & 511]; }, NOTHING)

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned long long* d;
  hipMalloc(&d, blocks * 8 * 8);
  std::vector<unsigned long long> h(blocks * 8);
  struct { const char* name; void (*k)(unsigned long long*, int); int ninstr; } ks[] = {
      {"256 B", k_256, 64}, {"4 KB", k_4k, 1024}, {"8 KB", k_8k, 2048}, {"16 KB", k_16k, 4096}, {"24 KB", k_24k, 6144},
      {"32 KB", k_32k, 8192}, {"48 KB", k_48k, 12288},
      {"256 B + barrier", b_256, 64}, {"8 KB + barrier", b_8k, 2048}, {"32 KB + barrier", b_32k, 8192}};
  printf("%d block(s) x 256 threads, %d trips; s_memtime ticks per trip (wave 0 of block 0)\n", blocks, iters);
  printf("%-18s %12s %12s %14s\n", "loop body", "body", "back-edge", "body/instr");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& e : ks) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("[%.3f ms] ", ms);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    printf("[%.0f ticks/us] ", (double)(h[0] + h[1]) / (ms * 1e3));
    printf("%-18s %12.1f %12.1f %14.3f\n", e.name, (double)h[0] / iters, (double)h[1] / (iters - 1), (double)h[0] / iters / e.ninstr);
  }
  struct { const char* name; void (*k)(unsigned long long*, int); } k2[] = {
      {"plain + barrier", x_plain}, {"VALU mov in latch", x_valu_latch}, {"f64 fma in body, mov in latch", x_valu_f64_body},
      {"LDS in body", x_lds_body}, {"LDS in body, mov in latch", x_lds_valu}, {"two barriers", x_two_barriers},
      {"wave-divergent LDS read", x_div_branch}};
  for (auto& e : k2) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%-34s %12.1f %12.1f\n", e.name, (double)h[0] / iters, (double)h[1] / (iters - 1));
  }
  return 0;
}
