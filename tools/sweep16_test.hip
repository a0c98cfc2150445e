// GPU-box check of wave_sweep16m (psmf_impute3.hip): sweep of a random SPD matrix augmented with a column, against the host inverse.
#define PSMF_IMPUTE_KERNEL_ONLY 1
#include "../rpsmf_amd/csrc/psmf_impute.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace psmf;
__global__ void k_sweep(double* M, int r2, int* badout) {
  const int lane = threadIdx.x, lk = lane >> 4, lr = lane & 15;
  double A[4];
  for (int q = 0; q < 4; ++q) A[q] = M[(lk + 4 * q) * 16 + lr];
  Sw16K c; sw16k_init(c, lk, lr);
  bool bad = false;
  wave_sweep16m(A, r2, c, bad);
  for (int q = 0; q < 4; ++q) M[(lk + 4 * q) * 16 + lr] = A[q];
  if (lane == 0) *badout = bad;
}
int main() {
  for (int r : {10, 7, 14, 2}) {
    const int r2 = r + (r & 1);
    std::vector<double> B(r * r), M(256, 0.0), b(r);
    unsigned s = 5 + r; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (auto& v : B) v = rnd();
    for (auto& v : b) v = rnd();
    for (int i = 0; i < 16; ++i) M[i * 16 + i] = 1.0;
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) { double a = (i == j) ? 0.5 : 0.0; for (int k = 0; k < r; ++k) a += B[i * r + k] * B[j * r + k]; M[i * 16 + j] = a; }
    for (int i = 0; i < r; ++i) { M[i * 16 + r2] = b[i]; M[r2 * 16 + i] = b[i]; }
    std::vector<double> M0 = M;
    double* d; int* db; hipMalloc(&d, 256 * 8); hipMalloc(&db, 4); hipMemcpy(d, M.data(), 256 * 8, hipMemcpyHostToDevice);
    k_sweep<<<1, 64>>>(d, r2, db); hipDeviceSynchronize();
    int bad; hipMemcpy(M.data(), d, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, db, 4, hipMemcpyDeviceToHost);
    // check: (-A) * M0 = I on the r x r block; column r2 = inv * b
    double e1 = 0, e2 = 0;
    for (int i = 0; i < r; ++i) {
      for (int j = 0; j < r; ++j) { double a = 0; for (int k = 0; k < r; ++k) a += -M[i * 16 + k] * M0[k * 16 + j]; e1 = fmax(e1, fabs(a - (i == j))); }
      double z = 0; for (int k = 0; k < r; ++k) z += -M[i * 16 + k] * b[k]; e2 = fmax(e2, fabs(z - M[i * 16 + r2]));
    }
    printf("r=%d bad=%d  |(-A) M - I| = %.2e  |column - inv b| = %.2e  corner %.6f\n", r, bad, e1, e2, M[r2 * 16 + r2]);
  }
  return 0;
}
