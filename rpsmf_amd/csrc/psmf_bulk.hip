// Streaming versions of the two d-sized products of a pipelined block (float32 storage):
//
//   psmf_blk_xgram2   XG = [Z | Y_next]^T Y_next      reads C, Y_cur, Y_next once     (HBM / f64-MFMA bound)
//   psmf_blk_xreduce2 fixed-order sum of the per-workgroup partials
//   psmf_blk_apply2   [C_new | Y_hat] = Z [A | b_1 .. b_nb]   reads C, Y_cur, writes C, Y_hat once
//
// (same mathematics as psmf_blk_xgram_mfma / psmf_blk_apply_mfma in psmf_block.hip, which remain the path for
// float64 storage, d_local not a multiple of 4 and r < 16).  What changed is how the bytes move:
//   * every WAVE streams its own 16-row tiles: wave-private LDS image, no workgroup barrier in the loop;
//   * 16-byte global loads laid out so that each instruction covers whole 64-byte pieces of 16 series rows
//     (Y is time-major: 16 consecutive rows of one time step are contiguous) or a contiguous 2 KiB of C;
//   * the next tile is prefetched into registers while the f64 MFMAs of the current one run;
//   * results leave through LDS so that C and y_hat are written as 16-byte pieces too.
// On gfx950 v_mfma_f64_16x16x4_f64 takes 64 cycles per SIMD: at d = 1e5, r = 32 the cross-Gram is 0.61 GFLOP
// (7.8 us at the 78.6 TFLOP/s float64 peak) against 38.4 MB (4.8 us at 8 TB/s): matrix-core bound, as is the apply
// (0.82 GFLOP, 10.4 us; 51.2 MB, 6.4 us).
#pragma once
#include "psmf_block.hip"

namespace psmf {

// per-wave time stamps for tools/bulk_prof.hip (BK_STAMPS); no-ops in the product
#ifdef BK_STAMPS
// slot i: s_memrealtime (100 MHz), slot 4 + i: s_memtime (shader clock) -- their ratio is the clock the CU really ran at
#define BK_STAMP(i) do { if ((threadIdx.x & 63) == 0) { long long* s_ = reinterpret_cast<long long*>(b.Kpart) + (blockIdx.x * BK_WAVES + (threadIdx.x >> 6)) * 8; s_[i] = (long long)__builtin_amdgcn_s_memrealtime(); s_[4 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define BK_STAMP(i) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK_TR = 16;      // rows per wave tile
constexpr int BK_FS = 20;      // column stride (floats) of a wave's tile image [128 columns][16 rows + 4]: a lane's four rows
                               // of a column are one aligned 16-byte store, operand reads are 2-way bank conflicts at worst
constexpr int BK_FIMG = 128 * BK_FS;   // floats per wave image (10 KB)
constexpr int BK_WAVES = 8;
constexpr int BK_NT = 64 * BK_WAVES;
constexpr int BK_XG_WG = 256;  // workgroups (= partials) of the cross-Gram

// the wave images (float32: the tile is stored as loaded and converted where the MFMA operands are read -- the store phase
// of a tile is 12 LDS writes instead of 24 conversions + 24 writes, which is what the second wave of a SIMD could not hide)
// or the two buffers of the tree reduction at the end (7 x 3 tiles of 256 doubles each), whichever is larger
inline size_t blk_xgram2_lds_bytes() {
  const size_t img = (size_t)BK_WAVES * BK_FIMG * 4, red = (size_t)2 * 7 * 3 * 256 * 8;
  return img > red ? img : red;
}

// 16-byte load of 4 consecutive rows of one series column / 4 consecutive elements of C; the guarded form serves the
// last (partial) tile
__device__ __forceinline__ f32x4 bk_load4(const float* __restrict__ p, const bool full, const int n_ok) {
  if (full) return *reinterpret_cast<const f32x4*>(p);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n_ok > 0) v[0] = p[0];
  if (n_ok > 1) v[1] = p[1];
  if (n_ok > 2) v[2] = p[2];
  if (n_ok > 3) v[3] = p[3];
  return v;
}

// ------------------------------------------------------------------------------------------------------------
// XG partial of one workgroup.  NCT = 16-column tiles of Y_next (2: r = 32; 3: 16 <= r < 32).
// Image columns: [0, r) C, [r, r + nb) Y_cur, zeros to 64, [64, 64 + nb1) Y_next, zeros to 128.
// Output tiles (rt, ct): rt 0..3 = rows of Z, rt 4.. = rows of Y_next; partial layout [rt * NCT + ct][q][lane].
// ------------------------------------------------------------------------------------------------------------
template <int NCT>
__global__ __launch_bounds__(BK_NT) void psmf_blk_xgram2(BlockParams b) {
  constexpr int NRT = 4 + NCT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* smem = reinterpret_cast<double*>(smem_raw);
  const StepParams& p = b.sp;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lrow = lane >> 4, lcol = lane & 15;
  const int r = p.r, rp = p.rp, dl = p.d_local, nb = b.nb, nb1 = b.nb1;
  const float* __restrict__ C = reinterpret_cast<const float*>(p.C);
  const float* __restrict__ Y = reinterpret_cast<const float*>(p.Y) + (size_t)(b.k0 - p.series_t0) * dl;
  const float* __restrict__ Y1 = reinterpret_cast<const float*>(p.Y) + (size_t)(b.k1 - p.series_t0) * dl;
  float* img = reinterpret_cast<float*>(smem_raw) + (size_t)w * BK_FIMG;      // [column][row], column stride BK_FS
  BK_STAMP(0);
  f64x4 acc[NRT * NCT];
#pragma unroll
  for (int t = 0; t < NRT * NCT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  // C piece of this lane: vectors lane, lane + 64 of the tile's 16 * rp contiguous floats
  int crow[2], ccol[2];
  bool cok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 4 * (lane + 64 * i);
    cok[i] = e < BK_TR * rp;
    crow[i] = cok[i] ? e / rp : 0;
    ccol[i] = cok[i] ? e - crow[i] * rp : 0;
  }
  const int ntile = (dl + BK_TR - 1) / BK_TR;
  const int nwave = gridDim.x * BK_WAVES;
  f32x4 cv[2], yv[NCT], nv[NCT];
  auto load_tile = [&](const int t) {
    const int row0 = t * BK_TR;
    const bool full = row0 + BK_TR <= dl;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = min(row0 + crow[i], dl - 1);
      cv[i] = (cok[i] && row0 + crow[i] < dl) ? *reinterpret_cast<const f32x4*>(C + (size_t)rr * rp + ccol[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int rs = row0 + 4 * lrow;                 // first of this lane's 4 rows
    const int n_ok = dl - rs;
#pragma unroll
    for (int i = 0; i < NCT; ++i) {
      const int q = lcol + 16 * i;
      yv[i] = (q < nb) ? bk_load4(Y + (size_t)q * dl + min(rs, dl - 1), full, n_ok) : f32x4{0.f, 0.f, 0.f, 0.f};
      nv[i] = (q < nb1) ? bk_load4(Y1 + (size_t)q * dl + min(rs, dl - 1), full, n_ok) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // tile k of this wave: whole rounds over all waves, then the remainder spread over the workgroups, one wave (= one
  // SIMD) at a time: the two waves that share a SIMD never both get an extra tile (the loop is matrix-core bound)
  const int gw = blockIdx.x * BK_WAVES + w, nfull = ntile / nwave, rem = ntile - nfull * nwave;
  auto tile_of = [&](const int k) -> int {
    if (k < nfull) return k * nwave + gw;
    const int j = w * (int)gridDim.x + (int)blockIdx.x;
    return (k == nfull && j < rem) ? nfull * nwave + j : ntile;
  };
  int kt = 0;
  int t = tile_of(0);
  if (t < ntile) load_tile(t);
  for (int i = lane; i < BK_FIMG; i += 64) img[i] = 0.f;       // (behind the first tile's loads)
  BK_STAMP(1);
  while (t < ntile) {
    // registers -> image, as loaded (float32)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (cok[i]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ccol[i] + j < r) img[(ccol[i] + j) * BK_FS + crow[i]] = cv[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < NCT; ++i) {
      const int q = lcol + 16 * i;
      if (q < nb) *reinterpret_cast<f32x4*>(img + (r + q) * BK_FS + 4 * lrow) = yv[i];
      if (q < nb1) *reinterpret_cast<f32x4*>(img + (64 + q) * BK_FS + 4 * lrow) = nv[i];
    }
    const int tn = tile_of(++kt);
    if (tn < ntile) load_tile(tn);                 // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK_TR / 4; ++kk) {
      const float* rowp = img + lcol * BK_FS + 4 * kk + lrow;
      double av[NRT], bv[NCT];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) av[rt] = (double)rowp[16 * rt * BK_FS];          // rt >= 4: columns 64 + 16 (rt - 4) = 16 rt
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) bv[ct] = (double)rowp[(64 + 16 * ct) * BK_FS];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
          acc[rt * NCT + ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt], bv[ct], acc[rt * NCT + ct], 0, 0, 0);
    }
    t = tn;
  }
  BK_STAMP(2);
  // ---- fixed-order tree over the 8 waves (two buffers at a time), then one partial per workgroup ----
  constexpr int PSZ = NRT * NCT * 256;          // doubles per partial
  __syncthreads();
  for (int s = BK_WAVES / 2; s >= 1; s >>= 1) {
    for (int c0 = 0; c0 < s; c0 += 2) {
      // senders s + c0, s + c0 + 1 -> buffers 0, 1; receivers c0, c0 + 1
      if (w >= s + c0 && w < s + c0 + 2 && w < 2 * s) {
        double* buf = smem + (size_t)(w - s - c0) * PSZ;
#pragma unroll
        for (int tI = 0; tI < NRT * NCT; ++tI)
#pragma unroll
          for (int q = 0; q < 4; ++q) buf[(tI * 4 + q) * 64 + lane] = acc[tI][q];
      }
      __syncthreads();
      if (w >= c0 && w < c0 + 2 && w < s) {
        const double* buf = smem + (size_t)(w - c0) * PSZ;
#pragma unroll
        for (int tI = 0; tI < NRT * NCT; ++tI)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[tI][q] += buf[(tI * 4 + q) * 64 + lane];
      }
      __syncthreads();
    }
  }
  if (w == 0) {
    double* out = b.XGpart + (size_t)blockIdx.x * PSZ;
#pragma unroll
    for (int tI = 0; tI < NRT * NCT; ++tI)
#pragma unroll
      for (int q = 0; q < 4; ++q) out[(tI * 4 + q) * 64 + lane] = acc[tI][q];
  }
  BK_STAMP(3);
}

// XG[(rowbase + (l >> 4) + 4 q) * XGB + 16 ct + (l & 15)] = sum over partials, fixed order.
// 256 threads = 32 raw elements x 8 partial groups; every thread keeps nparts / 8 loads in flight.
template <int NCT>
__global__ __launch_bounds__(256) void psmf_blk_xreduce2(const double* __restrict__ part, double* __restrict__ XGout, int nparts) {
  constexpr int NRT = 4 + NCT;
  constexpr int PSZ = NRT * NCT * 256;
  __shared__ double red[8][33];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;                  // raw element: (tile * 4 + q) * 64 + lane
  // every thread sums nparts / 8 partials (g, g + 8, ...), 8 loads in flight at a time; fixed order
  double a0 = 0.0, a1 = 0.0;
  for (int p0 = g; p0 < nparts; p0 += 64) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)min(p0 + 8 * u, nparts - 1) * PSZ + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (p0 + 8 * u < nparts) ? v[u] : 0.0;
    a0 += (v[0] + v[2]) + (v[4] + v[6]);
    a1 += (v[1] + v[3]) + (v[5] + v[7]);
  }
  red[g][el] = a0 + a1;
  __syncthreads();
  if (g == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][el];
    const int l = e & 63, q = (e >> 6) & 3, tI = e >> 8;
    const int rt = tI / NCT, ct = tI - rt * NCT;
    const int rowbase = rt < 4 ? 16 * rt : RB + 16 * (rt - 4);
    XGout[(size_t)(rowbase + (l >> 4) + 4 * q) * XGB + 16 * ct + (l & 15)] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------
// [C_new | Y_hat] = Z [A | b_1 .. b_nb]:  one wave per 16-row slab (grid-stride), coefficient matrix in LDS for the
// whole workgroup (row stride GZ_S), slab image and output staging wave-private.
// ------------------------------------------------------------------------------------------------------------
constexpr int AP2_SC = 36;     // staging row stride (floats) of the C part  [16 rows][rp <= 32]
constexpr int AP2_SY = 20;     // staging stride (floats) of the series part  [column][16 rows]
constexpr int AP2_NY = 48;     // series columns staged = the longest block (psmf_capi.hip caps blocks at 48 timesteps)
constexpr int AP2_FZ = 64 * BK_FS;      // floats of a wave's slab image [64 coefficient columns][16 rows + 4], float32 as loaded
constexpr size_t AP2_WAVE_BYTES = (size_t)AP2_FZ * 4 + BK_TR * AP2_SC * 4 + AP2_NY * AP2_SY * 4;
inline size_t blk_apply2_lds_bytes() { return (size_t)RB * GZ_S * 8 + (size_t)BK_WAVES * AP2_WAVE_BYTES; }

template <int NYC>       // 16-column pieces of the series block: ceil((64 - r) / 16)
__global__ __launch_bounds__(BK_NT) void psmf_blk_apply2(BlockParams b) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sW = reinterpret_cast<double*>(smem_raw);                      // RB x GZ_S: rows = coefficient index, cols [0,r) A, [r, r+nb) b_j
  const StepParams& p = b.sp;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lrow = lane >> 4, lcol = lane & 15;
  const int r = p.r, rp = p.rp, dl = p.d_local, nb = b.nb;
  char* wbase = smem_raw + (size_t)RB * GZ_S * 8 + (size_t)w * AP2_WAVE_BYTES;
  float* sZ = reinterpret_cast<float*>(wbase);                           // [64 columns of Z][BK_FS]: column-major, float32
  float* sC = reinterpret_cast<float*>(wbase + AP2_FZ * 4);              // 16 x AP2_SC
  float* sY = sC + BK_TR * AP2_SC;                                       // AP2_NY x AP2_SY  (index = output column - r)
  BK_STAMP(0);
  float* __restrict__ C = reinterpret_cast<float*>(p.C);
  const float* __restrict__ Y = reinterpret_cast<const float*>(p.Y) + (size_t)(b.k0 - p.series_t0) * dl;
  float* __restrict__ YP = p.store_yp ? reinterpret_cast<float*>(p.YP) + (size_t)(b.k0 - p.series_t0) * dl : nullptr;
  int crow[2], ccol[2];
  bool cok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 4 * (lane + 64 * i);
    cok[i] = e < BK_TR * rp;
    crow[i] = cok[i] ? e / rp : 0;
    ccol[i] = cok[i] ? e - crow[i] * rp : 0;
  }
  const int nslab = (dl + BK_TR - 1) / BK_TR;
  const int nwave = gridDim.x * BK_WAVES;
  f32x4 cv[2], yv[NYC];
  auto load_slab = [&](const int t) {
    const int row0 = t * BK_TR;
    const bool full = row0 + BK_TR <= dl;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = min(row0 + crow[i], dl - 1);
      cv[i] = (cok[i] && row0 + crow[i] < dl) ? *reinterpret_cast<const f32x4*>(C + (size_t)rr * rp + ccol[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int rs = row0 + 4 * lrow;
#pragma unroll
    for (int i = 0; i < NYC; ++i) {
      const int q = lcol + 16 * i;
      yv[i] = (q < nb) ? bk_load4(Y + (size_t)q * dl + min(rs, dl - 1), full, dl - rs) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  const int gw = blockIdx.x * BK_WAVES + w, nfull = nslab / nwave, rem = nslab - nfull * nwave;
  auto slab_of = [&](const int k) -> int {      // as tile_of in psmf_blk_xgram2
    if (k < nfull) return k * nwave + gw;
    const int j = w * (int)gridDim.x + (int)blockIdx.x;
    return (k == nfull && j < rem) ? nfull * nwave + j : nslab;
  };
  int kt = 0;
  int t = slab_of(0);
  if (t < nslab) load_slab(t);
  // the coefficient matrix and the LDS images, behind the first slab's loads
  {
    // all eight loads of a thread in flight at once: unconditional, clamped addresses, selected afterwards (under a runtime
    // predicate each load is branched around and waited for on its own: eight dependent L2 round trips, 3 us of prologue)
    constexpr int NL = RB * RB / BK_NT;
    double va[NL], vb[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int idx = tid + u * BK_NT, m = idx / RB, c = idx - m * RB;
      va[u] = b.Acoef[m * r + min(c, r - 1)];
      vb[u] = b.Bcoef[(size_t)min(max(c - r, 0), RB - 1) * RB + m];
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int idx = tid + u * BK_NT, m = idx / RB, c = idx - m * RB;
      sW[m * GZ_S + c] = c < r ? va[u] : (c < r + nb ? vb[u] : 0.0);
    }
  }
  for (int i = lane; i < AP2_FZ; i += 64) sZ[i] = 0.f;
  for (int i = lane; i < BK_TR * AP2_SC; i += 64) sC[i] = 0.f;
  __syncthreads();
  // the B operands (the block's coefficient matrix) are the same for every slab: 64 doubles per lane, kept in registers
  // (they were 64 of the 80 LDS reads of a slab, issued in two bursts that the MFMAs had to wait for)
  double bw[2][4][8];
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int kq = 0; kq < 8; ++kq)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bw[half][ct][kq] = sW[(32 * half + 4 * kq + lrow) * GZ_S + 16 * ct + lcol];
  BK_STAMP(1);
  while (t < nslab) {
    const int row0 = t * BK_TR;
    const bool full = row0 + BK_TR <= dl;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (cok[i]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ccol[i] + j < r) sZ[(ccol[i] + j) * BK_FS + crow[i]] = cv[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < NYC; ++i) {
      const int q = lcol + 16 * i;
      if (q < nb) *reinterpret_cast<f32x4*>(sZ + (r + q) * BK_FS + 4 * lrow) = yv[i];
    }
    const int tn = slab_of(++kt);
    if (tn < nslab) load_slab(tn);                 // in flight during the MFMAs below
    f64x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double av[8];
#pragma unroll
      for (int kq = 0; kq < 8; ++kq) av[kq] = (double)sZ[(32 * half + 4 * kq + lrow) * BK_FS + lcol];
#pragma unroll
      for (int kq = 0; kq < 8; ++kq)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kq], bw[half][ct][kq], acc[ct], 0, 0, 0);
    }
    // results -> staging (one rounding to float32 per block), then 16-byte stores
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int col = 16 * ct + lcol;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = lrow + 4 * q;
        if (col < r) sC[row * AP2_SC + col] = (float)acc[ct][q];
        else if (col - r < AP2_NY) sY[(col - r) * AP2_SY + row] = (float)acc[ct][q];     // (r < 16: columns r + 48 .. 63 are padding)
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (cok[i] && row0 + crow[i] < dl) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sC + crow[i] * AP2_SC + ccol[i]);
        *reinterpret_cast<f32x4*>(C + (size_t)(row0 + crow[i]) * rp + ccol[i]) = v;
      }
    }
    if (YP) {
      const int rs = row0 + 4 * lrow;
#pragma unroll
      for (int i = 0; i < NYC; ++i) {
        const int q = lcol + 16 * i;
        if (q < nb) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sY + q * AP2_SY + 4 * lrow);
          float* dst = YP + (size_t)q * dl + rs;
          if (full) *reinterpret_cast<f32x4*>(dst) = v;
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (rs + j < dl) dst[j] = v[j];
          }
        }
      }
    }
    t = tn;
  }
  BK_STAMP(2);
  BK_STAMP(3);
}

}  // namespace psmf
