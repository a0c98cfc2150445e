// Masked timesteps on the large-d per-step engine (cfg.masked = 1): the filter of ExperimentImpute/PSMF.py:59-84 and
// ExperimentImpute/rPSMF.py:75-135 at ANY d and r <= PSMF_RMAX, on row shards.  SURVEY App. A with a 0/1 observation mask m:
//
//   e_i   = m_i (y_i - c_i . mu_bar)             rows with m_i = 0 are neither used nor updated; y_hat is stored unmasked
//   G_m   = sum_i m_i c_i c_i^T                  the mask changes with every step, so neither the algebraically tracked Gram
//                                                nor time-blocking apply: one masked Gram pass per step (mgram_body)
//   eta   = (rho n_obs + <G_m, P_bar>) / d       divided by d, NOT by the observed count (PSMF.py:77)
//   kappa_i = m_i / (rho + s)                    =>  P+ = (P_bar^-1 + kappa G_m)^-1,  b = kappa h,  q = kappa ee  (uniform rho)
//   lambda <- lambda + d                         (d again, rPSMF.py:135)
//   bands : PSMF  y_hat -+ sig sqrt(N)  (PSMF.py:83-84);  rPSMF  y_hat_i -+ sig sqrt(s m_i + eta)  (rPSMF.py:112,121-123)
// and, sharing these contractions (StepParams.masked_method), the two baseline filters of the imputation tables: MLE-SMF (weights
// m_i / rho, C += gam / eta (m o e) x_p^T, bands -+ sig sqrt(eta): MLESMF.py:57-88) and TMF (Pbar = I / nu, kappa = 1, C += gam (m o e) x_p^T:
// TMF.py:47-66).
//
// A step is: psmf_sweep_solve with the mask -- every workgroup first forms eta, N, the update direction from the step's reduced Gram
// (masked_prep_block, psmf_kernels.hip; block 0 publishes them and the step's (s, eta) for the bands) -- (-> all-reduce of r + 1
// doubles) -> psmf_serial_mgram: the serial stage of the step in block 0 and, beside it, the masked Gram of the NEXT step in the other
// blocks -> psmf_mgram_reduce (-> all-reduce of r^2 + 1 doubles).  The metrics of a pass (RMSE of the predictions and of C X over the held-out entries, coverage of the bands:
// ExperimentImpute/common.py:79-94) are reduced on the device by psmf_masked_metrics_k; nothing d x n travels.
#pragma once
#include "psmf_kernels.hip"
#include "psmf_blk3.hip"      // f64x4, readlane_f64

namespace psmf {

// Masked Gram of the step st->k on the float64 matrix cores: gpart[wg][0 .. r*r) = sum m_i c_i c_i^T over the workgroup's rows,
// gpart[wg][r*r] = sum m_i.  One WAVE per slab of 16 rows (wave-private LDS image, float64, rows with m_i = 0 stored as zeros, so
// that one image serves both operands: m^2 = m); the next slab is already in registers while the current one is multiplied
// (16 row loads in flight per lane); per slab and 16 x 16 output tile four v_mfma_f64_16x16x4_f64 (K = 4 rows each), upper
// triangle of tiles only.  NT = column tiles (r <= 16 NT), NW = waves per workgroup.  The waves' accumulators are summed in a
// fixed order through LDS: one partial per workgroup, reduced in fixed order by psmf_mgram_reduce -> deterministic.
// LDS row stride S == 16 (mod 32) doubles: the two rows a half-wave reads sit 32 banks apart (conflict-free ds_read_b64).
__host__ __device__ constexpr int mgram_stride(int nt) { return nt == 1 ? 16 : (nt <= 3 ? 48 : 80); }

__host__ __device__ constexpr int mgram_lds_doubles(int nt, int nw) {
  return (nw * 16 * mgram_stride(nt) > nt * nt * 256 ? nw * 16 * mgram_stride(nt) : nt * nt * 256) + nw;
}

// sZ: mgram_lds_doubles(NT, NW) doubles of LDS (dynamic: the serial stage of the same kernel has its own static arrays)
// MODE 1 (round 5): the WEIGHTED Gram of a non-uniform diagonal R, sum_i kappa_i c_i c_i^T with kappa_i = 1 / (rho rho_i + s) of the
// current step (psmf.py:140-152 with a diagonal R) -- the same slabs, every row scaled by sqrt(kappa_i) as it is
// stored, so that one image still serves both operands; partial rows of r * r doubles (no count).
template <typename T, int NT, int NW, int MODE = 0>
__device__ __forceinline__ void mgram_body(const StepParams& p, double* __restrict__ gpart, const int bid, const int nblk, double* sZ) {
  constexpr int S = mgram_stride(NT);
  constexpr int NTT = NT * (NT + 1) / 2;
  double* sCnt = sZ + (mgram_lds_doubles(NT, NW) - NW);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int r = p.r, rp = p.rp, dl = p.d_local;
  const T* __restrict__ C = reinterpret_cast<const T*>(p.C);
  // the step whose Gram this is: st->kq (written by the previous kernel; the serial stage running beside these blocks advances
  // st->k, not kq); clamped: the Gram "of the step after the last" is computed and never used
  long long trow = p.st->kq - p.series_t0;
  if (trow > (long long)p.mask_rows - 1) trow = (long long)p.mask_rows - 1;
  const uint8_t* __restrict__ mk = MODE == 0 ? p.mask + (size_t)trow * dl : nullptr;
  const double* __restrict__ rrow = p.rho_rows;
  const double rsc = p.st->rho, sk = p.st->s;
  double* slab = sZ + w * 16 * S;
  for (int idx = lane; idx < 16 * S; idx += 64) slab[idx] = 0.0;        // columns rp .. 16 NT stay zero
  f64x4 acc[NTT];
#pragma unroll
  for (int t = 0; t < NTT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  double cnt = 0.0;
  const int nslab = (dl + 15) / 16, stride = nblk * NW;
  // a slab = 16 rows x rp elements, contiguous: 16-byte vectors lane, lane + 64, ... of it (rp is a multiple of the vector length)
  constexpr int VEC = 16 / sizeof(T);
  constexpr int NV = (16 * 16 * NT / VEC + 63) / 64;         // vectors per lane at most (rp <= 16 NT)
  typedef typename VecOf<T>::type VT;
  int vrow[NV], vcol[NV];
  bool von[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = VEC * (lane + 64 * i);
    von[i] = e < 16 * rp;
    vrow[i] = von[i] ? e / rp : 0;
    vcol[i] = von[i] ? e - vrow[i] * rp : 0;
  }
  int sl = bid * NW + w;
  VT v[NV];
  uint8_t m[NV];
  double wr[NV];
  auto load_slab = [&](const int s_) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = min(s_ * 16 + vrow[i], dl - 1);
      v[i] = *reinterpret_cast<const VT*>(C + (size_t)row * rp + vcol[i]);
      if (MODE == 0) m[i] = mk[row];
      else wr[i] = rrow[row];
    }
  };
  if (sl < nslab) load_slab(sl);
  for (; sl < nslab; sl += stride) {
    const int row0 = sl * 16;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool on = von[i] && (row0 + vrow[i] < dl) && (MODE == 1 || m[i] != 0);
      const double sw = MODE == 1 ? rsqrt(rsc * wr[i] + sk) : 1.0;
      if (von[i]) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) slab[vrow[i] * S + vcol[i] + j] = on ? (MODE == 1 ? (double)v[i][j] * sw : (double)v[i][j]) : 0.0;
      }
      cnt += (on && vcol[i] == 0) ? 1.0 : 0.0;          // the lane that holds a row's first vector counts the row
    }
    const int nx = sl + stride;
    if (nx < nslab) load_slab(nx);          // the next slab's loads fly during this slab's products
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS stores (global loads stay in flight)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) a[t] = slab[(4 * q + lk) * S + 16 * t + lr];
      int tt = 0;
#pragma unroll
      for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int tb = ta; tb < NT; ++tb, ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], a[tb], acc[tt], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();        // all lanes have read the image before it is overwritten
  }
  // ---- sum of the waves' tiles by a fixed tree (waves [h, 2h) into [0, h), h = NW / 2, NW / 4, .. 1), then the workgroup's partial
  __syncthreads();
  double* buf = sZ;                           // slot x [upper tile][q][lane]
  for (int hlf = NW / 2; hlf >= 1; hlf >>= 1) {
    if (w >= hlf && w < 2 * hlf) {
      double* o = buf + (size_t)(w - hlf) * NTT * 256;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) o[(tt * 4 + q) * 64 + lane] = acc[tt][q];
    }
    __syncthreads();
    if (w < hlf) {
      const double* o = buf + (size_t)w * NTT * 256;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[tt][q] += o[(tt * 4 + q) * 64 + lane];
    }
    __syncthreads();
  }
  if (w == 0) {                               // full tile grid [ta * NT + tb] (upper tiles) for the write-out below
    int tt = 0;
#pragma unroll
    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
      for (int tb = ta; tb < NT; ++tb, ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) buf[((ta * NT + tb) * 4 + q) * 64 + lane] = acc[tt][q];
  }
  cnt = wave_sum(cnt);
  if (lane == 0) sCnt[w] = cnt;
  __syncthreads();
  const size_t gs = (size_t)r * r + (MODE == 0 ? 1 : 0);
  double* out = gpart + (size_t)bid * gs;
  for (int idx = tid; idx < r * r; idx += NW * 64) {
    int i = idx / r, j = idx - i * r;
    if ((i >> 4) > (j >> 4)) { const int t_ = i; i = j; j = t_; }      // lower tiles: the transposed entry of the upper one
    const int ii = i & 15, jj = j & 15;
    out[idx] = buf[(((i >> 4) * NT + (j >> 4)) * 4 + (ii >> 2)) * 64 + (ii & 3) * 16 + jj];
  }
  if (MODE == 0 && tid == 0) {
    double c = 0.0;
    for (int ww = 0; ww < NW; ++ww) c += sCnt[ww];
    out[(size_t)r * r] = c;
  }
}

// the weighted Gram of the current step on its own (non-uniform diagonal R, per-step engine): gpart[blockIdx.x][r * r], reduced by
// psmf_mgram_reduce into st->GR.  (psmf_gram_partial, the vector-unit Gram of psmf_set_state, took 170 us per timestep at d = 1e5, r = 32.)
template <typename T, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void psmf_wgram_mfma(StepParams p, double* __restrict__ gpart) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  mgram_body<T, NT, NW, 1>(p, gpart, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<double*>(smem_raw));
}

// One launch for the serial stage of step k (block 0: psmf_serial's body, psmf_kernels.hip) AND the masked Gram of step k + 1 (blocks
// 1 .. : mgram_body): both need only what the sweep of step k left behind -- the partial sums and P+ the one, the updated C the
// other -- so they run side by side instead of one after the other (the serial stage is ~9 us of one workgroup's latency, the Gram
// ~12 us of matrix-core work on every CU).  RPAD / NT / NW: serial_threads(RPAD) == 64 NW.  `first`: psmf_serial's flag (start of a run).
template <int RPAD, typename T, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void psmf_serial_mgram(StepParams p, int first, double* __restrict__ gpart) {
  static_assert(serial_threads(RPAD) == NW * 64, "block 0 runs the serial stage: same workgroup size");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (blockIdx.x == 0) serial_body<RPAD>(p, first);
  else mgram_body<T, NT, NW>(p, gpart, (int)blockIdx.x - 1, (int)gridDim.x - 1, reinterpret_cast<double*>(smem_raw));
}

// gpart (n_part rows of ne doubles) -> out[ne], every entry summed in the fixed order of the partials: 512 threads = 64 entries x 8
// segments of the partial rows (all loads of a thread independent), then the 8 segment sums in order.  Pbar != nullptr (ne = r*r + 1):
// also this workgroup's share of <G_m, Pbar> -> tpart[blockIdx.x] (the sweep sums the shares in order: eta without a pass over G).
__global__ __launch_bounds__(512) void psmf_mgram_reduce(const double* __restrict__ gpart, int n_part, int ne, double* __restrict__ out,
                                                         const double* __restrict__ Pbar, int r, double* __restrict__ tpart) {
  __shared__ double s8[8][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
  const int per = (n_part + 7) / 8, a0 = sg * per, a1 = min(a0 + per, n_part);
  double a = 0.0;
  const int ec = min(e, ne - 1);
  for (int w0 = a0; w0 < a1; w0 += 8) {        // 8 independent loads in flight (clamped index, masked value), summed in order
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = gpart[(size_t)min(w0 + q, a1 - 1) * ne + ec];
#pragma unroll
    for (int q = 0; q < 8; ++q) a += (w0 + q < a1) ? v[q] : 0.0;
  }
  s8[sg][threadIdx.x & 63] = a;
  __syncthreads();
  if (sg == 0) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += s8[q][threadIdx.x];
    if (e < ne) out[e] = t;
    if (Pbar) {
      double tp = 0.0;
      if (e < r * r) { const int i = e / r, j = e - i * r; tp = t * (0.5 * (Pbar[e] + Pbar[j * r + i])); }
      tp = wave_sum(tp);
      if (threadIdx.x == 0) tpart[blockIdx.x] = tp;
    }
  }
}

// the same shares from an already reduced Gram (row-sharded filter: the all-reduce over the ranks sits between the reduction and this)
__global__ __launch_bounds__(64) void psmf_mgram_trace(const double* __restrict__ mg, const double* __restrict__ Pbar, int r, double* __restrict__ tpart) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  double tp = 0.0;
  if (e < r * r) { const int i = e / r, j = e - i * r; tp = mg[e] * (0.5 * (Pbar[e] + Pbar[j * r + i])); }
  tp = wave_sum(tp);
  if (threadIdx.x == 0) tpart[blockIdx.x] = tp;
}

// Metrics of a pass over the held-out entries (Mmiss = 1) of this handle's rows, steps t0 .. t0 + nt:
//   part[blk][0] = sum (y_hat - y)^2        (Epred^2 * count, PSMF.py:88)       y_hat = the stored (unmasked) predictions
//   part[blk][1] = sum (c_i . x_t - y)^2    (Efull^2 * count, PSMF.py:86-89)    final C of the pass, x_t = mu_hist[t + 1]
//   part[blk][2] = number of entries strictly inside their band (common.py:87-94)
//   part[blk][3] = number of held-out entries
// grid = (row blocks, time chunks); a thread owns one row (its C row in registers) and walks its chunk of steps.
template <typename T>
__global__ __launch_bounds__(WG) void psmf_masked_metrics_k(StepParams p, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ mmiss,
                                                            const double* __restrict__ sc_hist, long long t0, int nt, int chunk, double sig,
                                                            int robust, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_x = reinterpret_cast<double*>(smem_raw);          // TC x r rows of the mean history
  __shared__ double s_red[4][4];
  const int tid = threadIdx.x, r = p.r, rp = p.rp, d_local = p.d_local;
  const int row = blockIdx.x * WG + tid;
  const bool on = row < d_local;
  const T* __restrict__ C = reinterpret_cast<const T*>(p.C);
  const T* __restrict__ Y = reinterpret_cast<const T*>(p.Y);
  const T* __restrict__ YP = reinterpret_cast<const T*>(p.YP);
  double c[RM];
  for (int l = 0; l < r; ++l) c[l] = on ? (double)C[(size_t)row * rp + l] : 0.0;
  double a_pred = 0.0, a_full = 0.0, a_in = 0.0, a_cnt = 0.0;
  const int tb = blockIdx.y * chunk, te = min(tb + chunk, nt);
  constexpr int TC = 32;
  for (int q0 = tb; q0 < te; q0 += TC) {
    const int nq = min(TC, te - q0);
    __syncthreads();
    for (int idx = tid; idx < nq * r; idx += WG) s_x[idx] = p.mu_hist[(size_t)(t0 + q0 + 1 - p.series_t0) * r + idx];   // row t + 1 = x_t
    __syncthreads();
    if (on) {
      for (int q = 0; q < nq; ++q) {
        const size_t t = (size_t)(t0 + q0 + q - p.series_t0);
        const size_t at = t * d_local + row;
        if (mmiss[(size_t)(q0 + q) * d_local + row]) {
          const double y = (double)Y[at], yh = (double)YP[at];
          double dot = 0.0;
          for (int l = 0; l < r; ++l) dot += c[l] * s_x[q * r + l];
          const double s = sc_hist[2 * t], eta = sc_hist[2 * t + 1];
          const double band = sig * sqrt(robust ? (mask[at] ? s : 0.0) + eta : s + eta);
          a_pred += (yh - y) * (yh - y);
          a_full += (dot - y) * (dot - y);
          a_in += (y < yh + band && yh - band < y) ? 1.0 : 0.0;
          a_cnt += 1.0;
        }
      }
    }
  }
  a_pred = wave_sum(a_pred); a_full = wave_sum(a_full); a_in = wave_sum(a_in); a_cnt = wave_sum(a_cnt);
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = a_pred; s_red[tid >> 6][1] = a_full; s_red[tid >> 6][2] = a_in; s_red[tid >> 6][3] = a_cnt; }
  __syncthreads();
  if (tid < 4) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
}

}  // namespace psmf
