// Masked timesteps on the large-d per-step engine (cfg.masked = 1): the filter of ExperimentImpute/PSMF.py:59-84 and
// ExperimentImpute/rPSMF.py:75-135 at ANY d and r <= PSMF_RMAX, on row shards.  SURVEY App. A with a 0/1 observation mask m:
//
//   e_i   = m_i (y_i - c_i . mu_bar)             rows with m_i = 0 are neither used nor updated; y_hat is stored unmasked
//   G_m   = sum_i m_i c_i c_i^T                  the mask changes with every step, so neither the algebraically tracked Gram
//                                                nor time-blocking apply: one masked Gram pass per step (psmf_mgram_mfma)
//   eta   = (rho n_obs + <G_m, P_bar>) / d       divided by d, NOT by the observed count (PSMF.py:77)
//   kappa_i = m_i / (rho + s)                    =>  P+ = (P_bar^-1 + kappa G_m)^-1,  b = kappa h,  q = kappa ee  (uniform rho)
//   lambda <- lambda + d                         (d again, rPSMF.py:135)
//   bands : PSMF  y_hat -+ sig sqrt(N)  (PSMF.py:83-84);  rPSMF  y_hat_i -+ sig sqrt(s m_i + eta)  (rPSMF.py:112,121-123)
// and, sharing these contractions (StepParams.masked_method), the two baseline filters of the imputation tables: MLE-SMF (weights
// m_i / rho, C += gam / eta (m o e) x_p^T, bands -+ sig sqrt(eta): MLESMF.py:57-88) and TMF (Pbar = I / nu, kappa = 1, C += gam (m o e) x_p^T:
// TMF.py:47-66).
//
// A step is: psmf_mgram_mfma -> psmf_mgram_reduce (-> all-reduce of r^2 + 1 doubles) -> psmf_masked_prep (eta, N, w / N, kappa and
// the step's (s, eta) into the history the bands are formed from) -> psmf_sweep_solve with the mask (-> all-reduce of r + 1 doubles)
// -> psmf_serial.  The metrics of a pass (RMSE of the predictions and of C X over the held-out entries, coverage of the bands:
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

template <typename T, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void psmf_mgram_mfma(StepParams p, const uint8_t* __restrict__ mask, double* __restrict__ gpart) {
  constexpr int S = mgram_stride(NT);
  constexpr int NTT = NT * (NT + 1) / 2;
  __shared__ double sZ[NW * 16 * S > NT * NT * 256 ? NW * 16 * S : NT * NT * 256];
  __shared__ double sCnt[NW];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int r = p.r, rp = p.rp, dl = p.d_local;
  const T* __restrict__ C = reinterpret_cast<const T*>(p.C);
  const uint8_t* __restrict__ mk = mask + (size_t)(p.st->k - p.series_t0) * dl;
  double* slab = sZ + w * 16 * S;
  for (int idx = lane; idx < 16 * S; idx += 64) slab[idx] = 0.0;        // columns rp .. 16 NT stay zero
  f64x4 acc[NTT];
#pragma unroll
  for (int t = 0; t < NTT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  double cnt = 0.0;
  const int nslab = (dl + 15) / 16, stride = gridDim.x * NW;
  const int cl = lane < rp ? lane : 0;
  const bool con = lane < rp;
  int sl = blockIdx.x * NW + w;
  T v[16];
  uint8_t m[16];
  if (sl < nslab) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int row = min(sl * 16 + rr, dl - 1);
      v[rr] = C[(size_t)row * rp + cl];
      m[rr] = mk[row];
    }
  }
  for (; sl < nslab; sl += stride) {
    const int row0 = sl * 16;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const bool on = (row0 + rr < dl) && m[rr] != 0;
      if (con) slab[rr * S + lane] = on ? (double)v[rr] : 0.0;
      cnt += on ? 1.0 : 0.0;
    }
    const int nx = sl + stride;
    if (nx < nslab) {                       // the next slab's loads fly during this slab's products
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = min(nx * 16 + rr, dl - 1);
        v[rr] = C[(size_t)row * rp + cl];
        m[rr] = mk[row];
      }
    }
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
  // ---- sum of the waves' tiles in wave order (fixed), then the workgroup's partial: both triangles
  __syncthreads();
  double* buf = sZ;                           // [tile ta * NT + tb][q][lane]
  for (int ww = 0; ww < NW; ++ww) {
    if (w == ww) {
      int tt = 0;
#pragma unroll
      for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int tb = ta; tb < NT; ++tb, ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            double* o = buf + ((ta * NT + tb) * 4 + q) * 64 + lane;
            *o = (ww == 0 ? 0.0 : *o) + acc[tt][q];
          }
    }
    __syncthreads();
  }
  cnt = readlane_f64(cnt, 0);                 // every lane of a wave counted the same rows
  if (lane == 0) sCnt[w] = cnt;
  __syncthreads();
  const size_t gs = (size_t)r * r + 1;
  double* out = gpart + (size_t)blockIdx.x * gs;
  for (int idx = tid; idx < r * r; idx += NW * 64) {
    int i = idx / r, j = idx - i * r;
    if ((i >> 4) > (j >> 4)) { const int t_ = i; i = j; j = t_; }      // lower tiles: the transposed entry of the upper one
    const int ii = i & 15, jj = j & 15;
    out[idx] = buf[(((i >> 4) * NT + (j >> 4)) * 4 + (ii >> 2)) * 64 + (ii & 3) * 16 + jj];
  }
  if (tid == 0) {
    double c = 0.0;
    for (int ww = 0; ww < NW; ++ww) c += sCnt[ww];
    out[(size_t)r * r] = c;
  }
}

// gpart (n_part rows of ne doubles) -> out[ne], every entry summed in the fixed order of the partials: 512 threads = 64 entries x 8
// segments of the partial rows (all loads of a thread independent), then the 8 segment sums in order.
__global__ __launch_bounds__(512) void psmf_mgram_reduce(const double* __restrict__ gpart, int n_part, int ne, double* __restrict__ out) {
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
  if (sg == 0 && e < ne) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += s8[q][threadIdx.x];
    out[e] = t;
  }
}

// One workgroup: mg[0 .. r*r) = G_m, mg[r*r] = n_obs (already summed over workgroups and ranks) -> st->G, eta, N, w / N, kappa of
// the step st->k, and (s, eta) of the step into sc_hist (bands of the metrics kernel).
__global__ __launch_bounds__(WG) void psmf_masked_prep(StepParams p, const double* __restrict__ mg, double* __restrict__ sc_hist) {
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x;
  __shared__ double s4[4];
  double gp = 0.0;
  for (int q = tid; q < r * r; q += WG) {
    const int i = q / r, j = q - i * r;
    const double g = 0.5 * (mg[q] + mg[j * r + i]);     // both triangles identical (the partial sums are symmetric up to the order of one product)
    st->G[q] = g;
    gp += g * 0.5 * (st->Pbar[q] + st->Pbar[j * r + i]);
  }
  gp = wave_sum(gp);
  if ((tid & 63) == 0) s4[tid >> 6] = gp;
  __syncthreads();
  const double tr = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  const int meth = p.masked_method;
  const double s = meth ? 0.0 : st->s, rho = st->rho;
  const double eta = (rho * mg[r * r] + tr) / (double)p.d;
  const double N = s + eta;
  // the direction of the rank-1 update of C: PSMF  w / N (w = V mu_bar);  MLE-SMF  (gam / eta) mu_bar;  TMF  gam mu_bar
  if (tid < r) st->wN[tid] = meth == 0 ? st->w[tid] * fast_rcp(N) : st->mu_bar[tid] * (meth == 2 ? st->sgd_gamma * fast_rcp(eta) : st->sgd_gamma);
  if (tid == 0) {
    st->eta = eta;
    st->N = N;
    st->kappa = meth == 3 ? 1.0 : fast_rcp(rho + s);
    if (sc_hist) {
      const long long t = st->k - p.series_t0;
      sc_hist[2 * t] = s;
      sc_hist[2 * t + 1] = eta;
    }
  }
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
