// Masked timesteps on the large-d per-step engine (cfg.masked = 1): the filter of ExperimentImpute/PSMF.py:59-84 and
// ExperimentImpute/rPSMF.py:75-135 at ANY d and r <= PSMF_RMAX, on row shards.  SURVEY App. A with a 0/1 observation mask m:
//
//   e_i   = m_i (y_i - c_i . mu_bar)             rows with m_i = 0 are neither used nor updated; y_hat is stored unmasked
//   G_m   = sum_i m_i c_i c_i^T                  the mask changes with every step, so neither the algebraically tracked Gram
//                                                nor time-blocking apply: one masked Gram pass per step (psmf_mgram_partial)
//   eta   = (rho n_obs + <G_m, P_bar>) / d       divided by d, NOT by the observed count (PSMF.py:77)
//   kappa_i = m_i / (rho + s)                    =>  P+ = (P_bar^-1 + kappa G_m)^-1,  b = kappa h,  q = kappa ee  (uniform rho)
//   lambda <- lambda + d                         (d again, rPSMF.py:135)
//   bands : PSMF  y_hat -+ sig sqrt(N)  (PSMF.py:83-84);  rPSMF  y_hat_i -+ sig sqrt(s m_i + eta)  (rPSMF.py:112,121-123)
//
// A step is: psmf_mgram_partial -> psmf_gram_reduce (-> all-reduce of r^2 + 1 doubles) -> psmf_masked_prep (eta, N, w / N, kappa and
// the step's (s, eta) into the history the bands are formed from) -> psmf_sweep_solve with the mask (-> all-reduce of r + 1 doubles)
// -> psmf_serial.  The metrics of a pass (RMSE of the predictions and of C X over the held-out entries, coverage of the bands:
// ExperimentImpute/common.py:79-94) are reduced on the device by psmf_masked_metrics_k; nothing d x n travels.
#pragma once
#include "psmf_kernels.hip"

namespace psmf {

// Masked Gram of this workgroup's rows for the step st->k: gpart[wg][0 .. r*r) = sum m_i c_i c_i^T, gpart[wg][r*r] = sum m_i.
template <typename T>
__global__ __launch_bounds__(WG) void psmf_mgram_partial(StepParams p, const uint8_t* __restrict__ mask, int rows_per_wg,
                                                         double* __restrict__ gpart) {
  constexpr int TR = 32;   // rows per LDS tile
  __shared__ double tile[TR][RM + 1];
  __shared__ double mrow[TR];
  __shared__ double s4[4];
  const int tid = threadIdx.x, r = p.r, rp = p.rp, d_local = p.d_local;
  const T* __restrict__ C = reinterpret_cast<const T*>(p.C);
  const uint8_t* __restrict__ mk = mask + (size_t)(p.st->k - p.series_t0) * d_local;
  constexpr int MU = (RM * RM) / WG;   // 16
  double acc[MU];
  int qa[MU], qb[MU];
#pragma unroll
  for (int u = 0; u < MU; ++u) {
    const int q = tid + u * WG;
    acc[u] = 0.0;
    qa[u] = q < r * r ? q / r : -1;
    qb[u] = q < r * r ? q - (q / r) * r : 0;
  }
  double cnt = 0.0;
  const int row_begin = blockIdx.x * rows_per_wg;
  const int row_end = min(row_begin + rows_per_wg, d_local);
  for (int base = row_begin; base < row_end; base += TR) {
    __syncthreads();
    if (tid < TR) {
      const int row = base + tid;
      const double m = (row < row_end && mk[row]) ? 1.0 : 0.0;
      mrow[tid] = m;
      cnt += m;
    }
    __syncthreads();
    for (int idx = tid; idx < TR * r; idx += WG) {
      const int rr = idx / r, c = idx - rr * r;
      const int row = base + rr;
      tile[rr][c] = (row < row_end && mrow[rr] != 0.0) ? (double)C[(size_t)row * rp + c] : 0.0;   // m^2 = m: one masked copy serves both factors
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MU; ++u) {
      if (qa[u] >= 0) {
        double a = acc[u];
        for (int rr = 0; rr < TR; ++rr) a += tile[rr][qa[u]] * tile[rr][qb[u]];
        acc[u] = a;
      }
    }
  }
  const size_t stride = (size_t)r * r + 1;
#pragma unroll
  for (int u = 0; u < MU; ++u)
    if (qa[u] >= 0) gpart[(size_t)blockIdx.x * stride + tid + u * WG] = acc[u];
  cnt = wave_sum(cnt);                         // only wave 0 holds counts (tid < TR)
  if (tid == 0) gpart[(size_t)blockIdx.x * stride + (size_t)r * r] = cnt;
  (void)s4;
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
  const double s = st->s, rho = st->rho;
  const double eta = (rho * mg[r * r] + tr) / (double)p.d;
  const double N = s + eta;
  if (tid < r) st->wN[tid] = st->w[tid] * fast_rcp(N);
  if (tid == 0) {
    st->eta = eta;
    st->N = N;
    st->kappa = fast_rcp(rho + s);
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
