// Exact time-blocked engine ("block engine") for unmasked data with uniform diagonal R.
//
// Within a block of nb <= B consecutive steps every innovation lies in span(Z),
// Z = [C_{k0} | y_{k0+1} .. y_{k0+nb}]  (d x RB, RB = r + B = 64), so ONE d-sized contraction per
// block, K = Z^T Z (float64), replaces the B row sweeps; the steps then run in coefficient space
// (C_j = Z A_j, e_j = Z a_j, h_j = A_{j-1}^T K a_j, ee_j = a_j^T K a_j) inside ONE workgroup with the
// whole state in LDS / registers -- no kernel boundary, no global round trip per step -- and
// C_{k0+nb} = Z A_nb, Y_hat = Z [b_1 .. b_nb] are two more d-sized products.  Exact (same recursion,
// float64); host model rpsmf_amd/blocked.py; C is rounded to the storage type once
// per block instead of once per step.
//
//   psmf_blk_gram<T>      K partials: row chunk staged in LDS as float64, 4 x 4 register tiles   (MFMA-free fp64)
//   psmf_blk_reduce       fixed-order sum of the partials
//   psmf_blk_filter<RPAD> the nb steps (one workgroup)
//   psmf_blk_apply<T>     C <- Z A_nb (rounded once), y_hat_j = Z b_j
#pragma once
#include "psmf_kernels.hip"

namespace psmf {

constexpr int RB = 64;           // r + B, padded coefficient dimension (r <= 32)
constexpr int BLK_GRAM_WG = 128; // workgroups (= partials) of the block Gram

struct BlockParams {
  StepParams sp;
  double* Kpart;      // BLK_GRAM_WG x RB*RB
  double* K;          // RB x RB
  double* Acoef;      // RB x r   (A_nb)
  double* Bcoef;      // RB x RB  (column j = b_j, stored [j][m])
  long long k0;       // first step of the block is k0 + 1 (0-based series row k0)
  int nb;             // steps in this block
  int gram_rows;      // rows per Gram workgroup
};

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(WG) void psmf_blk_gram(BlockParams b) {
  constexpr int TR = 32;
  __shared__ double sZ[TR][RB];
  const StepParams& p = b.sp;
  const int tid = threadIdx.x, ta = tid >> 4, tb = tid & 15;
  const int r = p.r, rp = p.rp, dl = p.d_local;
  const T* __restrict__ C = reinterpret_cast<const T*>(p.C);
  const T* __restrict__ Y = reinterpret_cast<const T*>(p.Y) + (size_t)(b.k0 - p.series_t0) * dl;
  double acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
  const int row_begin = blockIdx.x * b.gram_rows;
  const int row_end = min(row_begin + b.gram_rows, dl);
  for (int base = row_begin; base < row_end; base += TR) {
    __syncthreads();
    // stage TR rows of Z as float64: columns [0, r) from C, [r, r + nb) from the series block, 0 beyond
    for (int idx = tid; idx < TR * RB; idx += WG) {
      const int col = idx / TR, rr = idx - col * TR;       // consecutive threads -> consecutive rows (Y is time-major)
      const int row = min(base + rr, row_end - 1);
      double v;
      if (col < r) v = (double)C[(size_t)row * rp + col];
      else if (col < r + b.nb) v = (double)Y[(size_t)(col - r) * dl + row];
      else v = 0.0;
      sZ[rr][col] = (base + rr < row_end) ? v : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < TR; ++rr) {
      double za[4], zb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) { za[x] = sZ[rr][4 * ta + x]; zb[x] = sZ[rr][4 * tb + x]; }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] += za[x] * zb[y];
    }
  }
  double* out = b.Kpart + (size_t)blockIdx.x * RB * RB;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) out[(4 * ta + x) * RB + 4 * tb + y] = acc[x][y];
}

__global__ __launch_bounds__(128) void psmf_blk_reduce(BlockParams b) {
  const int e = blockIdx.x * 128 + threadIdx.x;   // RB*RB = 4096 = 32 x 128
  b.K[e] = strided_sum(b.Kpart + e, 0, 1, BLK_GRAM_WG, RB * RB);
}

// ------------------------------------------------------------------------------------------
// The nb steps of a block in coefficient space.  Thread mapping of the r x r state as in the
// serial stage (column j = tid % RPAD, rows ig + m * RG); RB x r coefficient matrices A, KA and
// the Gram K in LDS.
// ------------------------------------------------------------------------------------------
template <int RPAD>
__global__ __launch_bounds__(WG) void psmf_blk_filter(BlockParams b) {
  constexpr int RG = WG / RPAD;
  constexpr int M = (RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const StepParams& p = b.sp;
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x, j = tid % RPAD, ig = tid / RPAD;
  const int r2 = r + (r & 1);
  const double dd = (double)p.d;
  // ---- LDS carve ----
  double* sK = sm;                    // RB x RB
  double* sA = sK + RB * RB;          // RB x r
  double* sKA = sA + RB * RM / 2;     // RB x r   (r <= 32 = RM / 2)
  double* s_red = sKA + RB * RM / 2;  // WG
  double* s_mub = s_red + WG;         // RM each below
  double* s_f = s_mub + RM;
  double* s_w = s_f + RM;
  double* s_h = s_w + RM;
  double* s_vec = s_h + RM;
  double* s_mu = s_vec + RM;
  double* s_a = s_mu + RM;            // RB
  double* s_Ka = s_a + RB;            // RB
  double* s_p2 = s_Ka + RB;           // 4 x RB x 2 partial row dots
  double* rowbuf = s_p2 + 8 * RB;     // 4 * RM
  double* s4 = rowbuf + 4 * RM;       // 4 (+ errflag)
  int* errflag = reinterpret_cast<int*>(s4 + 4);

  for (int idx = tid; idx < RB * RB; idx += WG) sK[idx] = b.K[idx];
  if (tid == 0) *errflag = 0;
  if (tid < RM) { s_mub[tid] = 0.0; s_h[tid] = 0.0; s_w[tid] = 0.0; s_f[tid] = 1.0; }
  if (tid < r) s_mu[tid] = st->mu[tid];
  double Vv[M], Pv[M], Gv[M], Qv[M];
  bool val[M];
  int ii[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    ii[m] = ig + m * RG;
    val[m] = (j < r) && (ii[m] < r);
    const int idx = val[m] ? ii[m] * r + j : 0;
    const double lv = st->V[idx], lq = st->Q[idx], lp = st->P[idx];
    Vv[m] = val[m] ? lv : 0.0;
    Qv[m] = val[m] ? lq : 0.0;
    Pv[m] = val[m] ? lp : 0.0;
  }
  double rho = st->rho, lam = st->lam;
  double theta = (tid < p.n_theta) ? st->theta[tid] : 0.0;
  double gsum = (tid < p.n_theta) ? st->gradsum[tid] : 0.0;
  __syncthreads();
  // A_0 = [I; 0], K A_0 = first r columns of K, G_0 = K[0:r, 0:r] (exact Gram of the stored C)
  for (int idx = tid; idx < RB * r; idx += WG) {
    const int m = idx / r, c = idx - m * r;
    sA[idx] = (m == c) ? 1.0 : 0.0;
    sKA[idx] = sK[m * RB + c];
  }
#pragma unroll
  for (int m = 0; m < M; ++m) Gv[m] = val[m] ? sK[ii[m] * RB + j] : 0.0;
  __syncthreads();

  double s_last = 0.0, eta_last = 0.0, N_last = 0.0, phi = 1.0, omega = 1.0, ee_last = 0.0;
  for (int jb = 0; jb < b.nb; ++jb) {
    const long long kstep = b.k0 + jb + 1;   // 1-based step index
    // ---- S1: mu_bar, F ----
    if (tid < r) {
      double mb = s_mu[tid], f = 1.0;
      if (p.dyn_kind == 1) {
        const double arg = 2.0 * M_PI * theta * (double)kstep + mb;
        mb = cos(arg);
        f = -sin(arg);
      }
      s_mub[tid] = mb;
      s_f[tid] = f;
    }
    __syncthreads();
    // ---- S2: Pbar, w = V mu_bar, <G, Pbar>, s, eta, N, kappa ----
    double Pb[M];
    double part = 0.0, gp = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      Pb[m] = val[m] ? (p.pbar_predict ? s_f[ii[m]] * Pv[m] * s_f[j] + Qv[m] : Pv[m]) : 0.0;
      part += val[m] ? Vv[m] * s_mub[ii[m]] : 0.0;
      gp += Gv[m] * Pb[m];
    }
    col_reduce<RPAD>(part, s_red, s_w);
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < RPAD; ++l) s += s_mub[l] * s_w[l];
    double eta = rho;
    if (p.eta_full) eta += block_sum(gp, s4) / dd;
    const double N = s + eta;
    const double invN = fast_rcp(N);
    const double kappa = fast_rcp(rho + s);
    // ---- S3: P+ = (Pbar^-1 + kappa G)^-1 ----
    double Pp[M];
    if (p.coef_update) {
      double A1[M], Gk[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        A1[m] = val[m] ? Pb[m] : ((ii[m] == j && j < r2) ? 1.0 : 0.0);
        Gk[m] = kappa * Gv[m];
      }
      spd_update_solve<RPAD>(A1, Gk, r2, j, ig, rowbuf, errflag);
#pragma unroll
      for (int m = 0; m < M; ++m) Pp[m] = val[m] ? A1[m] : 0.0;
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) Pp[m] = Pb[m];
    }
    // ---- S4: coefficient space: b = A mu_bar, Ka = K[:, r+jb] - KA mu_bar, a = u - b ----
    {
      const int mrow = tid & (RB - 1), qtr = tid >> 6;        // 4 quarter-row partial dots per row
      const int c0 = qtr * (RPAD / 4), c1 = min(c0 + RPAD / 4, r);
      double pb = 0.0, pk = 0.0;
      for (int c = c0; c < c1; ++c) {
        const double mu_c = s_mub[c];
        pb += sA[mrow * r + c] * mu_c;
        pk += sKA[mrow * r + c] * mu_c;
      }
      s_p2[(qtr * RB + mrow) * 2] = pb;
      s_p2[(qtr * RB + mrow) * 2 + 1] = pk;
    }
    __syncthreads();
    if (tid < RB) {
      double bm = 0.0, km = 0.0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) { bm += s_p2[(qq * RB + tid) * 2]; km += s_p2[(qq * RB + tid) * 2 + 1]; }
      const double am = (tid == r + jb ? 1.0 : 0.0) - bm;
      const double kam = sK[tid * RB + r + jb] - km;
      s_a[tid] = am;
      s_Ka[tid] = kam;
      b.Bcoef[(size_t)jb * RB + tid] = bm;
    }
    __syncthreads();
    // ---- h = A^T Ka (8 row groups x RPAD columns), ee = a . Ka ----
    {
      double ph = 0.0;
      if (j < r) {
        for (int m = ig; m < RB; m += RG) ph += sA[m * r + j] * s_Ka[m];
      }
      col_reduce<RPAD>(ph, s_red, s_h);
    }
    double ee = (tid < RB) ? s_a[tid] * s_Ka[tid] : 0.0;
    ee = block_sum(ee, s4);
    // ---- S5: mu = mu_bar + kappa P+ h, quad ----
    double quad = kappa * ee;
    double mu_new = 0.0;
    if (p.coef_update) {
      double pt = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) pt += val[m] ? Pp[m] * s_h[ii[m]] : 0.0;
      col_reduce<RPAD>(pt, s_red, s_vec);
      double hPh = 0.0;
#pragma unroll
      for (int l = 0; l < RPAD; ++l) hPh += s_h[l] * s_vec[l];
      quad -= kappa * kappa * hPh;
      if (tid < r) mu_new = s_mub[tid] + kappa * s_vec[tid];
    } else {
      if (tid < r) mu_new = s_mub[tid];
    }
    // theta gradient at the pre-update state
    if (tid < p.n_theta && p.dyn_kind == 1) {
      const double tk = (double)kstep;
      const double arg = 2.0 * M_PI * theta * tk + s_mu[tid];
      const double jt = -sin(arg) * (2.0 * M_PI * tk);
      const double wi = s_w[tid], hi = s_h[tid];
      double gf;
      if (p.robust) {
        const double D = lam * N;
        gf = dd * wi / N + 0.5 * (dd + lam) * (-2.0 * hi / D - 2.0 * lam * ee * wi / (D * D)) / (1.0 + ee / D);
      } else {
        gf = dd * wi * invN - hi * invN - ee * wi * invN * invN;
      }
      gsum += jt * gf;
    }
    double vscale = 1.0, pscale = 1.0, qscale = 1.0;
    phi = 1.0; omega = 1.0;
    if (p.robust) {
      const double ild = fast_rcp(lam + dd);
      phi = (lam + ee * invN) * ild;
      omega = (lam + quad) * ild;
      vscale = p.alpha * phi;
      if (p.coef_update) { pscale = p.beta * omega; qscale = omega; }
      rho *= omega;
      if (!p.fixed_lambda) lam += dd;
    }
    // ---- S6: updates ----
    const double wj = s_w[j & (RM - 1)];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (val[m]) {
        const double wi = s_w[ii[m]], hi = s_h[ii[m]], hj = s_h[j];
        Vv[m] = vscale * (Vv[m] - wi * wj * invN);
        Pv[m] = pscale * Pp[m];
        Gv[m] += (hi * wj + wi * hj) * invN + ee * (wi * wj) * (invN * invN);
        Qv[m] *= qscale;
      }
    }
    for (int idx = tid; idx < RB * r; idx += WG) {
      const int m = idx / r, c = idx - m * r;
      const double wc = s_w[c] * invN;
      sA[idx] += s_a[m] * wc;
      sKA[idx] += s_Ka[m] * wc;
    }
    __syncthreads();           // all reads of s_mu, s_w, s_h, s_a, s_Ka of this step are done
    if (tid < r) s_mu[tid] = mu_new;
    s_last = s; eta_last = eta; N_last = N; ee_last = ee;
    __syncthreads();
  }

  // ---- block end: coefficients and state back to memory ----
  for (int idx = tid; idx < RB * r; idx += WG) b.Acoef[idx] = sA[idx];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (val[m]) {
      const int idx = ii[m] * r + j;
      st->V[idx] = Vv[m];
      st->P[idx] = Pv[m];
      st->Q[idx] = Qv[m];
      st->G[idx] = Gv[m];
    }
  }
  if (tid < r) st->mu[tid] = s_mu[tid];
  if (tid < p.n_theta) st->gradsum[tid] = gsum;
  if (tid == 0) {
    st->k = b.k0 + b.nb;
    st->rho = rho; st->lam = lam; st->phi = phi; st->omega = omega; st->ee = ee_last;
    st->s_done = s_last; st->eta_done = eta_last; st->N_done = N_last;
    if (*errflag && st->err == 0) st->err = (int)(b.k0 + 1);
  }
}

inline size_t blk_filter_lds_bytes() {
  const size_t doubles = (size_t)RB * RB + 2 * (size_t)RB * RM / 2 + WG + 6 * RM + 2 * RB + 8 * RB + 4 * RM + 4 + 2;
  return (doubles * 8 + 15) & ~(size_t)15;
}

// ------------------------------------------------------------------------------------------
// C <- Z A_nb (one rounding to the storage type per block), y_hat_{k0+j} = Z b_j.  Thread per row,
// the row of Z in registers (float64), the coefficient matrices broadcast from LDS.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(WG) void psmf_blk_apply(BlockParams b) {
  constexpr int RH = RM / 2;               // r <= 32
  __shared__ double sA[RB * RH];           // RB x r
  __shared__ double sB[RB * RB];           // nb x RB, stored [j][m]
  const StepParams& p = b.sp;
  const int r = p.r, rp = p.rp, dl = p.d_local, tid = threadIdx.x, nb = b.nb;
  for (int idx = tid; idx < RB * r; idx += WG) sA[idx] = b.Acoef[idx];
  for (int idx = tid; idx < nb * RB; idx += WG) sB[idx] = b.Bcoef[idx];
  __syncthreads();
  T* __restrict__ C = reinterpret_cast<T*>(p.C);
  const T* __restrict__ Y = reinterpret_cast<const T*>(p.Y) + (size_t)(b.k0 - p.series_t0) * dl;
  T* __restrict__ YP = p.store_yp ? reinterpret_cast<T*>(p.YP) + (size_t)(b.k0 - p.series_t0) * dl : nullptr;
  for (int row = blockIdx.x * WG + tid; row < dl; row += gridDim.x * WG) {
    // this row of Z in registers: zc = the C part (coefficients 0..r), zy = the series part (r..r+nb)
    double zc[RH], zy[RB];
#pragma unroll
    for (int m = 0; m < RH; ++m) {
      const double v = (double)C[(size_t)row * rp + min(m, r - 1)];     // unconditional load, masked
      zc[m] = m < r ? v : 0.0;
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const double v = (double)Y[(size_t)min(q, nb - 1) * dl + row];
      zy[q] = q < nb ? v : 0.0;
    }
    for (int c = 0; c < r; ++c) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < RH; ++m) acc += zc[m] * sA[min(m, r - 1) * r + c];          // zc[m >= r] = 0
#pragma unroll
      for (int q = 0; q < RB; ++q) acc += zy[q] * sA[min(r + q, RB - 1) * r + c];     // zy[q >= nb] = 0
      C[(size_t)row * rp + c] = (T)acc;
    }
    if (YP) {
      for (int jb = 0; jb < nb; ++jb) {
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < RH; ++m) acc += zc[m] * sB[jb * RB + min(m, r - 1)];
#pragma unroll
        for (int q = 0; q < RB; ++q) acc += zy[q] * sB[jb * RB + min(r + q, RB - 1)];
        YP[(size_t)jb * dl + row] = (T)acc;
      }
    }
  }
}

}  // namespace psmf
