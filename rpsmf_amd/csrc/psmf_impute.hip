// Masked, batched small-d engine: the filter loops of ExperimentImpute
//   ProbabilisticSequentialMatrixFactorizer   ExperimentImpute/PSMF.py:40-95
//   robust_PSMF                               ExperimentImpute/rPSMF.py:40-148
// and the two baseline filters of the imputation tables that share their masked contractions
//   stochasticGradientStateSpaceMF (MLE-SMF)  ExperimentImpute/MLESMF.py:40-92   weights m_i / rho_i, C += gam / eta (m o e) x_p^T
//   temporalRegularizedMF (TMF)               ExperimentImpute/TMF.py:30-73      x_t = x_p + (nu I + G)^-1 C^T e, C += gam (m o e) x_p^T
// plus the RMSEM / compute_number_inside_bars reductions made on their outputs
// (ExperimentImpute/common.py:79-94), for a whole batch of independent replicas (seeds).
//
// One 256-thread workgroup per replica; the replica's entire state (C d x r, V, P, Q, x) lives
// in LDS / registers in float64 and the kernel runs all n_iter * n columns without returning to
// the host.  The observation weights change every column (mask), so the r x r Gram
// sum_i m_i c_i c_i^T is recomputed per column (no algebraic tracking); d <= a few hundred here.
// Per column: masked residual, augmented Gram [C | e]^T diag(m) [C | e] (gives G, b = C^T e, e^T e in
// one pass), the two symmetric sweep inversions of the reference's Woodbury form
// (PSMF.py:30-36), the Kalman update of x, the rank-1 updates of C and V, error bands.
// Everything is latency-bound; inputs of column t+1 are prefetched while column t is processed, and the barriers inside the
// column loop order LDS only (solve_barrier<true>): the prefetch and the per-column stores (X, bands) stay in flight across them.
#include "../../include/psmf_hip.h"
#include "psmf_kernels.hip"

#include <cmath>
#include <cstring>
#include <string>

namespace psmf {

constexpr int IR = 16;   // largest rank of the masked engine (experiments use r = 10)

struct ImputeParams {
  int d, n, r, n_iter, robust, want_bands;
  int method;              // 0 PSMF, 1 rPSMF (robust = 1), 2 MLE-SMF, 3 TMF
  double sig, lambda0, rho0;
  const double* Yorg;      // n x d (shared)
  const uint8_t* M;        // batch x n x d
  const uint8_t* Mmiss;    // batch x n x d
  double* C;               // batch x d x r
  double* X;               // batch x n x r
  const double* V0;
  const double* P0;
  const double* Q0;
  double* Epred;           // batch x n_iter
  double* Efull;           // batch x n_iter
  double* inside;          // batch
  double* Yrec;            // batch x n x d or null
  double* YrecL;
  double* YrecH;
  int* err;                // batch
};

__global__ __launch_bounds__(WG) void psmf_impute_kernel(ImputeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int d = p.d, n = p.n, r = p.r, tid = threadIdx.x, rep = blockIdx.x;
  const int r2 = r + (r & 1);
  const int ra = r + 1;                       // augmented row [c_i | e_i]
  const int npair = ra * (ra + 1) / 2;
  // ---- LDS carve (doubles) ----
  double* sC = sm;                            // d * r
  double* sV = sC + d * r;                    // IR * IR (stride r)
  double* sP = sV + IR * IR;
  double* sQ = sP + IR * IR;
  double* sG = sQ + IR * IR;                  // masked Gram, r x r
  double* sPp = sG + IR * IR;                 // P+ of the current column
  double* sx = sPp + IR * IR;                 // prior mean (previous column's posterior), IR
  double* sw = sx + IR;                       // V x
  double* sb = sw + IR;                       // C^T e  (unweighted)
  double* sz = sb + IR;                       // P+ C^T e
  double* se = sz + IR;                       // d: masked residual
  double* smk = se + d;                       // d: mask as 0/1 double
  double* syh = smk + d;                      // d: unmasked prediction
  double* sgp = syh + d;                      // 4 * 160 Gram partials
  double* rowbuf = sgp + 4 * 160;             // 4 * RM
  double* sred = rowbuf + 4 * RM;             // 8 scalars / reductions
  int* errflag = reinterpret_cast<int*>(sred + 8);

  const double* Yorg = p.Yorg;
  const uint8_t* Mk = p.M + (size_t)rep * n * d;
  const uint8_t* Mm = p.Mmiss + (size_t)rep * n * d;
  double* Cg = p.C + (size_t)rep * d * r;
  double* Xg = p.X + (size_t)rep * n * r;

  for (int idx = tid; idx < d * r; idx += WG) sC[idx] = Cg[idx];
  for (int idx = tid; idx < r * r; idx += WG) {
    sV[idx] = p.V0[idx];
    sP[idx] = p.P0[idx];
    sQ[idx] = p.Q0[idx];
  }
  if (tid < r) sx[tid] = Xg[(size_t)(n - 1) * r + tid];   // t = 0 wraps to the last column (PSMF.py:65)
  if (tid == 0) *errflag = 0;
  double rho = p.rho0, lam = p.lambda0;
  const double dd = (double)d;
  // pair index -> (a, b), a <= b, over the ra columns of the augmented Gram (row-major upper
  // triangle); constant over the run.  4 row slices x 64 pair slots x up to 3 rounds (npair <= 153).
  const int q = tid & 63, slice = tid >> 6;
  int pa[3], pb[3];
#pragma unroll
  for (int rd = 0; rd < 3; ++rd) {
    int a = 0, rem = q + 64 * rd;
    while (a < ra && rem >= ra - a) { rem -= ra - a; ++a; }
    pa[rd] = a; pb[rd] = a + rem;
  }
  int ta = 0, tb = 0;   // pair of index tid, for the reduction of the partials
  { int a = 0, rem = tid; while (a < ra && rem >= ra - a) { rem -= ra - a; ++a; } ta = a; tb = a + rem; }
  __syncthreads();

  // element of the r x r solve owned by this thread (RPAD = 16: one element per thread)
  const int ei = tid / 16, ec = tid % 16;
  const bool ein = ei < r && ec < r;

  unsigned long long nmiss_l = 0;   // per-thread counts, rows tid, tid + 256, ...
  const bool sgd = p.method >= 2;     // MLE-SMF / TMF: gradient step on C along x_p, no V
  const bool tmf = p.method == 3;
  for (int it = 0; it < p.n_iter; ++it) {
    const double gam = 1e-6 / pow((double)(it + 1), 0.7);     // MLESMF.py:59-60, TMF.py:46-48
    if (p.robust) {                 // rPSMF.py:77-79: Q, R, lambda restart every pass; V, P, C carry over
      for (int idx = tid; idx < r * r; idx += WG) sQ[idx] = p.Q0[idx];
      rho = p.rho0;
      lam = p.lambda0;
    }
    double sse_pred = 0.0;
    unsigned long long inside_l = 0;
    nmiss_l = 0;
    // prefetch column 0
    double ny[2];
    uint8_t nm[2], nmm[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * WG;
      ny[u] = i < d ? Yorg[i] : 0.0;
      nm[u] = i < d ? Mk[i] : 0;
      nmm[u] = i < d ? Mm[i] : 0;
    }
    __syncthreads();
    for (int t = 0; t < n; ++t) {
      double yv[2];
      uint8_t mv[2], mmv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { yv[u] = ny[u]; mv[u] = nm[u]; mmv[u] = nmm[u]; }
      if (t + 1 < n) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = tid + u * WG;
          const size_t off = (size_t)(t + 1) * d + i;
          ny[u] = i < d ? Yorg[off] : 0.0;
          nm[u] = i < d ? Mk[off] : 0;
          nmm[u] = i < d ? Mm[off] : 0;
        }
      }
      // ---- A: residual rows, w = V x ----
      double yh[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + u * WG;
        yh[u] = 0.0;
        if (i < d) {
          double dot = 0.0;
          for (int l = 0; l < r; ++l) dot += sC[i * r + l] * sx[l];
          const double mi = mv[u] ? 1.0 : 0.0;
          const double yi = mv[u] ? yv[u] : 0.0;     // Y is 0 where unobserved (PSMF.py:147-148)
          se[i] = mi * (yi - dot);
          smk[i] = mi;
          syh[i] = dot;
          yh[u] = dot;
        }
      }
      if (tid >= WG - IR && tid - (WG - IR) < r) {   // last wave: w = V x
        const int i = tid - (WG - IR);
        double a = 0.0;
        for (int l = 0; l < r; ++l) a += sV[i * r + l] * sx[l];
        sw[i] = a;
      }
      solve_barrier<true>();
      double s = 0.0;
      for (int l = 0; l < r; ++l) s += sx[l] * sw[l];
      // weights of the observed rows: PSMF / rPSMF 1 / (rho + s) (PSMF.py:71-72), MLE-SMF 1 / rho (MLESMF.py:70), TMF 1
      const double kappa = tmf ? 1.0 : (sgd ? 1.0 / rho : 1.0 / (rho + s));
      // ---- B: augmented masked Gram  [C | e]^T diag(m) [C | e]  + sum(m) ----
#pragma unroll
      for (int rd = 0; rd < 3; ++rd) {
        const int qq = q + 64 * rd;
        if (qq < npair) {
          double acc = 0.0;
          for (int i = slice; i < d; i += 4) {
            const double va = pa[rd] < r ? sC[i * r + pa[rd]] : se[i];
            const double vb = pb[rd] < r ? sC[i * r + pb[rd]] : se[i];
            acc += smk[i] * va * vb;
          }
          sgp[slice * 160 + qq] = acc;
        }
      }
      double msum_l = 0.0;
      if (tid < 64) for (int i = tid; i < d; i += 64) msum_l += smk[i];
      msum_l = wave_sum(msum_l);
      if (tid == 0) sred[0] = msum_l;
      solve_barrier<true>();
      if (tid < npair) {
        const double g = (sgp[tid] + sgp[160 + tid]) + (sgp[320 + tid] + sgp[480 + tid]);
        if (tb < r) { sG[ta * r + tb] = g; sG[tb * r + ta] = g; }
        else if (ta < r) sb[ta] = g;         // C^T e   (e is already masked)
        else sred[1] = g;                    // e^T e
      }
      solve_barrier<true>();
      const double msum = sred[0], ee = sred[1];
      // ---- C: PP = P + Q, <G, PP>, P+ = (PP^-1 + kappa G)^-1 ----
      double A1[1], G1[1];
      // TMF: (nu I + G)^-1 is the same solve with PP = I / nu, nu = 2 (TMF.py:47,60)
      const double ppv = ein ? (tmf ? (ei == ec ? 0.5 : 0.0) : 0.5 * ((sP[ei * r + ec] + sQ[ei * r + ec]) + (sP[ec * r + ei] + sQ[ec * r + ei]))) : 0.0;
      const double gv = ein ? sG[ei * r + ec] : 0.0;
      A1[0] = ein ? ppv : ((ei == ec && ei < r2) ? 1.0 : 0.0);
      G1[0] = kappa * gv;
      double gpp = wave_sum(ppv * gv);
      if ((tid & 63) == 0) sred[4 + (tid >> 6)] = gpp;
      spd_update_solve<16, true>(A1, G1, r2, ec, ei, rowbuf, errflag);   // contains barriers
      if (ein) sPp[ei * r + ec] = A1[0];
      solve_barrier<true>();
      const double trGP = (sred[4] + sred[5]) + (sred[6] + sred[7]);
      const double eta = (rho * msum + trGP) / dd;      // divide by d, not by #observed (PSMF.py:77)
      const double N = s + eta;
      // ---- D: x_t = x_p + P+ (kappa C^T e),  omega, phi ----
      double xnew = 0.0;
      if (tid < r) {
        double a = 0.0;
        for (int l = 0; l < r; ++l) a += 0.5 * (sPp[tid * r + l] + sPp[l * r + tid]) * sb[l];
        sz[tid] = a;
        xnew = sx[tid] + kappa * a;
        Xg[(size_t)t * r + tid] = xnew;                 // the reference overwrites X[:, t] in place
      }
      double bPb = 0.0;
      if (p.robust) {
        solve_barrier<true>();
        for (int l = 0; l < r; ++l) bPb += sb[l] * sz[l];
        bPb *= kappa * kappa;
      }
      double omega = 1.0, phi = 1.0;
      if (p.robust) {
        omega = (lam + kappa * ee - bPb) / (lam + dd);  // rPSMF.py:105
        phi = (lam + ee / N) / (lam + dd);              // rPSMF.py:112-114 (e = 0 on unobserved rows)
      }
      const double wsc = 1.0 / N;
      // ---- E: C, V, P, bands, metrics ----
      const double csc = tmf ? gam : gam / eta;        // MLESMF.py:79, TMF.py:63
      for (int idx = tid; idx < d * r; idx += WG) {
        const int i = idx / r, l = idx - i * r;
        sC[idx] += sgd ? se[i] * sx[l] * csc : se[i] * sw[l] * wsc;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + u * WG;
        if (i < d) {
          const double band = p.sig * sqrt(p.robust ? (s * (mv[u] ? 1.0 : 0.0) + eta) : (sgd ? eta : N));   // rPSMF.py:121-123 / PSMF.py:83-84 / MLESMF.py:81-82
          const double lo = yh[u] - band, hi = yh[u] + band;
          if (mmv[u]) {
            const double dl = yh[u] - yv[u];
            sse_pred += dl * dl;
            nmiss_l += 1;
            if (it == p.n_iter - 1 && !tmf && yv[u] < hi && lo < yv[u]) inside_l += 1;
          }
          if (p.want_bands) {
            const size_t off = ((size_t)rep * n + t) * d + i;
            p.Yrec[off] = yh[u];
            p.YrecL[off] = lo;
            p.YrecH[off] = hi;
          }
        }
      }
      solve_barrier<true>();   // all reads of sV, sx, sP, sQ, sPp of this column are done
      if (ein) {
        if (!sgd) sV[ei * r + ec] = phi * (sV[ei * r + ec] - sw[ei] * sw[ec] * wsc);
        sP[ei * r + ec] = omega * 0.5 * (sPp[ei * r + ec] + sPp[ec * r + ei]);
        if (p.robust) sQ[ei * r + ec] *= omega;
      }
      if (tid < r) sx[tid] = xnew;
      if (p.robust) { rho *= omega; lam += dd; }
      solve_barrier<true>();
    }
    // ---- end of pass: RMSE of the one-step predictions, RMSE of C @ X, coverage ----
    double nm_d = (double)nmiss_l;
    double sse_full = 0.0;
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * WG;
      if (i < d) {
        for (int t = 0; t < n; ++t) {
          if (Mm[(size_t)t * d + i]) {
            double dot = 0.0;
            // X was written by other threads of this workgroup: read around this CU's L1
            for (int l = 0; l < r; ++l) dot += sC[i * r + l] * __builtin_nontemporal_load(&Xg[(size_t)t * r + l]);
            const double dl = dot - Yorg[(size_t)t * d + i];
            sse_full += dl * dl;
          }
        }
      }
    }
    double v0 = wave_sum(sse_pred), v1 = wave_sum(sse_full), v2 = wave_sum(nm_d), v3 = wave_sum((double)inside_l);
    __syncthreads();
    if ((tid & 63) == 0) { sgp[(tid >> 6) * 4 + 0] = v0; sgp[(tid >> 6) * 4 + 1] = v1; sgp[(tid >> 6) * 4 + 2] = v2; sgp[(tid >> 6) * 4 + 3] = v3; }
    __syncthreads();
    if (tid == 0) {
      const double tp = (sgp[0] + sgp[4]) + (sgp[8] + sgp[12]);
      const double tf = (sgp[1] + sgp[5]) + (sgp[9] + sgp[13]);
      const double tn = (sgp[2] + sgp[6]) + (sgp[10] + sgp[14]);
      const double ti = (sgp[3] + sgp[7]) + (sgp[11] + sgp[15]);
      p.Epred[(size_t)rep * p.n_iter + it] = sqrt(tp / tn);
      p.Efull[(size_t)rep * p.n_iter + it] = sqrt(tf / tn);
      if (it == p.n_iter - 1) p.inside[rep] = ti / tn;
    }
    __syncthreads();
  }
  for (int idx = tid; idx < d * r; idx += WG) Cg[idx] = sC[idx];
  if (tid == 0) p.err[rep] = *errflag;
}

inline size_t impute_lds_bytes(int d, int r) {
  const size_t doubles = (size_t)d * r + 5 * IR * IR + 4 * IR + 3 * (size_t)d + 4 * 160 + 4 * RM + 8 + 2;
  return (doubles * 8 + 15) & ~(size_t)15;
}

}  // namespace psmf

extern "C" int psmf_impute_run(const psmf_impute_config* cfg, const double* YorgInt, const uint8_t* M,
                               const uint8_t* Mmiss, double* C, double* X, const double* V, const double* P,
                               const double* Q, double rho, double* Epred, double* Efull, double* inside,
                               double* Yrec, double* YrecL, double* YrecH, float* elapsed_ms) {
  using namespace psmf;
  auto fail = [&](int code, const std::string& msg) { g_create_error = "psmf_impute_run: " + msg; return code; };
  if (!cfg || !YorgInt || !M || !Mmiss || !C || !X || !V || !P || !Q || !Epred || !Efull || !inside)
    return fail(PSMF_ERR_ARG, "null argument");
  if (cfg->abi_version != PSMF_ABI_VERSION) return fail(PSMF_ERR_ARG, "ABI version mismatch");
  const int d = cfg->d, n = cfg->n, r = cfg->r, B = cfg->batch;
  if (r < 1 || r > IR) return fail(PSMF_ERR_ARG, "need 1 <= r <= 16");
  if (d < 1 || d > 2 * WG) return fail(PSMF_ERR_ARG, "need 1 <= d <= 512 (one workgroup per replica)");
  if (n < 2 || B < 1 || cfg->n_iter < 1) return fail(PSMF_ERR_ARG, "bad n / batch / n_iter");
  if (cfg->method < 0 || cfg->method > 3) return fail(PSMF_ERR_ARG, "method must be 0 (PSMF), 1 (rPSMF), 2 (MLE-SMF) or 3 (TMF)");
  if (cfg->want_bands && (!Yrec || !YrecL || !YrecH)) return fail(PSMF_ERR_ARG, "want_bands needs Yrec, YrecL, YrecH");
  const size_t lds = impute_lds_bytes(d, r);
  if (lds > 160 * 1024) return fail(PSMF_ERR_ARG, "d * r does not fit one workgroup's LDS");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(PSMF_ERR_NO_DEVICE, "no HIP device visible");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(PSMF_ERR_ARG, "bad device ordinal");

#define I_TRY(expr)                                                                          \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) { rc = fail(PSMF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); goto done; } \
  } while (0)

  int rc = PSMF_OK;
  const size_t nd = (size_t)n * d, bnd = (size_t)B * nd;
  double *dY = nullptr, *dC = nullptr, *dX = nullptr, *dV = nullptr, *dP = nullptr, *dQ = nullptr, *dEp = nullptr,
         *dEf = nullptr, *dIn = nullptr, *dYr = nullptr, *dYl = nullptr, *dYh = nullptr;
  uint8_t *dM = nullptr, *dMm = nullptr;
  int* dErr = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<int> herr(B, 0);
  ImputeParams ip;
  I_TRY(hipSetDevice(cfg->device));
  I_TRY(hipMalloc((void**)&dY, nd * 8));
  I_TRY(hipMalloc((void**)&dM, bnd));
  I_TRY(hipMalloc((void**)&dMm, bnd));
  I_TRY(hipMalloc((void**)&dC, (size_t)B * d * r * 8));
  I_TRY(hipMalloc((void**)&dX, (size_t)B * n * r * 8));
  I_TRY(hipMalloc((void**)&dV, r * r * 8));
  I_TRY(hipMalloc((void**)&dP, r * r * 8));
  I_TRY(hipMalloc((void**)&dQ, r * r * 8));
  I_TRY(hipMalloc((void**)&dEp, (size_t)B * cfg->n_iter * 8));
  I_TRY(hipMalloc((void**)&dEf, (size_t)B * cfg->n_iter * 8));
  I_TRY(hipMalloc((void**)&dIn, (size_t)B * 8));
  I_TRY(hipMalloc((void**)&dErr, (size_t)B * 4));
  if (cfg->want_bands) {
    I_TRY(hipMalloc((void**)&dYr, bnd * 8));
    I_TRY(hipMalloc((void**)&dYl, bnd * 8));
    I_TRY(hipMalloc((void**)&dYh, bnd * 8));
  }
  I_TRY(hipMemcpy(dY, YorgInt, nd * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dM, M, bnd, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dMm, Mmiss, bnd, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dC, C, (size_t)B * d * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dX, X, (size_t)B * n * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dV, V, r * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dP, P, r * r * 8, hipMemcpyHostToDevice));
  I_TRY(hipMemcpy(dQ, Q, r * r * 8, hipMemcpyHostToDevice));
  ip.d = d; ip.n = n; ip.r = r; ip.n_iter = cfg->n_iter; ip.robust = cfg->method == 1; ip.method = cfg->method; ip.want_bands = cfg->want_bands;
  ip.sig = cfg->sig; ip.lambda0 = cfg->lambda0; ip.rho0 = rho;
  ip.Yorg = dY; ip.M = dM; ip.Mmiss = dMm; ip.C = dC; ip.X = dX; ip.V0 = dV; ip.P0 = dP; ip.Q0 = dQ;
  ip.Epred = dEp; ip.Efull = dEf; ip.inside = dIn; ip.Yrec = dYr; ip.YrecL = dYl; ip.YrecH = dYh; ip.err = dErr;
  if (lds > 48 * 1024)
    I_TRY(hipFuncSetAttribute((const void*)psmf_impute_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  I_TRY(hipEventCreate(&e0));
  I_TRY(hipEventCreate(&e1));
  I_TRY(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(psmf_impute_kernel, dim3(B), dim3(WG), lds, 0, ip);
  I_TRY(hipGetLastError());
  I_TRY(hipEventRecord(e1, 0));
  I_TRY(hipEventSynchronize(e1));
  if (elapsed_ms) I_TRY(hipEventElapsedTime(elapsed_ms, e0, e1));
  I_TRY(hipMemcpy(C, dC, (size_t)B * d * r * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(X, dX, (size_t)B * n * r * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(Epred, dEp, (size_t)B * cfg->n_iter * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(Efull, dEf, (size_t)B * cfg->n_iter * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(inside, dIn, (size_t)B * 8, hipMemcpyDeviceToHost));
  I_TRY(hipMemcpy(herr.data(), dErr, (size_t)B * 4, hipMemcpyDeviceToHost));
  if (cfg->want_bands) {
    I_TRY(hipMemcpy(Yrec, dYr, bnd * 8, hipMemcpyDeviceToHost));
    I_TRY(hipMemcpy(YrecL, dYl, bnd * 8, hipMemcpyDeviceToHost));
    I_TRY(hipMemcpy(YrecH, dYh, bnd * 8, hipMemcpyDeviceToHost));
  }
  for (int b = 0; b < B; ++b)
    if (herr[b]) { rc = fail(PSMF_ERR_NUMERIC, "singular r x r system in replica " + std::to_string(b)); break; }
done:
  hipFree(dY); hipFree(dM); hipFree(dMm); hipFree(dC); hipFree(dX); hipFree(dV); hipFree(dP); hipFree(dQ);
  hipFree(dEp); hipFree(dEf); hipFree(dIn); hipFree(dErr); hipFree(dYr); hipFree(dYl); hipFree(dYh);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
#undef I_TRY
  return rc;
}
