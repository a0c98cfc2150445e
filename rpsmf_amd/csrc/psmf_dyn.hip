// State-transition functions f(theta, x, t) of the filter evaluated on the device, with analytic Jacobians
// (the reference differentiates them with autograd: pypsmf/psmf/psmf.py:41-44,107-115,167-177):
//
//   PSMF_DYN_RANDOM_WALK  f = x                                              nonlinearities.py:42-56
//   PSMF_DYN_COS_PHASE    f = cos(2 pi theta t + x)                          ExperimentSynthetic/synthetic_psmf.py:105-106
//   PSMF_DYN_SCALED_WALK  f = A x (+ b)                                      nonlinearities.py:59-78
//   PSMF_DYN_SINUSOID     f = [A] sin(2 pi b t + [c o] x)                    nonlinearities.py:81-114
//   PSMF_DYN_FOURIER      f = sum_n A_n sin(2 pi b_n t + c_n o x) + D_n cos(2 pi e_n t + f_n o x)    nonlinearities.py:117-150
//
// All but the scaled walk are sums of TERMS  M_t trig_t(2 pi b_t t + c_t o x)  (M_t an r x r matrix or the identity, c_t a
// gain vector or ones, trig_t = sin or cos); theta packs the blocks in the reference's order (`dims`).  Per step the
// filter needs  mu_bar = f(theta, mu, k),  F = df/dx  (P_bar = F P F^T + Q)  and, for the theta gradient,
// J_theta^T g_f with g_f = d(incremental likelihood)/df:
//     dF:        F[i][j]      = sum_t M_t[i][j] trig_t'(arg_tj) c_tj
//     d/dM_t:    g[i][j]     += g_f[i] trig_t(arg_tj)
//     d/db_t:    g[j]        += (M_t^T g_f)_j trig_t'(arg_tj) 2 pi k
//     d/dc_t:    g[j]        += (M_t^T g_f)_j trig_t'(arg_tj) x_j
// Everything here is O(r^2) per term and runs inside the one-workgroup r x r stage of the blocked engine
// (psmf_blk_filter); theta, the summed gradient and the Adam moments live in global memory (StepParams.theta ...:
// n_theta can be 2 N r^2 + 4 N r), every element always touched by the same thread.
#pragma once
#include "psmf_device.h"

namespace psmf {

constexpr int DYN_RANDOM_WALK = 0, DYN_COS_PHASE = 1, DYN_SCALED_WALK = 2, DYN_SINUSOID = 3, DYN_FOURIER = 4, DYN_HOST = 5;
constexpr int DYN_MAX_TERMS = 8;     // Fourier: 2 N terms, N <= 4

struct DynTerm {
  int m_off, b_off, c_off;   // offsets into theta; -1: identity matrix / unit gains
  int is_cos;
};

__host__ __device__ inline int dyn_n_terms(int kind, int nterms_cfg) {
  return kind == DYN_COS_PHASE || kind == DYN_SINUSOID ? 1 : (kind == DYN_FOURIER ? 2 * nterms_cfg : 0);
}

// dense Jacobian?  (otherwise F is diagonal)
__host__ __device__ inline bool dyn_dense(int kind, int flags) {
  return kind == DYN_SCALED_WALK || kind == DYN_FOURIER || (kind == DYN_SINUSOID && (flags & 1));
}

__host__ __device__ inline int dyn_n_theta(int kind, int flags, int N, int r) {
  switch (kind) {
    case DYN_COS_PHASE: return r;
    case DYN_SCALED_WALK: return r * r + ((flags & 1) ? r : 0);
    case DYN_SINUSOID: return ((flags & 1) ? r * r : 0) + r + ((flags & 2) ? r : 0);
    case DYN_FOURIER: return N * (2 * r * r + 4 * r);
    default: return 0;
  }
}

__host__ __device__ inline DynTerm dyn_term(int kind, int flags, int N, int r, int t) {
  DynTerm d = {-1, 0, -1, 0};
  if (kind == DYN_COS_PHASE) { d.is_cos = 1; }
  else if (kind == DYN_SINUSOID) {
    int o = 0;
    if (flags & 1) { d.m_off = 0; o = r * r; }
    d.b_off = o;
    if (flags & 2) d.c_off = o + r;
  } else if (kind == DYN_FOURIER) {       // term t = 2 n (sin, A_n, b_n, c_n) or 2 n + 1 (cos, D_n, e_n, f_n)
    const int n = t >> 1, odd = t & 1;
    d.m_off = t * r * r;
    d.b_off = 2 * N * r * r + (4 * n + 2 * odd) * r;
    d.c_off = d.b_off + r;
    d.is_cos = odd;
  }
  return d;
}

// sin(pi a), cos(pi a): exact reduction to |f| <= 1/4 half-turns (a - n / 2 is exact in float64), the fdlibm kernel polynomials on
// |pi f| <= pi / 4, quadrant rotation -- branch-free, ~40 instructions.  The arguments here are 2 pi b k + c x with k up to the
// series length: sincos() of the radian argument goes through its large-argument path (scratch memory, branches) for every
// element; in half-turns the reduction is one rint.  Agrees with sincos(2 pi b k + c x) to the rounding of that argument (~1e-12).
__device__ __forceinline__ void dyn_sincospi(const double a, double& sn, double& cs) {
  const double n = rint(2.0 * a);
  const double f = fma(-0.5, n, a);
  const double x = f * M_PI, z = x * x;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                           -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double s0 = fma(x * z, ps, x);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                           2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int qd = (int)((long long)n) & 3;
  const double s1 = (qd & 1) ? c0 : s0, c1 = (qd & 1) ? s0 : c0;
  sn = (qd == 2 || qd == 3) ? -s1 : s1;
  cs = (qd == 1 || qd == 2) ? -c1 : c1;
}

// trig_t(arg_tj) and trig_t'(arg_tj) of every term at x = s_x, step index tk: the first phase of the forward pass, and all the
// gradient pass needs of it (the per-step engine's serial stage evaluates them again at the step's end instead of keeping
// them across two launches).  All NTH threads; ends with a barrier.  Nothing to do for the scaled walk (no trig terms).
template <int NTH>
__device__ __forceinline__ void dyn_trig(const StepParams& p, const double tk, const double* s_x, double* s_val, double* s_tp, const int tid) {
  const int r = p.r, kind = p.dyn_kind, flags = p.dyn_flags, N = p.dyn_terms;
  const double* th = p.theta;
  const int nt = dyn_n_terms(kind, N);
  for (int idx = tid; idx < nt * r; idx += NTH) {
    const int t = idx / r, j = idx - t * r;
    const DynTerm d = dyn_term(kind, flags, N, r, t);
    const double c = d.c_off >= 0 ? th[d.c_off + j] : 1.0;
    double sn, cs;
    dyn_sincospi(2.0 * th[d.b_off + j] * tk + (c * s_x[j]) * 0.31830988618379067154, sn, cs);
    s_val[t * RM + j] = d.is_cos ? cs : sn;
    s_tp[t * RM + j] = d.is_cos ? -sn : cs;
  }
  __syncthreads();
}

// Forward pass, all NTH threads of the workgroup; ends with a barrier.
//   s_x    mu_{k-1} (LDS, r)             s_mub  out: mu_bar (r)
//   s_fd   out: diagonal of F when !dense (r)
//   sF     out: dense F, row stride ldf, when dense
//   s_val, s_tp   out: [term][RM] trig(arg), trig'(arg) for the gradient
//   s_part scratch, DYN_MAX_TERMS * RM doubles (partial mat-vecs of the terms)
// Every O(nt r^2) loop is spread over the workgroup (one (row, term) or (row, column) pair per thread): with r threads
// walking the terms one after the other a FourierBasis step spent 8 000 cycles here (tools/blkgen_prof.hip).
template <int NTH>
__device__ __forceinline__ void dyn_forward(const StepParams& p, const double tk, const double* s_x, double* s_mub, double* s_fd,
                                            double* sF, const int ldf, double* s_val, double* s_tp, double* s_part, const int tid) {
  const int r = p.r, kind = p.dyn_kind, flags = p.dyn_flags, N = p.dyn_terms;
  const double* th = p.theta;
  if (kind == DYN_RANDOM_WALK) {
    if (tid < r) { s_mub[tid] = s_x[tid]; s_fd[tid] = 1.0; }
    __syncthreads();
    return;
  }
  if (kind == DYN_SCALED_WALK) {
    if (tid < r) {
      double a = (flags & 1) ? th[r * r + tid] : 0.0;
      for (int j = 0; j < r; ++j) a += th[tid * r + j] * s_x[j];
      s_mub[tid] = a;
    }
    for (int idx = tid; idx < r * r; idx += NTH) sF[(idx / r) * ldf + idx % r] = th[idx];
    __syncthreads();
    return;
  }
  const int nt = dyn_n_terms(kind, N);
  dyn_trig<NTH>(p, tk, s_x, s_val, s_tp, tid);
  const bool dense = dyn_dense(kind, flags);
  if (!dense && kind != DYN_FOURIER) {
    // no matrix in any term (cos-phase, unscaled sinusoid -- the ExperimentSynthetic modes): mu_bar and the diagonal of F
    // straight from the trig values, one barrier less
    if (tid < r) {
      double a = 0.0, fd = 0.0;
      for (int t = 0; t < nt; ++t) {
        const DynTerm d = dyn_term(kind, flags, N, r, t);
        a += s_val[t * RM + tid];
        fd += s_tp[t * RM + tid] * (d.c_off >= 0 ? th[d.c_off + tid] : 1.0);
      }
      s_mub[tid] = a;
      s_fd[tid] = fd;
    }
    __syncthreads();
    return;
  }
  // per (term, row): the term's share of mu_bar_i (and of the diagonal of F for identity-matrix terms)
  for (int idx = tid; idx < nt * r; idx += NTH) {
    const int t = idx / r, i = idx - t * r;
    const DynTerm d = dyn_term(kind, flags, N, r, t);
    double a = 0.0;
    if (d.m_off >= 0) {
      const double* row = th + d.m_off + i * r;
      for (int j = 0; j < r; ++j) a += row[j] * s_val[t * RM + j];
    } else {
      a = s_val[t * RM + i];
    }
    s_part[t * RM + i] = a;
  }
  if (dense) {
    for (int idx = tid; idx < r * r; idx += NTH) {
      const int i = idx / r, j = idx - i * r;
      double a = 0.0;
      for (int t = 0; t < nt; ++t) {
        const DynTerm d = dyn_term(kind, flags, N, r, t);
        const double dv = s_tp[t * RM + j] * (d.c_off >= 0 ? th[d.c_off + j] : 1.0);
        a += (d.m_off >= 0 ? th[d.m_off + idx] : ((i == j) ? 1.0 : 0.0)) * dv;
      }
      sF[i * ldf + j] = a;
    }
  }
  __syncthreads();
  if (tid < r) {
    double a = 0.0, fd = 0.0;
    for (int t = 0; t < nt; ++t) {
      const DynTerm d = dyn_term(kind, flags, N, r, t);
      a += s_part[t * RM + tid];
      if (d.m_off < 0) fd += s_tp[t * RM + tid] * (d.c_off >= 0 ? th[d.c_off + tid] : 1.0);
    }
    s_mub[tid] = a;
    if (!dense) s_fd[tid] = fd;
  }
  __syncthreads();
}

// Gradient pass: gradsum += J_theta^T g_f.  s_gf: g_f (LDS, r), s_x: mu_{k-1}.  All NTH threads; the caller has a barrier
// between writing s_gf and this call; ends with a barrier.  One (term, column) pair or one matrix element per thread.
template <int NTH>
__device__ __forceinline__ void dyn_backward(const StepParams& p, const double tk, const double* s_x, const double* s_gf,
                                             const double* s_val, const double* s_tp, const int tid) {
  const int r = p.r, kind = p.dyn_kind, flags = p.dyn_flags, N = p.dyn_terms;
  const double* th = p.theta;
  double* g = p.gradsum;
  if (kind == DYN_RANDOM_WALK) return;
  if (kind == DYN_SCALED_WALK) {
    for (int idx = tid; idx < r * r; idx += NTH) g[idx] += s_gf[idx / r] * s_x[idx % r];
    if ((flags & 1) && tid < r) g[r * r + tid] += s_gf[tid];
    __syncthreads();
    return;
  }
  const int nt = dyn_n_terms(kind, N);
  // d/dM_t[i][j] = g_f[i] trig_t(arg_tj)
  for (int idx = tid; idx < nt * r * r; idx += NTH) {
    const int t = idx / (r * r), e = idx - t * r * r;
    const DynTerm d = dyn_term(kind, flags, N, r, t);
    if (d.m_off >= 0) g[d.m_off + e] += s_gf[e / r] * s_val[t * RM + e % r];
  }
  // d/db_t[j], d/dc_t[j] = (M_t^T g_f)_j trig_t'(arg_tj) {2 pi k, x_j}
  for (int idx = tid; idx < nt * r; idx += NTH) {
    const int t = idx / r, j = idx - t * r;
    const DynTerm d = dyn_term(kind, flags, N, r, t);
    double u;
    if (d.m_off >= 0) {
      u = 0.0;
      for (int i = 0; i < r; ++i) u += th[d.m_off + i * r + j] * s_gf[i];
    } else {
      u = s_gf[j];
    }
    const double ut = u * s_tp[t * RM + j];
    g[d.b_off + j] += ut * (2.0 * M_PI * tk);
    if (d.c_off >= 0) g[d.c_off + j] += ut * s_x[j];
  }
  __syncthreads();
}

// Optimiser step on theta inside the time loop (PSMFRecursive, psmf.py:224-248,299-304; StepParams.recursive = 1 Adam, 2 SGD): bias correction with the step index,
// projection theta >= 0, gradient sum restarted.  All NTH threads; ends with a barrier.
template <int NTH>
__device__ __forceinline__ void dyn_adam_step(const StepParams& p, const long long knext, const int tid) {
  const double kk = (double)knext;
  const double lr = p.lr_steps > 0.0 ? p.lr * pow(p.lr_end / p.lr, kk / p.lr_steps) : p.lr;
  const double c1 = 1.0 / (1.0 - pow(p.b1, kk)), c2 = 1.0 / (1.0 - pow(p.b2, kk));
  if (p.recursive == 2) {      // plain SGD (psmf.py:244-248): theta <- max(theta - lr g, 0)
    for (int idx = tid; idx < p.n_theta; idx += NTH) {
      p.theta[idx] = fmax(p.theta[idx] - lr * p.gradsum[idx], 0.0);
      p.gradsum[idx] = 0.0;
    }
    __threadfence_block();
    __syncthreads();
    return;
  }
  for (int idx = tid; idx < p.n_theta; idx += NTH) {
    const double gs = p.gradsum[idx];
    const double am = p.b1 * p.adam_m[idx] + (1.0 - p.b1) * gs;
    const double av = p.b2 * p.adam_v[idx] + (1.0 - p.b2) * gs * gs;
    p.adam_m[idx] = am;
    p.adam_v[idx] = av;
    p.theta[idx] = fmax(p.theta[idx] - lr * (am * c1) / (sqrt(av * c2) + 1e-8), 0.0);
    p.gradsum[idx] = 0.0;
  }
  __threadfence_block();
  __syncthreads();
}

}  // namespace psmf
