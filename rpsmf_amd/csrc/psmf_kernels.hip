// CDNA4 (gfx950) kernels of the per-timestep PSMF / rPSMF filter, large-d engine.
//
// One filter step = two launches (captured back to back in a hipGraph by the host):
//
//   psmf_sweep_solve   n_sweep_wg row blocks: ONE pass over the rows of C.  With the Gram
//                      matrix G = C^T C tracked algebraically in r x r space, N_k and
//                      w_k = V mu_bar are known before the pass, so the rank-1 update
//                      C <- C + e w^T / N is applied in the same pass that forms
//                      y_hat = C mu_bar and e = y - y_hat: C is read once and
//                      written once per step, y read once, y_hat written once
//                      = 8 d (r+1) algorithmic bytes in fp32.  d->r contractions
//                      (h = C^T e, ee = e^T e) are accumulated in float64 per lane, reduced
//                      with wavefront shuffles, then through LDS, one partial per workgroup.
//                      block 0 (only if coef_update; sweep blocks then start at 1): the r x r solve
//                      P+ = (I + kappa Pbar G)^-1 Pbar of the SAME step, concurrently with the
//                      sweep (it needs G_{k-1}, not the sweep's output) -- two symmetric SWEEP
//                      inversions in float64, matrix in registers, one barrier per pivot.
//   psmf_serial        one workgroup: deterministic reduction of the partials, Kalman mean
//                      update, V / P / G / Q / rho / lambda updates, theta gradient (+ Adam in
//                      recursive mode), then everything O(r^2) the NEXT sweep needs
//                      (mu_bar, Pbar, w, s, eta, N, kappa).
//
// Reference equations: pypsmf/psmf/psmf.py:104-177, rpsmf.py:116-184 (see SURVEY App. A for
// the d x d -> r x r reduction).  fp32 (or fp64) storage for C, y, y_hat; everything r-sized
// is float64.
#pragma once
#include "psmf_device.h"
#include "psmf_dyn.hip"

namespace psmf {

// In-kernel stamps for tools/serial_prof.hip (diagnostic builds define PSMF_SERIAL_STAMPS); no-ops in the product.
#ifdef PSMF_SERIAL_STAMPS
#define PSMF_STAMP(n) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(p.partials)[4096 + (n)] = t_; } while (0)
#else
#define PSMF_STAMP(n) do { } while (0)
#endif

template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef float __attribute__((ext_vector_type(4))) type; };
template <> struct VecOf<double> { typedef double __attribute__((ext_vector_type(2))) type; };

// start of a run: step counter.  The numeric-error flag is NOT cleared here: runs may be queued back to back without a
// sync in between, and a failure in an earlier one must still be reported by the next psmf_sync (cleared by psmf_set_state).
__global__ void psmf_prepare_k(DevState* st, long long k) { st->k = k; st->kq = k; st->ticket = 0u; if (st->ns_valid == 7) st->ns_valid = 0; }   // (7: the per-step engine's carried Lbar -- a new run re-derives it)
// end of a run: the numeric-error flag to mapped host memory (system-scope store)
__global__ void psmf_publish_err_k(const DevState* st, int* host_flag) { __hip_atomic_store(host_flag, st->err, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// (wave_sum, fast_rcp: psmf_device.h -- shared with the persistent per-step kernel's translation unit)


// fixed-order sum of base[w * ps] for w = first, first + step, ... < n, 16 independent loads in flight.
// The loads are UNCONDITIONAL (index clamped, value masked afterwards): a load under a runtime
// predicate makes hipcc branch around it and wait for it alone -- 16 dependent L2 round trips.
__device__ __forceinline__ double strided_sum(const double* base, int first, int step, int n, int ps) {
  double acc = 0.0;
  for (int w0 = first; w0 < n; w0 += 16 * step) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = base[(size_t)min(w0 + q * step, n - 1) * ps];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = (w0 + q * step < n) ? v[q] : 0.0;
    acc += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
           (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
  }
  return acc;
}

// ------------------------------------------------------------------------------------------
// r x r solve block:  Pplus = (Pbar^-1 + kappa G)^-1  -- the reference's own formulation
// (inv(P_bar), then inv(Pi + C^T Ri C): pypsmf/psmf/psmf.py:147-149, ExperimentImpute/PSMF.py:34-36)
// as two symmetric SWEEP passes in float64.  Sweeping pivot k of a symmetric matrix A,
//     a_ij <- a_ij - a_ik a_kj / a_kk   (i, j != k),   a_ik = a_ki <- a_ik / a_kk,   a_kk <- -1 / a_kk,
// for all k turns A into -A^-1; for SPD A every pivot is positive, no pivot search is needed and
// the matrix stays (bitwise) symmetric, so one pivot ROW per step is all the waves exchange:
// matrix in registers (thread = column c, rows rg + m * RG), pivot row through a ping-pong LDS
// line, one barrier per pivot.  A non-positive pivot raises the numeric-error flag (the
// reference raises LinAlgError from np.linalg.inv in the same situation).
// ------------------------------------------------------------------------------------------
// LDS_ONLY: the barriers order LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier) instead of __syncthreads(), which also
// drains vmcnt -- for callers that keep global loads / stores in flight across the solve (psmf_impute.hip)
template <bool LDS_ONLY>
__device__ __forceinline__ void solve_barrier() {
  if (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

template <int RPAD, bool LDS_ONLY = false>
__device__ __forceinline__ void sweep_all(double (&A)[(RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1], const int r2,
                                          const int c, const int rg, double* rowbuf, int* errflag) {
  // (a 512-thread workgroup may run two independent sweeps in lockstep, one per 256-thread half, each
  //  with its own rowbuf: the wave id is taken modulo 4 and the barriers are shared)
  constexpr int RG = WG / RPAD;                 // row groups; wave w holds row groups [w*RGW, (w+1)*RGW)
  constexpr int RGW = RG / 4 > 0 ? RG / 4 : 1;  // (RPAD = 64: one row group per wave)
  constexpr int M = (RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1;
  // Measured on MI355X (tools/solve_prof.hip): one LDS publish -> barrier -> LDS read exchange costs
  // ~450 cycles whatever is exchanged, the arithmetic of a pivot ~250.  So pivots are taken as 2 x 2
  // SPD blocks: two pivot rows per exchange, the 2 x 2 inverse recomputed by every thread.
  //   K = [[a, b], [b, e]] = rows/cols (k, k+1);  Ki = K^-1 = [[p, q], [q, s]]
  //   a_ic <- a_ic - [u_i w_i] Ki [u_c w_c]^T            (i, c outside the block; u = row k, w = row k+1)
  //   rows k, k+1 <- Ki [u_c; w_c]    columns k, k+1 <- the same by symmetry    block <- -Ki
  // r2 = r rounded up to even (the caller pads with an identity row/column).
  const int wv = (threadIdx.x >> 6) & 3;
  const bool con = c < r2;
  const int cc = con ? c : r2 - 1;
  int ic[M];
#pragma unroll
  for (int m = 0; m < M; ++m) ic[m] = min(rg + m * RG, r2 - 1);
  // rows 0 and 1 live in slot m = 0 of row groups 0 and 1 (RG >= 4)
  if (rg < 2 && con) rowbuf[rg * RM + c] = A[0];
  solve_barrier<LDS_ONLY>();
  bool bad = false;
  for (int k = 0; k < r2; k += 2) {
    const double* rb0 = rowbuf + ((k >> 1) & 1) * 2 * RM;
    const double* rb1 = rb0 + RM;
    double* rn0 = rowbuf + (((k >> 1) + 1) & 1) * 2 * RM;
    double* rn1 = rn0 + RM;
    const double ka = rb0[k], kb = rb0[k + 1], ke = rb1[k + 1];
    const double uc = rb0[cc], wc = rb1[cc];
    double ui[M], wi[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      ui[m] = rb0[ic[m]];
      wi[m] = rb1[ic[m]];
    }
    const double det = ka * ke - kb * kb;
    bad |= !(ka > 0.0) | !(det > 0.0);
    const double dinv = fast_rcp(det);
    const double kp = ke * dinv, kq = -kb * dinv, ks = ka * dinv;
    const bool c0 = (c == k), c1 = (c == k + 1);
    // coefficients of this thread's column: generic  t = Ki [u_c; w_c];  pivot columns: -row of Ki, no a_ic term
    double t1 = kp * uc + kq * wc;
    double t2 = kq * uc + ks * wc;
    const double keep = (c0 | c1) ? 0.0 : 1.0;
    const double g1 = c0 ? -kp : (c1 ? -kq : t1);
    const double g2 = c0 ? -kq : (c1 ? -ks : t2);
#pragma unroll
    for (int m = 0; m < M; ++m) A[m] = fma(-wi[m], g2, fma(-ui[m], g1, keep * A[m]));
    // pivot rows: a_kc <- t1, a_(k+1)c <- t2; inside the block <- -Ki   (only in the waves that hold them)
    if (((k % RG) / RGW) == wv || (((k + 1) % RG) / RGW) == wv) {
      const double r0v = c0 ? -kp : (c1 ? -kq : t1);
      const double r1v = c0 ? -kq : (c1 ? -ks : t2);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i = rg + m * RG;
        A[m] = (i == k) ? r0v : ((i == k + 1) ? r1v : A[m]);
      }
    }
    // next two pivot rows -> LDS
    if (k + 2 < r2) {
      if ((((k + 2) % RG) / RGW) == wv || (((k + 3) % RG) / RGW) == wv) {
        double nx = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const int i = rg + m * RG;
          nx = (i == k + 2 || i == k + 3) ? A[m] : nx;
        }
        if (con && ((k + 2) % RG) == rg) rn0[c] = nx;
        if (con && ((k + 3) % RG) == rg) rn1[c] = nx;
      }
    }
    solve_barrier<LDS_ONLY>();
  }
  if (bad && (threadIdx.x & (WG - 1)) == 0) *errflag = 1;
}

// A (in): symmetric Pbar elements of this thread, identity-padded to r2;  Gk (in): kappa * G elements
// (0 in the padding).  A (out): elements of (Pbar^-1 + kappa G)^-1.  rowbuf: 4 * RM doubles of LDS.
template <int RPAD, bool LDS_ONLY = false>
__device__ __forceinline__ void spd_update_solve(double (&A)[(RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1],
                                                 const double (&Gk)[(RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1],
                                                 const int r2, const int c, const int rg, double* rowbuf, int* errflag) {
  constexpr int M = (RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1;
  sweep_all<RPAD, LDS_ONLY>(A, r2, c, rg, rowbuf, errflag);       // A = -Pbar^-1
#pragma unroll
  for (int m = 0; m < M; ++m) A[m] = Gk[m] - A[m];
  sweep_all<RPAD, LDS_ONLY>(A, r2, c, rg, rowbuf, errflag);       // A = -(Pbar^-1 + kappa G)^-1
#pragma unroll
  for (int m = 0; m < M; ++m) A[m] = -A[m];
}

template <int RPAD>
__device__ __forceinline__ void solve_block_t(const StepParams& p, double* sm) {
  constexpr int RG = WG / RPAD;
  constexpr int M = (RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1;
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x, c = tid % RPAD, rg = tid / RPAD;
  const int r2 = r + (r & 1);                            // identity padding to an even size
  double* rowbuf = sm;                                   // [2][2][RM]
  int* errflag = reinterpret_cast<int*>(sm + 4 * RM);
  if (tid == 0) *errflag = 0;
  const double kappa = st->kappa;
  const double* __restrict__ gsrc = p.rho_rows ? st->GR : (p.mask ? p.mg : st->G);
  const double gsc = p.rho_rows ? 1.0 : kappa;
  double A[M], Gk[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i = rg + m * RG;
    const bool in = (i < r && c < r);
    const double pv = in ? 0.5 * (st->Pbar[i * r + c] + st->Pbar[c * r + i]) : 0.0;
    A[m] = in ? pv : ((i == c && i < r2) ? 1.0 : 0.0);
    // non-uniform R: sum_i c_i c_i^T / (rho_i + s), this step's; masked step: the step's reduced Gram p.mg (both triangles averaged)
    double gv = gsrc[in ? i * r + c : 0];
    if (p.mask) gv = 0.5 * (gv + gsrc[in ? c * r + i : 0]);
    Gk[m] = in ? gv * gsc : 0.0;
  }
  spd_update_solve<RPAD>(A, Gk, r2, c, rg, rowbuf, errflag);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i = rg + m * RG;
    if (i < r && c < r) st->Pplus[i * r + c] = A[m];    // symmetrised by the consumer (serial stage)
  }
  if (tid == 0 && *errflag && st->err == 0) st->err = (int)(st->k + 1);
}

// r <= 32: wave-local sweeps on the matrix cores, no LDS, no barrier -- and, for the random walk with Q = q I, the two inversions
// side by side on two waves (psmf_wave16.hip: solve_block_wave, defined after the helpers it needs; same translation unit)
__device__ void solve_block_wave(const StepParams& p);
__device__ __forceinline__ void solve_block_wave_big(const StepParams& p);      // 33 <= r <= 64 (psmf_wave16.hip)

template <bool BIGWAVE>      // the kernel instance is the 256-thread one of r > 32: wave-local tile sweeps there too
__device__ __forceinline__ void solve_block(const StepParams& p, double* sm) {
  const int r = p.r;
  if (r <= 32 && !p.solve_lds) {
    solve_block_wave(p);
    return;
  }
  if constexpr (BIGWAVE) {
    if (r > 32 && !p.solve_lds) {
      solve_block_wave_big(p);
      return;
    }
  }
  if (blockDim.x > WG && threadIdx.x >= WG) return;   // the LDS sweep uses 4 waves; surplus waves retire (barriers count live waves)
  if (r <= 8) solve_block_t<8>(p, sm);
  else if (r <= 16) solve_block_t<16>(p, sm);
  else if (r <= 32) solve_block_t<32>(p, sm);
  else solve_block_t<64>(p, sm);
}

// ------------------------------------------------------------------------------------------
// Masked step (psmf_masked.hip), start of every workgroup of the sweep: eta, N of the step from the reduced masked Gram
//   eta = (rho n_obs + <G_m, Pbar>) / d   (divided by d, ExperimentImpute/PSMF.py:77),   N = s + eta
// p.mg[0 .. r*r) = G_m, p.mg[r*r] = n_obs (summed over workgroups and ranks); p.mg_tr: the shares of <G_m, Pbar>.  Every thread forms
// the same sum in the same order (same bits everywhere); `publish` (block 0, before its solve): eta, N, kappa and the step's (s, eta)
// for the bands.  Returns in sc[0..2]: eta, N, and the factor in front of the update direction (1 / N for PSMF / rPSMF).
// masked_method 2 (MLE-SMF): weights m_i / rho (s does not enter), direction (gam / eta) mu_bar;  3 (TMF): kappa = 1, gam mu_bar.
// sm: 16 doubles of LDS.  All threads of the workgroup call it.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void masked_prep_block(const StepParams& p, double* sm, const bool publish, double (&sc)[3]) {
  (void)sm;
  DevState* st = p.st;
  const int r = p.r;
  // <G_m, Pbar> arrives as p.mg_ntr shares (psmf_mgram_reduce / psmf_mgram_trace): every thread sums them in order -- a handful of
  // broadcast loads in the same round trip as the other r-sized operands of the sweep, no pass over G, no barrier
  double tr = 0.0;
  for (int w = 0; w < p.mg_ntr; ++w) tr += p.mg_tr[w];
  const int meth = p.masked_method;
  const double s = meth ? 0.0 : st->s, rho = st->rho;
  const double eta = (rho * p.mg[r * r] + tr) / (double)p.d;
  const double N = s + eta;
  sc[0] = eta; sc[1] = N;
  sc[2] = meth == 0 ? fast_rcp(N) : (meth == 2 ? st->sgd_gamma * fast_rcp(eta) : st->sgd_gamma);
  if (publish && threadIdx.x == 0) {
    st->eta = eta;
    st->N = N;
    st->kappa = meth == 3 ? 1.0 : fast_rcp(rho + s);
    st->kq = st->k + 1;               // the Gram that the serial-stage launch computes beside the serial stage is the NEXT step's
    if (p.sc_hist) {
      const long long t = st->k - p.series_t0;
      p.sc_hist[2 * t] = s;
      p.sc_hist[2 * t + 1] = eta;
    }
  }
  if (publish) __syncthreads();       // block 0 with the LDS solve (r > 32): eta, kappa are in memory before the solve reads them
}

// ------------------------------------------------------------------------------------------
// StepParams.tail_reduce: the fixed-order sum of the sweep's partial rows by the LAST row workgroup to finish, into st->red,
// instead of by the serial stage (which then starts from r + 1 numbers, external_reduce = 1).  The rows leave their
// workgroups as agent-scope stores, a ticket counts the workgroups that are done (the last one resets it), the last one reads all
// rows with agent-scope loads -- whichever workgroup that is, the order of the sum is the same.  At r > 32 and small d the
// row blocks finish long before the solve block: the reduction is off the step's path entirely; elsewhere it trades the serial
// stage's single-CU fetch of n_sweep_wg rows (9.6 k of 28.8 k cycles at r = 40, tools/serial_prof.hip) for a tail on one
// row workgroup.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void part_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double part_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NT>
__device__ __forceinline__ void tail_reduce_partials(const StepParams& p) {
  __shared__ int s_last;
  __shared__ double s_tail[8][2 * (RM + 1)];
  DevState* st = p.st;
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this workgroup's row is where every XCD sees it
  __syncthreads();
  if (tid == 0) {
    const unsigned tk = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = tk == (unsigned)p.n_sweep_wg - 1u;
    if (last) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const int ne = p.rho_rows ? 2 * (p.r + 1) : p.r + 1, n = p.n_sweep_wg;
  int nseg = NT / ne;
  if (nseg > 8) nseg = 8;
  if (tid < ne * nseg) {
    const int e = tid % ne, sg = tid / ne;
    const double* base = p.partials + e;
    double acc = 0.0;
    for (int w0 = sg; w0 < n; w0 += 16 * nseg) {       // as strided_sum: 16 clamped loads in flight, masked afterwards, fixed tree
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = part_load(base + (size_t)min(w0 + q * nseg, n - 1) * p.ps);
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = (w0 + q * nseg < n) ? v[q] : 0.0;
      acc += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
             (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    }
    s_tail[sg][e] = acc;
  }
  __syncthreads();
  if (tid < ne) {
    double a = 0.0;
    for (int sg = 0; sg < nseg; ++sg) a += s_tail[sg][tid];
    st->red[tid] = a;
  }
}

// ------------------------------------------------------------------------------------------
// Row sweep.  GS = lanes cooperating on one row (power of two >= nv = ceil(r / VEC)); each lane
// owns one 16-byte vector of the row; a 256-thread workgroup covers 256/GS rows per pass and
// keeps U passes of loads in flight.
// ------------------------------------------------------------------------------------------
template <typename T, int GS, int U, int NT>
__global__ __launch_bounds__(NT) void psmf_sweep_solve(StepParams p) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int RPP = NT / GS;
  constexpr int NW = NT / 64;
  typedef typename VecOf<T>::type VT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);

  // block 0 runs the r x r solve (dispatched first: it is the longest block)
  const int has_solve = p.coef_update;
  double msc[3] = {0.0, 0.0, 0.0};
  // masked step: eta, N from the step's Gram -- every row workgroup for itself; block 0 publishes them: by a wave of its own beside
  // the solve waves when the solve is wave-local (r <= 32), else by the whole block before the LDS solve
  const bool solve_here = has_solve && blockIdx.x == 0;
  constexpr bool BIGWAVE = NT == 256 && GS * VEC > 32;         // (r > 32 runs the 256-thread instances only: a wave there may hold 512 registers)
  if (p.mask && !(solve_here && (p.r <= 32 || BIGWAVE) && !p.solve_lds)) masked_prep_block(p, sm, solve_here, msc);
  if (has_solve && blockIdx.x == 0) {
    // (raising the solve waves' issue priority on the CU they share with a row-sweep workgroup was measured neutral: the block
    //  is a chain of LDS exchanges and barriers, not short of issue slots)
    solve_block<BIGWAVE>(p, sm);
    return;
  }
  const int wgid = (int)blockIdx.x - has_solve;
  DevState* st = p.st;
  const int tid = threadIdx.x, j = tid % GS, g = tid / GS;
  const int r = p.r, rp = p.rp;
  const long long t = st->k - p.series_t0;  // row of the series buffer holding y_k
  const T* __restrict__ y = reinterpret_cast<const T*>(p.Y) + (size_t)t * p.d_local;
  T* __restrict__ yp = p.store_yp ? reinterpret_cast<T*>(p.YP) + (size_t)t * p.d_local : nullptr;
  T* __restrict__ C = reinterpret_cast<T*>(p.C);
  // masked step (psmf_masked.hip): e_i = m_i (y_i - y_hat_i); rows with m_i = 0 keep their c_i, y_hat is stored unmasked
  const uint8_t* __restrict__ msk = p.mask ? p.mask + (size_t)t * p.d_local : nullptr;

  // r-sized operands and all row arithmetic in float64; only the storage of C, y, y_hat is T
  double mub[VEC], wn[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int e = j * VEC + v;
    mub[v] = e < r ? st->mu_bar[e] : 0.0;
    wn[v] = e < r ? st->wN[e] : 0.0;
    if (p.mask) wn[v] = e < r ? (p.masked_method == 0 ? st->w[e] : st->mu_bar[e]) * msc[2] : 0.0;      // direction of the rank-1 update of C
  }
  double hacc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) hacc[v] = 0.0;
  double eacc = 0.0;
  // non-uniform diagonal R: also the weighted sums b = sum_i kappa_i c_i e_i, q = sum_i kappa_i e_i^2, kappa_i = 1 / (rho_i + s)
  // (psmf.py:140-159 with a diagonal R: SURVEY App. A); rows of partials then hold 2 (r + 1) values
  const double* __restrict__ rrow = p.rho_rows;
  const double rsc = st->rho, sk = st->s;
  double hacc2[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) hacc2[v] = 0.0;
  double eacc2 = 0.0;

  const int row_begin = wgid * p.rows_per_wg;
  const int row_end = min(row_begin + p.rows_per_wg, p.d_local);
  const bool lane_on = j < p.nv;
  const int jl = lane_on ? j : 0;

  for (int base = row_begin; base < row_end; base += RPP * U) {
    VT cv[U];
    T yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // unconditional loads from a clamped (valid) row / lane: a load under a runtime predicate is
      // branched around and waited for on its own; the value is masked below instead
      const int row = min(base + u * RPP + g, row_end - 1);
      cv[u] = *reinterpret_cast<const VT*>(C + (size_t)row * rp + jl * VEC);
      yv[u] = y[row];
    }
    double kr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kr[u] = rrow ? rrow[min(base + u * RPP + g, row_end - 1)] : 0.0;
    bool mo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) mo[u] = msk ? msk[min(base + u * RPP + g, row_end - 1)] != 0 : true;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * RPP + g;
      const bool ok = row < row_end;
      double cd[VEC];
      double dot = 0.0;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        cd[v] = (ok && lane_on) ? (double)cv[u][v] : 0.0;
        dot += cd[v] * mub[v];
      }
#pragma unroll
      for (int m = GS / 2; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
      const double e = (ok && mo[u]) ? (double)yv[u] - dot : 0.0;
      VT cn;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        cn[v] = (T)(cd[v] + e * wn[v]);      // one rounding to the storage type per step
        hacc[v] += cd[v] * e;
      }
      if (rrow) {
        const double ke = e * fast_rcp(rsc * kr[u] + sk);     // kappa_i e_i  (0 on rows beyond the end: e = 0)
#pragma unroll
        for (int v = 0; v < VEC; ++v) hacc2[v] += cd[v] * ke;
        if (j == 0) eacc2 += e * ke;
      }
      if (ok && lane_on) *reinterpret_cast<VT*>(C + (size_t)row * rp + j * VEC) = cn;
      if (ok && j == 0) {
        if (yp) yp[row] = (T)dot;
        eacc += e * e;
      }
    }
  }

  // lanes with equal j inside a wave, then the 4 waves through LDS
#pragma unroll
  for (int m = 32; m >= GS; m >>= 1) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) hacc[v] += __shfl_xor(hacc[v], m, 64);
    eacc += __shfl_xor(eacc, m, 64);
  }
  constexpr int NE = GS * VEC;  // >= rp
  const int lane = tid & 63, wv = tid >> 6;
  if (lane < GS) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) sm[wv * (NE + 1) + lane * VEC + v] = hacc[v];
    if (lane == 0) sm[wv * (NE + 1) + NE] = eacc;
  }
  __syncthreads();
  double* out = p.partials + (size_t)wgid * p.ps;
  if (tid <= NE) {
    double s4 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s4 += sm[w * (NE + 1) + tid];   // fixed order
    if (tid < r) part_store(out + tid, s4);
    if (tid == NE) part_store(out + r, s4);
  }
  if (rrow) {          // the weighted sums the same way (second half of the partial row)
    __syncthreads();
#pragma unroll
    for (int m = 32; m >= GS; m >>= 1) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) hacc2[v] += __shfl_xor(hacc2[v], m, 64);
      eacc2 += __shfl_xor(eacc2, m, 64);
    }
    if (lane < GS) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) sm[wv * (NE + 1) + lane * VEC + v] = hacc2[v];
      if (lane == 0) sm[wv * (NE + 1) + NE] = eacc2;
    }
    __syncthreads();
    if (tid <= NE) {
      double s4 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s4 += sm[w * (NE + 1) + tid];
      if (tid < r) part_store(out + r + 1 + tid, s4);
      if (tid == NE) part_store(out + 2 * r + 1, s4);
    }
  }
  if (p.tail_reduce) tail_reduce_partials<NT>(p);
}

// ------------------------------------------------------------------------------------------
// Serial stage: one workgroup of SWG = 1024 threads.  All 16 waves take part in the
// deterministic (fixed-order) reduction of the sweep partials -- each thread sums <= 16 terms with
// every load in flight at once, i.e. ONE memory round trip -- then waves 4..15 retire and the four
// "worker" waves do the r x r work.  RPAD = power of two >= max(r, 8).  Worker thread tid owns
// column j = tid % RPAD and rows i_m = tid / RPAD + m * (256 / RPAD) of every r x r matrix, in
// registers.  All matrices are symmetric, so a matrix-vector product is a per-thread partial
// plus one LDS column reduce.  Every global load of the stage is issued before the first
// barrier (the stage is latency-bound: what matters is the number of dependent round trips).
// ------------------------------------------------------------------------------------------
constexpr int SWG = 1024;
// r > 16: the worker threads keep 4 matrices x (4..16) elements in registers -- 8 waves (256 VGPRs each) instead of 16 (128: spills);
// r > 32 (RPAD = 64): psmf_serial_wide -- 512 WORKERS, 8 elements of each matrix per thread (12.9 us per timestep).  psmf_serial<64>
// (256 workers x 16 elements, a whole SIMD's register file per wave: 20 us, 18.5 k of its 41.5 k cycles spent issuing loads) remains
// behind PSMF_SERIAL_WIDE=0 and as block 0 of psmf_serial_mgram<64, ...>; with 512 threads and 256 workers the stage had spilled 196
// registers and taken 33 us (profiles/r4_step_engine.txt)
__host__ __device__ constexpr int serial_threads(int rpad) { return rpad >= 64 ? WG : (rpad >= 32 ? 512 : SWG); }

template <int RPAD, int NWK = WG>
__device__ __forceinline__ void col_reduce(double partial, double* s_red, double* s_out) {
  constexpr int RG = NWK / RPAD;
  const int tid = threadIdx.x;
  s_red[tid] = partial;  // index = (tid / RPAD) * RPAD + tid % RPAD
  __syncthreads();
  if (tid < RPAD) {
    double a = 0.0;
#pragma unroll
    for (int gI = 0; gI < RG; ++gI) a += s_red[gI * RPAD + tid];
    s_out[tid] = a;
  }
  __syncthreads();
}

#ifndef PSMF_SERIAL_BUTTERFLY
#define PSMF_SERIAL_BUTTERFLY 1
#endif
constexpr bool SERIAL_BUTTERFLY = PSMF_SERIAL_BUTTERFLY != 0;      // the r-long dot products of the serial stage as wave butterflies (0: every thread sums RPAD products from LDS)

template <int NWK = WG>
__device__ __forceinline__ double block_sum(double x, double* s4) {   // worker waves only
  x = wave_sum(x);
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = x;
  __syncthreads();
  double a = 0.0;
#pragma unroll
  for (int g = 0; g < NWK / 64; g += 4) a += (s4[g] + s4[g + 1]) + (s4[g + 2] + s4[g + 3]);      // fixed order
  return a;
}

// NWK = worker threads (the r x r work): 256, or SERIAL_WIDE_NT = 512 for RPAD = 64 (psmf_serial_wide: 8 elements of every matrix
// per thread instead of 16; 1 024 workers x 4 elements spill under the 128-register cap: 32.7 us per timestep at r = 40 against 29.9)
template <int RPAD, int NWK = WG>
__device__ __forceinline__ void serial_body(const StepParams& p, const int first) {
  constexpr int RG = NWK / RPAD;
  constexpr int M = (RPAD * RPAD) / NWK > 0 ? (RPAD * RPAD) / NWK : 1;
  constexpr int NSEG = 32;
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x;
  const bool worker = tid < NWK;
  const int j = tid % RPAD, ig = (tid % NWK) / RPAD;
  const double dd = (double)p.d;

  __shared__ double s_red[NWK];
  __shared__ double s_he[2 * (RM + 1)];   // h[0..r), ee at [r]; non-uniform R: b[r+1 .. 2r], q at [2r+1]
  __shared__ double s_w[RM], s_mub[RM], s_f[RM], s_vec[RM];
  __shared__ double s_part[NSEG][2 * (RM + 1)];
  __shared__ double s4[NWK / 64];
  // dynamics with a matrix in them (scaled walk, sinusoid, Fourier basis: psmf_dyn.hip): mu_{k-1} / mu_k, g_f, the terms' trig values
  __shared__ double s_dx[RM], s_gf[RM];
  __shared__ double s_val[DYN_MAX_TERMS * RM], s_tp[DYN_MAX_TERMS * RM], s_dpart[DYN_MAX_TERMS * RM];

  PSMF_STAMP(0);
  // ---------------- every global load of the stage, issued up front ----------------
  const bool wR = p.rho_rows != nullptr;     // non-uniform diagonal R: the partial rows carry the weighted sums too
  const int ne = wR ? 2 * (r + 1) : r + 1;
  int nseg = (int)blockDim.x / ne;
  if (nseg > NSEG) nseg = NSEG;
  const int pe = tid % ne, psg = tid / ne;
  double psum = 0.0;
  if (!first) {
    if (p.external_reduce) {
      psum = st->red[min(tid, ne - 1)];
    } else {
      psum = strided_sum(p.partials + pe, min(psg, nseg - 1), nseg, p.n_sweep_wg, p.ps);
    }
  }
  double Vv[M], Pv[M], Gv[M], Qv[M];
  bool val[M];
  int ii[M];
  const double* psrc = first ? st->P : (p.coef_update ? st->Pplus : st->Pbar);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    ii[m] = ig + m * RG;
    val[m] = worker && (j < r) && (ii[m] < r);
    const int idx = val[m] ? ii[m] * r + j : 0;
    const double lv = st->V[idx], lg = st->G[idx], lq = st->Q[idx], lp = psrc[idx];   // unconditional
    Vv[m] = val[m] ? lv : 0.0;
    Gv[m] = val[m] ? lg : 0.0;
    Qv[m] = val[m] ? lq : 0.0;
    Pv[m] = lp;           // as stored; symmetrised through LDS below
  }
  // inversions side by side (solve_dual): W of this step -> Lbar of the next (written with the r x r updates below)
  const bool dual = p.solve_dual && !first;
  double Wv[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int idx = val[m] ? ii[m] * r + j : 0;
    Wv[m] = st->XpY[idx];
  }
  const double q_old = st->Q[0];
  double rho = st->rho, lam = st->lam;
  const double N0 = st->N, kappa0 = st->kappa, s0 = st->s, eta0 = st->eta;
  const long long k0 = st->k;       // read once (thread 0 rewrites it below)
  long long knext = k0;             // index (0-based) of the step to prepare
  const bool vl = tid < r;
  // the per-step engine evaluates the random walk and cos(2 pi theta t + x) itself (n_theta = r); every other f is
  // host-stepped here (dyn_kind 5: mu_bar, P_bar come from the host, g_f goes back) or runs in the blocked engine
  const bool host_dyn = p.dyn_kind == 5;
  const bool gen_dyn = p.dyn_kind >= DYN_SCALED_WALK && p.dyn_kind <= DYN_FOURIER;      // evaluated through psmf_dyn.hip (uniform)
  const bool tl = tid < p.n_theta && p.dyn_kind == 1;
  const int tc = tid & (RM - 1);   // every r-sized array has RM entries: load unconditionally, mask afterwards
  const double l_mu = st->mu[tc], l_w = st->w[tc], l_mub = st->mu_bar[tc], l_th = p.theta[tc], l_gs = p.gradsum[tc],
               l_am = p.adam_m[tc], l_av = p.adam_v[tc];
  double mu_new = vl ? l_mu : 0.0;
  const double mu_old = mu_new;
  const double w_t = vl ? l_w : 0.0;
  const double mub_t = vl ? l_mub : 0.0;
  double theta = tl ? l_th : 0.0;
  double gsum = tl ? l_gs : 0.0;
  double am = tl ? l_am : 0.0;
  double av = tl ? l_av : 0.0;

  asm volatile("" :: "v"(psum), "v"(Vv[0]), "v"(Pv[0]));
  PSMF_STAMP(1);
  // P+ (P at a run's start) and W come from the solve block symmetric only up to round-off: (X + X^T) / 2, the transpose taken through
  // LDS -- s_part's memory, free until the partial rows are staged below.  (As transposed GLOBAL loads -- 64 cache lines per wave
  // instruction -- they were 8 000 of the 18 000 cycles the stage spent on its loads at RPAD = 64, tools/serial_prof.hip.)
  {
    constexpr int TS = RPAD + 1;
    static_assert(RPAD * TS <= NSEG * 2 * (RM + 1), "transposition buffer aliases s_part");
    double* s_T = &s_part[0][0];
#pragma unroll
    for (int m = 0; m < M; ++m) if (val[m]) s_T[ii[m] * TS + j] = Pv[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) Pv[m] = val[m] ? 0.5 * (Pv[m] + s_T[j * TS + ii[m]]) : 0.0;
    if (dual) {        // uniform
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; ++m) if (val[m]) s_T[ii[m] * TS + j] = Wv[m];
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; ++m) Wv[m] = val[m] ? 0.5 * (Wv[m] + s_T[j * TS + ii[m]]) : 0.0;
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) Wv[m] = 0.0;
    }
    __syncthreads();
  }
  if (tid < 2 * (RM + 1)) s_he[tid] = 0.0;
  if (tid < RM) { s_mub[tid] = 0.0; s_w[tid] = 0.0; }
  if (!first) {
    // ---- fixed-order reduction of the per-workgroup partials ----
    if (!p.external_reduce) {
      if (tid < ne * nseg) s_part[psg][pe] = psum;
      __syncthreads();
      if (!worker) return;          // helper waves retire; later barriers count the 4 worker waves only
      if (tid < ne) {
        double a = 0.0;      // fixed order; 8 LDS reads in flight per batch, clamped rows, masked afterwards
#pragma unroll
        for (int b8 = 0; b8 < NSEG; b8 += 8) {
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = s_part[min(b8 + q, nseg - 1)][tid];
#pragma unroll
          for (int q = 0; q < 8; ++q) a += (b8 + q < nseg) ? v[q] : 0.0;
        }
        s_he[tid] = a;
      }
    } else {
      if (!worker) return;
      if (tid < ne) s_he[tid] = psum;
    }
    if (vl) {
      s_w[tid] = w_t;
      s_mub[tid] = mub_t;
    }
    __syncthreads();
    PSMF_STAMP(2);
    const double N = N0, kappa = kappa0;
    const double ee = s_he[r];
    const double wj = j < r ? s_w[j] : 0.0;

    // ---- coefficient mean / covariance   psmf.py:155-165 ----
    // uniform R: b = kappa h, q = kappa ee; non-uniform diagonal R: the sweep's weighted sums
    const double* s_b = wR ? s_he + (r + 1) : s_he;
    const double bsc = wR ? 1.0 : kappa;
    double quad = wR ? s_he[2 * r + 1] : kappa * ee;
    if (p.coef_update) {
      double part = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) part += val[m] ? Pv[m] * s_b[min(ii[m], r - 1)] : 0.0;
      col_reduce<RPAD, NWK>(part, s_red, s_vec);   // s_vec = Pplus b / bsc
      double bPb = 0.0;
      if constexpr (SERIAL_BUTTERFLY) {       // r <= 64 = one wave's width: a butterfly instead of 2 RPAD LDS reads per thread
        static_assert(NWK == WG || RPAD == 64, "wide serial stage: RPAD = 64");
        const int l = tid & 63;
        bPb = wave_sum(l < r ? s_b[l] * s_vec[l] : 0.0);       // (s_vec is written up to RPAD only: select the product, do not multiply by 0)
      } else {
#pragma unroll
        for (int l = 0; l < RPAD; ++l) bPb += (l < r ? s_b[l] : 0.0) * s_vec[l];   // s_vec[l >= r] = 0
      }
      quad -= bsc * bsc * bPb;
      if (vl) mu_new = mub_t + bsc * s_vec[tid];
    } else {
      if (vl) mu_new = mub_t;
    }

    PSMF_STAMP(3);
    // ---- theta gradient at the pre-update state   psmf.py:48-66,167-177; rpsmf.py:53-73 ----
    if (tl && p.dyn_kind == 1) {
      const double tk = (double)(k0 + 1);
      const double arg = 2.0 * M_PI * theta * tk + mu_old;
      const double jt = -sin(arg) * (2.0 * M_PI * tk);
      const double wi = w_t, hi = s_he[tid];
      double gf;
      if (p.robust) {
        const double D = lam * N;
        gf = dd * wi / N + 0.5 * (dd + lam) * (-2.0 * hi / D - 2.0 * lam * ee * wi / (D * D)) / (1.0 + ee / D);
      } else {
        gf = dd * wi / N - hi / N - ee * wi / (N * N);
      }
      gsum += jt * gf;
    }
    if (gen_dyn && p.n_theta > 0) {      // gradsum += J_theta^T g_f through the terms' trig values at x = mu_{k-1} (nonlinearities.py:59-150)
      const double tk = (double)(k0 + 1);
      if (vl) {
        const double wi = w_t, hi = s_he[tid];
        double gf;
        if (p.robust) {
          const double D = lam * N;
          gf = dd * wi / N + 0.5 * (dd + lam) * (-2.0 * hi / D - 2.0 * lam * ee * wi / (D * D)) / (1.0 + ee / D);
        } else {
          gf = dd * wi / N - hi / N - ee * wi / (N * N);
        }
        s_gf[tid] = gf;
        s_dx[tid] = mu_old;
      }
      __syncthreads();
      dyn_trig<NWK>(p, tk, s_dx, s_val, s_tp, tid);
      dyn_backward<NWK>(p, tk, s_dx, s_gf, s_val, s_tp, tid);
    }
    if (host_dyn && vl) {     // g_f for the host, which holds J_theta (same closed forms)
      const double wi = w_t, hi = s_he[tid];
      if (p.robust) {
        const double D = lam * N;
        st->gf[tid] = dd * wi / N + 0.5 * (dd + lam) * (-2.0 * hi / D - 2.0 * lam * ee * wi / (D * D)) / (1.0 + ee / D);
      } else {
        st->gf[tid] = dd * wi / N - hi / N - ee * wi / (N * N);
      }
    }

    // ---- robust scalars   rpsmf.py:133-171 ----
    double vscale = 1.0, pscale = 1.0, qscale = 1.0, phi = 1.0, omega = 1.0;
    const double invN = fast_rcp(N);
    if (p.robust) {
      const double ild = fast_rcp(lam + dd);
      phi = (lam + ee * invN) * ild;
      omega = (lam + quad) * ild;
      vscale = p.alpha * phi;
      if (p.coef_update) { pscale = p.beta * omega; qscale = omega; }
      rho *= omega;
      if (!p.fixed_lambda) lam += dd;
    }

    PSMF_STAMP(4);
    // ---- r x r elementwise updates (V, P, Q, tracked Gram) ----
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (val[m]) {
        const int idx = ii[m] * r + j;
        const double wi = s_w[ii[m]], hi = s_he[ii[m]], hj = s_he[j];
        Vv[m] = vscale * (Vv[m] - wi * wj * invN);
        Pv[m] *= pscale;
        if (p.masked_method == 3) Pv[m] = 0.0;          // TMF: Pbar of every step is Q = I / nu (TMF.py:47,60), nothing is carried
        st->V[idx] = Vv[m];
        st->P[idx] = Pv[m];
        if (p.track_g) {
          Gv[m] += (hi * wj + wi * hj) * invN + ee * (wi * wj) * (invN * invN);
          st->G[idx] = Gv[m];
        }
        if (qscale != 1.0) { Qv[m] *= qscale; st->Q[idx] = Qv[m]; }
        if (dual) {      // Pbar'^-1 = (I / q - W / q^2) / omega  (Pbar' = omega (beta P+ + q I); non-robust: omega = beta = 1)
          const double iq = 1.0 / q_old;
          st->Lbar[idx] = ((ii[m] == j ? iq : 0.0) - Wv[m] * iq * iq) / omega;
        }
      }
    }
    knext = k0 + 1;

    // ---- Adam on theta inside the time loop (PSMFRecursive, psmf.py:299-304,224-242) ----
    if (tl) {
      if (p.recursive && (knext % p.update_every) == 0) {
        const double kk = (double)knext;
        const double lr = p.lr_steps > 0.0 ? p.lr * pow(p.lr_end / p.lr, kk / p.lr_steps) : p.lr;
        if (p.recursive == 2) {           // plain SGD (psmf.py:244-248)
          theta = fmax(theta - lr * gsum, 0.0);
        } else {
          am = p.b1 * am + (1.0 - p.b1) * gsum;
          av = p.b2 * av + (1.0 - p.b2) * gsum * gsum;
          p.adam_m[tid] = am;
          p.adam_v[tid] = av;
          const double mh = am / (1.0 - pow(p.b1, kk));
          const double vh = av / (1.0 - pow(p.b2, kk));
          theta = fmax(theta - lr * mh / (sqrt(vh) + 1e-8), 0.0);
        }
        p.theta[tid] = theta;
        gsum = 0.0;
      }
      p.gradsum[tid] = gsum;
    }
    if (vl) {
      st->mu[tid] = mu_new;
      if (p.mu_hist) p.mu_hist[(size_t)(knext - p.series_t0) * r + tid] = mu_new;
    }
    if (tid == 0) {
      st->k = knext;
      st->ns_valid = dual ? 7 : 0;
      st->rho = rho;
      st->lam = lam;
      st->phi = phi;
      st->omega = omega;
      st->ee = ee;
      st->s_done = s0;
      st->eta_done = eta0;
      st->N_done = N;
    }
    if (host_dyn) return;        // the host evaluates f for the next step, then launches this stage with first = 1
    // PSMFRecursive with these dynamics: the optimiser step on theta every update_every observations (psmf.py:299-304)
    if (gen_dyn && p.recursive && p.n_theta > 0 && (knext % p.update_every) == 0) dyn_adam_step<NWK>(p, knext, tid);
  } else {
    if (!worker) return;
  }

  PSMF_STAMP(5);
  // =============== everything the NEXT sweep / solve needs (step index knext + 1) ===========
  // PSMFIter reads Q[k], R[k] of the step itself (psmf.py:115,123,141): scalar schedules (never with rPSMF's running Q, R)
  const double qs = p.q_sched ? p.q_sched[knext + 1 - p.series_t0] : 1.0;
  if (p.rho_sched) rho = p.rho_sched[knext + 1 - p.series_t0];
  const double* qm = p.q_mat ? p.q_mat + (size_t)(knext + 1 - p.series_t0) * r * r : nullptr;     // Q_k as a matrix of its own
  double* sF = &s_part[0][0];          // dense F = df/dx, row stride RPAD + 1 (s_part is free since the reduction)
  constexpr int FS = RPAD + 1;
  if (gen_dyn) {
    if (vl) s_dx[tid] = mu_new;
    __syncthreads();
    dyn_forward<NWK>(p, (double)(knext + 1), s_dx, s_mub, s_f, sF, FS, s_val, s_tp, s_dpart, tid);     // ends with a barrier
    if (vl) st->mu_bar[tid] = s_mub[tid];
  } else {
    if (vl) {
      double mb = mu_new, f = 1.0;
      if (p.dyn_kind == 1) {   // cos(2 pi theta t + x)
        const double arg = 2.0 * M_PI * theta * (double)(knext + 1) + mu_new;
        mb = cos(arg);
        f = -sin(arg);
      }
      if (host_dyn) mb = mub_t;     // mu_bar of the step as the host uploaded it
      s_mub[tid] = mb;         // (the current step's mu_bar was consumed from registers above)
      s_f[tid] = f;
      st->mu_bar[tid] = mb;
    }
    __syncthreads();
  }
  double Fpf[M];            // (F P F^T)[i][j] of a dense F
  if (gen_dyn && p.pbar_predict && dyn_dense(p.dyn_kind, p.dyn_flags)) {
    // two r x r x r products, one output element per (i_m, j) as everywhere in this stage; P (symmetrised, scaled) and T = F P go
    // through st->Lbar / st->XpY, which are scratch here: the side-by-side inversions that carry them are a random-walk form
    double* Ps = st->Lbar;
    double* Ts = st->XpY;
#pragma unroll
    for (int m = 0; m < M; ++m) if (val[m]) Ps[ii[m] * r + j] = Pv[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (val[m]) {
        double a = 0.0;
        for (int l = 0; l < r; ++l) a += sF[ii[m] * FS + l] * Ps[l * r + j];
        Ts[ii[m] * r + j] = a;
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double a = 0.0;
      if (val[m]) for (int l = 0; l < r; ++l) a += Ts[ii[m] * r + l] * sF[j * FS + l];
      Fpf[m] = a;
    }
  } else {
#pragma unroll
    for (int m = 0; m < M; ++m) Fpf[m] = val[m] ? s_f[ii[m]] * Pv[m] * s_f[j] : 0.0;
  }
  double part = 0.0, gp = 0.0;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (val[m]) {
      double pb = p.pbar_predict ? Fpf[m] + (qm ? qm[ii[m] * r + j] : qs * Qv[m]) : Pv[m];
      if (host_dyn) pb = 0.5 * (st->Pbar[ii[m] * r + j] + st->Pbar[j * r + ii[m]]);   // P_bar = F P F^T + Q as the host formed it
      else st->Pbar[ii[m] * r + j] = pb;
      part += Vv[m] * s_mub[ii[m]];
      gp += Gv[m] * pb;
    }
  }
  PSMF_STAMP(6);
  col_reduce<RPAD, NWK>(part, s_red, s_vec);   // s_vec = V mu_bar
  PSMF_STAMP(7);
  double s = 0.0;
  if constexpr (SERIAL_BUTTERFLY) {
    const int l = tid & 63;
    s = wave_sum(l < r ? s_mub[l] * s_vec[l] : 0.0);
  } else {
#pragma unroll
    for (int l = 0; l < RPAD; ++l) s += s_mub[l] * s_vec[l];       // both are 0 beyond r
  }
  double eta = rho * p.rho_mean;                   // tr(R) / d  (rho_mean = 1 unless R is a non-uniform diagonal)
  if (p.eta_full) eta += block_sum<NWK>(gp, s4) / dd;   // (tr R + <G, Pbar>) / d   psmf.py:121-125
  const double N = s + eta;
  if (vl) {
    st->w[tid] = s_vec[tid];
    st->wN[tid] = s_vec[tid] * fast_rcp(N);
  }
  if (tid == 0) {
    st->s = s;
    st->eta = eta;
    st->N = N;
    st->kappa = fast_rcp(rho + s);
  }
  PSMF_STAMP(8);
}

template <int RPAD>
__global__ __launch_bounds__(serial_threads(RPAD)) void psmf_serial(StepParams p, int first) {
  serial_body<RPAD>(p, first);
}

// r > 32 (RPAD = 64) with every thread of a 512-thread workgroup a worker (PSMF_SERIAL_WIDE=0: psmf_serial<64>, 256 workers)
constexpr int SERIAL_WIDE_NT = 512;
__global__ __launch_bounds__(SERIAL_WIDE_NT) void psmf_serial_wide(StepParams p, int first) {
  serial_body<64, SERIAL_WIDE_NT>(p, first);
}

// local reduction of the per-workgroup partials into st->red (multi-GPU: input of the all-reduce)
__global__ __launch_bounds__(WG) void psmf_reduce_partials(StepParams p) {
  __shared__ double s_part[8][2 * (RM + 1)];
  const int tid = threadIdx.x, ne = p.rho_rows ? 2 * (p.r + 1) : p.r + 1;
  int nseg = WG / ne;
  if (nseg > 8) nseg = 8;
  if (tid < ne * nseg) {
    const int e = tid % ne, sg = tid / ne;
    s_part[sg][e] = strided_sum(p.partials + e, sg, nseg, p.n_sweep_wg, p.ps);
  }
  __syncthreads();
  if (tid < ne) {
    double a = 0.0;
    for (int sg = 0; sg < nseg; ++sg) a += s_part[sg][tid];
    p.st->red[tid] = a;
  }
}

// ------------------------------------------------------------------------------------------
// Exact Gram matrix G = C^T C of the local rows (float64 accumulation), used at set_state and
// at the optional periodic refresh.  gpart: n_wg x r*r partials, reduced in fixed order.
// ------------------------------------------------------------------------------------------
// `wst` != nullptr: the weighted Gram of the current step for a non-uniform diagonal R, row weights 1 / (wst->rho * rho_rows[i] + wst->s)
template <typename T>
__global__ __launch_bounds__(WG) void psmf_gram_partial(const T* __restrict__ C, int d_local, int r, int rp,
                                                        int rows_per_wg, double* __restrict__ gpart,
                                                        const DevState* __restrict__ wst = nullptr, const double* __restrict__ rho_rows = nullptr) {
  constexpr int TR = 32;   // rows per LDS tile
  __shared__ double tile[TR][RM + 1];
  const int tid = threadIdx.x;
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
  const int row_begin = blockIdx.x * rows_per_wg;
  const int row_end = min(row_begin + rows_per_wg, d_local);
  for (int base = row_begin; base < row_end; base += TR) {
    __syncthreads();
    for (int idx = tid; idx < TR * r; idx += WG) {
      const int rr = idx / r, c = idx - rr * r;
      const int row = base + rr;
      double v = row < row_end ? (double)C[(size_t)row * rp + c] : 0.0;
      if (wst && row < row_end) v *= sqrt(fast_rcp(wst->rho * rho_rows[row] + wst->s));      // both factors of a pair carry sqrt(kappa_i)
      tile[rr][c] = v;
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
#pragma unroll
  for (int u = 0; u < MU; ++u)
    if (qa[u] >= 0) gpart[(size_t)blockIdx.x * r * r + tid + u * WG] = acc[u];
}

__global__ void psmf_gram_reduce(const double* __restrict__ gpart, int n_wg, int rr, double* __restrict__ G) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < rr) {
    double a = 0.0;
    for (int w = 0; w < n_wg; ++w) a += gpart[(size_t)w * rr + q];
    G[q] = a;
  }
}

// ------------------------------------------------------------------------------------------
// predict roll-out: out[t][row] = C[row] . mu_pred[t]          psmf.py:182-188
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(WG) void psmf_predict_rows(const T* __restrict__ C, int d_local, int r, int rp,
                                                        const double* __restrict__ mu_pred, int n_pred,
                                                        double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_mu = reinterpret_cast<double*>(smem_raw);   // chunk of mu_pred rows
  const int row = blockIdx.x * WG + threadIdx.x;
  double c[RM];
  if (row < d_local)
    for (int l = 0; l < r; ++l) c[l] = (double)C[(size_t)row * rp + l];
  constexpr int TC = 64;
  for (int t0 = 0; t0 < n_pred; t0 += TC) {
    const int nt = min(TC, n_pred - t0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nt * r; idx += WG) s_mu[idx] = mu_pred[(size_t)t0 * r + idx];
    __syncthreads();
    if (row < d_local) {
      for (int t = 0; t < nt; ++t) {
        double a = 0.0;
        for (int l = 0; l < r; ++l) a += c[l] * s_mu[t * r + l];
        out[(size_t)(t0 + t) * d_local + row] = a;
      }
    }
  }
}

// plain streaming copy (16-byte accesses, grid-stride): the measured HBM bandwidth bench.py quotes beside the nominal peak
__global__ __launch_bounds__(WG) void psmf_copy_k(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * WG;
  size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {      // four independent 16-byte loads in flight per thread
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

// sum of squared prediction errors over a block of steps (tracking.py:63-76 norms)
template <typename T>
__global__ __launch_bounds__(WG) void psmf_sq_error_k(const T* __restrict__ YP, const T* __restrict__ Y, size_t n,
                                                      double* __restrict__ part) {
  __shared__ double s4[4];
  double a = 0.0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const double dlt = (double)YP[i] - (double)Y[i];
    a += dlt * dlt;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

}  // namespace psmf
