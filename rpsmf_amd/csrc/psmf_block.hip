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
//   psmf_blk_gram_mfma<T>   K partials on the f64 matrix cores (row chunk staged in LDS as float64)
//   psmf_blk_reduce         fixed-order sum of the partials
//   psmf_blk_filter<RPAD>   the nb steps (one workgroup): the general kernel; the role-specialised ones live in psmf_blk3/4/16/32.hip
//   psmf_blk_apply_mfma<T>  C <- Z A_nb (rounded once), y_hat_j = Z b_j
// (round 1's MFMA-free psmf_blk_gram / psmf_blk_apply and their switch PSMF_BLOCK_MFMA were removed in round 5)
#pragma once
#include "psmf_kernels.hip"
#include "psmf_ns.hip"
#include "psmf_dyn.hip"

namespace psmf {

// per-phase cycle accumulation for tools/blk_prof.hip (PSMF_BLK_STAMPS); no-ops in the product
#ifdef PSMF_BLK_STAMPS
#define BLK_T0() unsigned long long bt_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, bl_, bn_; { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bl_) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define BLK_T(n) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bn_) :: "memory"); __builtin_amdgcn_sched_barrier(0); bt_[n] += bn_ - bl_; bl_ = bn_; }
#define BLK_COUNT(i, v) if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(b.Kpart)[200 + (i)] += (v);
#define BLK_TOUT() if ((threadIdx.x & 63) == 0) for (int q_ = 0; q_ < 12; ++q_) reinterpret_cast<unsigned long long*>(b.Kpart)[(threadIdx.x >> 6) * 12 + q_] = bt_[q_];
#else
#define BLK_T0()
#define BLK_T(n)
#define BLK_COUNT(i, v)
#define BLK_TOUT()
#endif

constexpr int RB = 64;           // r + B, padded coefficient dimension (r <= 32)
constexpr int RS = RM / 2 + 1;   // LDS row stride of the RB x r coefficient matrices (odd: lane = row reads are conflict-free)
constexpr int BLK_GRAM_WG = 256; // workgroups (= partials) of the block Gram
constexpr int XGB = 64;          // column capacity of the cross-Gram (>= block length)
constexpr int BLK_TH_CAP = 2304; // theta / gradient sums of at most this many parameters live in LDS during a block (FourierBasis N = 1 at r = 32: 2176; 147 KB of LDS in all)

struct BlockParams {
  StepParams sp;
  double* Kpart;      // BLK_GRAM_WG x RB*RB
  double* K;          // RB x RB
  double* Acoef;      // RB x r   (A_nb)
  double* Bcoef;      // RB x RB  (column j = b_j, stored [j][m])
  long long k0;       // first step of the block is k0 + 1 (0-based series row k0)
  int nb;             // steps in this block
  int gram_rows;      // rows per Gram workgroup
  // pipelined blocks: K of this block is ASSEMBLED from the previous block instead of read from K:
  //   K[0:r,0:r] = G (tracked), K[0:r, r+q] = Aprev^T XG[0:RB, q], K[r+q, r+q'] = XG[RB+q, q']
  // with XG = [Z_prev^T Y ; Y^T Y] computed off the critical path (psmf_blk_xgram_mfma).
  int assemble;
  const double* XG;       // (RB + XGB) x XGB
  const double* Aprev;    // RB x r: coefficient matrix at the end of the previous block
  double* XGpart;         // BLK_GRAM_WG x (RB + XGB) * XGB
  long long k1;           // xgram: first row of the NEXT block in the series
  int nb1;                // xgram: steps of the next block
  // device-flag hand-off of the pipelined blocks (nullptr: the host orders the kernels with events):
  //   flags[0] = xg_seq   highest block sequence number whose K / cross-Gram is complete      (set on the bulk stream)
  //   flags[1] = filt_seq number of blocks whose filter kernel has finished                   (set by the NEXT filter kernel)
  //   flags[2] = abort    a wait timed out
  long long* flags;
  long long seq;          // this block's sequence number
  int last;               // filter3: last block of the run -> also write the row-major r x r state (DevState)
  // chain (filter3): ONE launch advances `chain` consecutive blocks of `chain_B` steps (the last one may be shorter, the
  // run ends at chain_kend); block j of the launch uses slot j & 1 of the ping-pong buffers below, sequence number seq + j,
  // and is assembled from block j - 1 (j > 0).  See psmf_blk_filter3.
  int chain;
  int chain_B;
  long long chain_kend;
  double* Acoef0;         // 2 x RB x RM
  double* Bcoef0;         // 2 x RB x RB
  const double* XG0;      // 2 x (RB + XGB) x XGB
  int dual6;              // psmf_blk_filter6: random walk, Q = q I, full filter, no schedules -> the two inversions of a step side by side
};

// ---- hand-off through device flags -------------------------------------------------------------------------------
// An event wait or an event record between two kernels of one stream costs 5-8 us on this stack (measured gap between
// consecutive filter kernels: 4 us bare, 9.4 with the record, 14.9 with both).  So the filter stream carries nothing
// but filter kernels; each one, when it starts, (a) announces that its predecessor has finished -- the kernel boundary
// made that kernel's stores visible -- which releases the bulk stream's apply, and (b) checks that its own K is there
// (it practically always is: the cross-Gram runs one block ahead).  Every wait is bounded.
constexpr long long HANDOFF_MAX_TICKS = 20LL * 100000000LL;   // 20 s of the 100 MHz real-time counter, then the abort flag
                                                               // (a first RCCL collective may take seconds to connect)

// Relaxed agent-scope accesses: every flag is written by a kernel that starts AFTER the kernel whose data it announces has
// ended, and (in the normal case) read before the reader touches that data for the first time in a kernel that started
// after it was set -- the kernel boundaries are the release and the acquire.  (Acquire loads / release stores here cost an
// L1 invalidate / L2 write-back each: ~1 us per block.)  The one exception, a flag observed only after polling, is
// followed by an explicit acquire fence in blk_handoff_begin.
__device__ __forceinline__ long long flag_load(const long long* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store(long long* f, long long v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns false (uniformly) if the run was aborted; ends with a workgroup barrier
__device__ __forceinline__ bool blk_handoff_begin(const BlockParams& b) {
  __shared__ int s_ok;            // a slot of its own: nothing else writes it
  int* s_flag = &s_ok;
  if (!b.flags) return true;
  if (threadIdx.x == 0) {
    int ok = 1;
    flag_store(b.flags + 1, b.seq);                      // blocks < seq are complete
    const long long xg0 = flag_load(b.flags + 0), ab0 = flag_load(b.flags + 2);   // both loads in flight together
    if (ab0 != 0) ok = 0;
    if (ok && xg0 < b.seq) {
      long long polls = 0;
      const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
      while (flag_load(b.flags + 0) < b.seq) {
        __builtin_amdgcn_s_sleep(16);
        ++polls;
        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > HANDOFF_MAX_TICKS || flag_load(b.flags + 2) != 0) { ok = 0; break; }
      }
      if (polls > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // K was written after this kernel started
      if (!ok) { flag_store(b.flags + 2, 1); if (b.sp.st->err == 0) b.sp.st->err = -7; }
    }
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// chain: between two blocks of one launch.  The stores of the block that just ended are complete (coefficients for the
// apply kernel were stored with agent scope, see coef_store), so it is announced; then the block's own cross-Gram is
// awaited (it practically always is there).  Returns false (uniformly) if the run was aborted; ends with a barrier.
__device__ __forceinline__ bool blk_chain_next(const BlockParams& b, const long long seq) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    flag_store(b.flags + 1, seq);                      // blocks < seq are complete
    const long long xg0 = flag_load(b.flags + 0), ab0 = flag_load(b.flags + 2);
    if (ab0 != 0) ok = 0;
    if (ok && xg0 < seq) {
      const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
      while (flag_load(b.flags + 0) < seq) {
        __builtin_amdgcn_s_sleep(4);
        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > HANDOFF_MAX_TICKS || flag_load(b.flags + 2) != 0) { ok = 0; break; }
      }
      if (!ok) { flag_store(b.flags + 2, 1); if (b.sp.st->err == 0) b.sp.st->err = -7; }
    }
    s_ok = ok;
  }
  __syncthreads();
  __builtin_amdgcn_s_dcache_inv();      // the scalar cache does not see this kernel's own vector stores (DevState fields)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return s_ok != 0;
}

// Coefficients the apply kernel (other XCDs, released by a device flag instead of a kernel boundary when blocks are
// chained) reads: agent-scope stores go through to where every XCD sees them.  The cross-Gram is read the same way.
__device__ __forceinline__ void coef_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double xg_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void psmf_flag_set_k(long long* f, long long v) { flag_store(f, v); }

// Can a kernel of one stream wait for a kernel of another stream that was submitted AFTER it?  (The chained filter kernel
// does; a tool that serialises kernel dispatches -- counter collection -- makes that a dead wait.)  The waiter gives up
// after `ticks` of the 100 MHz counter and reports in out[0]: 1 = the flag came, 0 = it did not.
__global__ void psmf_probe_wait_k(long long* f, long long v, long long ticks, int* out) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  int ok = 1;
  while (flag_load(f) < v) {
    __builtin_amdgcn_s_sleep(8);
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > ticks) { ok = 0; break; }
  }
  out[0] = ok;
}

// bulk stream: hold the stream until flags[1] >= v (the filter kernel of block v - 1 has finished)
__global__ void psmf_flag_wait_k(long long* flags, long long v, DevState* st) {
  if (threadIdx.x != 0) return;
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while (flag_load(flags + 1) < v) {
    __builtin_amdgcn_s_sleep(32);
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > HANDOFF_MAX_TICKS || flag_load(flags + 2) != 0) {
      flag_store(flags + 2, 1);
      if (st->err == 0) st->err = -7;
      break;
    }
  }
}


// ---- f64-MFMA versions of the two d-sized products of a block -------------------------------
// K = Z^T Z with v_mfma_f64_16x16x4_f64: a tile of 32 rows of Z is staged in LDS as float64 (row
// stride GZ_S = 80 doubles: the two rows a 32-lane group touches sit 32 banks apart -> conflict-free);
// wave w accumulates the four 16 x 16 output tiles (w, 0..3) -- both operands are read with the same
// (row k, column) pattern because A = Z^T.  One partial per workgroup, reduced in fixed order.
constexpr int GZ_S = 80;
constexpr int GZ_TR = 32;
template <typename T>
__global__ __launch_bounds__(WG) void psmf_blk_gram_mfma(BlockParams b) {
  __shared__ double sZ[GZ_TR * GZ_S];
  const StepParams& p = b.sp;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int r = p.r, rp = p.rp, dl = p.d_local, nb = b.nb;
  const T* __restrict__ C = reinterpret_cast<const T*>(p.C);
  const T* __restrict__ Y = reinterpret_cast<const T*>(p.Y) + (size_t)(b.k0 - p.series_t0) * dl;
  f64x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  // zero the never-written columns once (r + nb .. RB)
  for (int idx = tid; idx < GZ_TR * RB; idx += WG) sZ[(idx / RB) * GZ_S + (idx % RB)] = 0.0;
  const int row_begin = blockIdx.x * b.gram_rows;
  const int row_end = min(row_begin + b.gram_rows, dl);
  for (int base = row_begin; base < row_end; base += GZ_TR) {
    __syncthreads();
    // C part: thread = (row, column) with columns fastest (row-major C); clamped row, masked value
    for (int idx = tid; idx < GZ_TR * r; idx += WG) {
      const int rr = idx / r, c = idx - rr * r;
      const double v = (double)C[(size_t)min(base + rr, row_end - 1) * rp + c];
      sZ[rr * GZ_S + c] = (base + rr < row_end) ? v : 0.0;
    }
    // series part: rows fastest (Y is time-major)
    for (int idx = tid; idx < GZ_TR * nb; idx += WG) {
      const int q = idx / GZ_TR, rr = idx - q * GZ_TR;
      const double v = (double)Y[(size_t)q * dl + min(base + rr, row_end - 1)];
      sZ[rr * GZ_S + r + q] = (base + rr < row_end) ? v : 0.0;
    }
    __syncthreads();
    double av[GZ_TR / 4], bv[4][GZ_TR / 4];
#pragma unroll
    for (int q = 0; q < GZ_TR / 4; ++q) {
      av[q] = sZ[(4 * q + lk) * GZ_S + 16 * w + lr];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[t][q] = sZ[(4 * q + lk) * GZ_S + 16 * t + lr];
    }
#pragma unroll
    for (int q = 0; q < GZ_TR / 4; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[t][q], acc[t], 0, 0, 0);
  }
  double* out = b.Kpart + (size_t)blockIdx.x * RB * RB;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) out[(16 * w + lk + 4 * q) * RB + 16 * t + lr] = acc[t][q];
}

// [C_new | Y_hat] = Z [A | Bc]  with f64 MFMA: one wave per slab of 16 rows (wave-private LDS image of
// the slab, no workgroup barrier in the loop), the 64 x 64 coefficient matrix in LDS for all waves.
constexpr int AP_S = 66;     // slab row stride (A operand: lanes = rows)
template <typename T>
__global__ __launch_bounds__(WG) void psmf_blk_apply_mfma(BlockParams b) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sW = reinterpret_cast<double*>(smem_raw);     // RB x GZ_S: rows = coefficient index, cols [0,r) = A, [r, r+nb) = b_j
  double* sZall = sW + RB * GZ_S;                      // 4 waves x 16 x AP_S
  const StepParams& p = b.sp;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int r = p.r, rp = p.rp, dl = p.d_local, nb = b.nb;
  for (int idx = tid; idx < RB * RB; idx += WG) {
    const int m = idx / RB, c = idx - m * RB;
    double v = 0.0;
    if (c < r) v = b.Acoef[m * r + c];
    else if (c < r + nb) v = b.Bcoef[(size_t)(c - r) * RB + m];
    sW[m * GZ_S + c] = v;
  }
  double* sZ = sZall + w * 16 * AP_S;
  for (int idx = lane; idx < 16 * AP_S; idx += 64) sZ[idx] = 0.0;
  __syncthreads();
  T* __restrict__ C = reinterpret_cast<T*>(p.C);
  const T* __restrict__ Y = reinterpret_cast<const T*>(p.Y) + (size_t)(b.k0 - p.series_t0) * dl;
  T* __restrict__ YP = p.store_yp ? reinterpret_cast<T*>(p.YP) + (size_t)(b.k0 - p.series_t0) * dl : nullptr;
  const int nslab = (dl + 15) / 16;
  for (int slab = blockIdx.x * 4 + w; slab < nslab; slab += gridDim.x * 4) {
    const int row0 = slab * 16;
    // stage the slab (wave-private): C part columns fastest, series part rows fastest
    for (int idx = lane; idx < 16 * r; idx += 64) {
      const int rr = idx / r, c = idx - rr * r;
      sZ[rr * AP_S + c] = (double)C[(size_t)min(row0 + rr, dl - 1) * rp + c];
    }
    for (int idx = lane; idx < 16 * nb; idx += 64) {
      const int q = idx >> 4, rr = idx & 15;
      sZ[rr * AP_S + r + q] = (double)Y[(size_t)q * dl + min(row0 + rr, dl - 1)];
    }
    __builtin_amdgcn_s_waitcnt(0);          // wave-private LDS image complete (vmcnt, lgkmcnt = 0)
    __builtin_amdgcn_wave_barrier();
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double av[8], bv[4][8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 32 * half + 4 * q + lk;
        av[q] = sZ[lr * AP_S + k];
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t][q] = sW[k * GZ_S + 16 * t + lr];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[t][q], acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();         // the slab image may be overwritten by the next iteration
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = 16 * t + lr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = row0 + lk + 4 * q;
        if (row < dl) {
          if (col < r) C[(size_t)row * rp + col] = (T)acc[t][q];
          else if (col < r + nb && YP) YP[(size_t)(col - r) * dl + row] = (T)acc[t][q];
        }
      }
    }
  }
}

inline size_t blk_apply_lds_bytes() { return ((size_t)RB * GZ_S + 4 * 16 * AP_S) * 8; }

// Cross-Gram for the NEXT block, computed while the current block is being filtered:
//   XG[0:RB, q]      = [C_k0 | Y_cur]^T y_next_q        (RB x nb1)
//   XG[RB + q, q']   = y_next_q . y_next_q'             (nb1 x nb1)
// Same staging and MFMA structure as psmf_blk_gram_mfma; wave w owns the row tiles 2w, 2w+1 of the
// 8 row tiles of [Z | Y_next] (tiles 0-3 = Z, 4-7 = Y_next) against the column tiles of Y_next.
template <typename T>
__global__ __launch_bounds__(WG) void psmf_blk_xgram_mfma(BlockParams b) {
  __shared__ double sZ[GZ_TR * GZ_S];
  __shared__ double sY[GZ_TR * GZ_S];
  const StepParams& p = b.sp;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int r = p.r, rp = p.rp, dl = p.d_local, nb = b.nb, nb1 = b.nb1;
  const int nct = (nb1 + 15) >> 4;            // column tiles actually needed
  const T* __restrict__ C = reinterpret_cast<const T*>(p.C);
  const T* __restrict__ Y = reinterpret_cast<const T*>(p.Y) + (size_t)(b.k0 - p.series_t0) * dl;
  const T* __restrict__ Y1 = reinterpret_cast<const T*>(p.Y) + (size_t)(b.k1 - p.series_t0) * dl;
  f64x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[a][t] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int idx = tid; idx < GZ_TR * RB; idx += WG) { sZ[(idx / RB) * GZ_S + (idx % RB)] = 0.0; sY[(idx / RB) * GZ_S + (idx % RB)] = 0.0; }
  const int row_begin = blockIdx.x * b.gram_rows;
  const int row_end = min(row_begin + b.gram_rows, dl);
  const double* sA_ = (w < 2) ? sZ : sY;      // this wave's row tiles come from Z (w = 0, 1) or Y_next (w = 2, 3)
  const int t0 = (w & 1) * 2;                 // its first 16-column tile inside that image
  for (int base = row_begin; base < row_end; base += GZ_TR) {
    __syncthreads();
    for (int idx = tid; idx < GZ_TR * r; idx += WG) {
      const int rr = idx / r, c = idx - rr * r;
      const double v = (double)C[(size_t)min(base + rr, row_end - 1) * rp + c];
      sZ[rr * GZ_S + c] = (base + rr < row_end) ? v : 0.0;
    }
    for (int idx = tid; idx < GZ_TR * nb; idx += WG) {
      const int q = idx / GZ_TR, rr = idx - q * GZ_TR;
      const double v = (double)Y[(size_t)q * dl + min(base + rr, row_end - 1)];
      sZ[rr * GZ_S + r + q] = (base + rr < row_end) ? v : 0.0;
    }
    for (int idx = tid; idx < GZ_TR * nb1; idx += WG) {
      const int q = idx / GZ_TR, rr = idx - q * GZ_TR;
      const double v = (double)Y1[(size_t)q * dl + min(base + rr, row_end - 1)];
      sY[rr * GZ_S + q] = (base + rr < row_end) ? v : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < GZ_TR / 4; ++q) {
      const int krow = (4 * q + lk) * GZ_S;
      const double a0 = sA_[krow + 16 * t0 + lr], a1 = sA_[krow + 16 * (t0 + 1) + lr];
      double bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[t] = sY[krow + 16 * t + lr];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < nct) {
          acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv[t], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv[t], acc[1][t], 0, 0, 0);
        }
      }
    }
  }
  double* out = b.XGpart + (size_t)blockIdx.x * (RB + XGB) * XGB;
  const int rowbase = (w < 2 ? 0 : RB) + 16 * t0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) out[(size_t)(rowbase + 16 * a + lk + 4 * q) * XGB + 16 * t + lr] = acc[a][t][q];
}

__global__ __launch_bounds__(128) void psmf_blk_xreduce(BlockParams b, double* XGout, int nparts) {
  const int e = blockIdx.x * 128 + threadIdx.x;   // (RB + XGB) * XGB = 8192 = 64 x 128
  XGout[e] = strided_sum(b.XGpart + e, 0, 1, nparts, (RB + XGB) * XGB);
}

__global__ __launch_bounds__(128) void psmf_blk_reduce(BlockParams b, int nparts) {
  const int e = blockIdx.x * 128 + threadIdx.x;   // RB*RB = 4096 = 32 x 128
  b.K[e] = strided_sum(b.Kpart + e, 0, 1, nparts, RB * RB);
}

// K of a pipelined block from the previous block's quantities (see BlockParams):
//   staging: Aprev (RB x r) -> sA image, XG[0:RB, 0:nb] -> sKA image reused as a (RB x nb') strip per pass
// Outputs the full symmetric sK.  Called by all NTH threads of the workgroup; ends with a barrier.
template <int NTH>
__device__ __forceinline__ void assemble_K(const BlockParams& b, double* sK, double* sA, double* sKA, const int r, const int tid) {
  const int nb = b.nb;
  const DevState* st = b.sp.st;
  for (int idx = tid; idx < RB * RB; idx += NTH) sK[idx] = 0.0;
  for (int idx = tid; idx < RB * r; idx += NTH) { const int m = idx / r, c = idx - m * r; sA[m * RS + c] = b.Aprev[idx]; }
  __syncthreads();
  // G block and the series block
  for (int idx = tid; idx < r * r; idx += NTH) { const int i = idx / r, c = idx - i * r; sK[i * RB + c] = st->G[idx]; }
  for (int idx = tid; idx < nb * nb; idx += NTH) { const int q = idx / nb, q2 = idx - q * nb; sK[(r + q) * RB + r + q2] = b.XG[(size_t)(RB + q) * XGB + q2]; }
  // cross block K[i][r+q] = sum_m Aprev[m][i] XG[m][q].  The top RB rows of the cross-Gram go through LDS, 32
  // columns at a time (sKA is free until the caller fills it): one coalesced round trip instead of RB
  // dependent L2 reads per output.
  for (int q0 = 0; q0 < nb; q0 += 32) {
    for (int idx = tid; idx < RB * 32; idx += NTH) {
      const int m = idx >> 5, q = idx & 31;
      if (q0 + q < nb) sKA[m * RS + q] = b.XG[(size_t)m * XGB + q0 + q];
    }
    __syncthreads();
    const int nq = nb - q0 < 32 ? nb - q0 : 32;
    for (int idx = tid; idx < r * nq; idx += NTH) {
      const int q = idx / r, i = idx - q * r;          // consecutive threads -> consecutive i (conflict-free sA reads, sKA broadcast)
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 8
      for (int m = 0; m < RB; m += 2) {
        acc0 += sA[m * RS + i] * sKA[m * RS + q];
        acc1 += sA[(m + 1) * RS + i] * sKA[(m + 1) * RS + q];
      }
      const double acc = acc0 + acc1;
      sK[i * RB + r + q0 + q] = acc;
      sK[(r + q0 + q) * RB + i] = acc;
    }
    __syncthreads();
  }
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
  double* sKA = sA + RB * RS;         // RB x r   (r <= 32 = RM / 2)
  double* s_red = sKA + RB * RS;      // WG
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
  // dynamics (psmf_dyn.hip): dense Jacobian F, P and P F^T images (row stride RS), trig values of the terms, g_f
  double* sF = s4 + 6;                // RM/2 x RS
  double* sPm = sF + (RM / 2) * RS;
  double* sT = sPm + (RM / 2) * RS;
  double* s_val = sT + (RM / 2) * RS; // DYN_MAX_TERMS x RM
  double* s_tp = s_val + DYN_MAX_TERMS * RM;
  double* s_gf = s_tp + DYN_MAX_TERMS * RM;   // RM
  double* s_u = s_gf + RM;            // RM
  double* s_theta = s_u + RM;         // BLK_TH_CAP   (theta and gradsum of the block, when they fit)
  double* s_grad = s_theta + BLK_TH_CAP;
  const bool dense = dyn_dense(p.dyn_kind, p.dyn_flags);
  const bool th_lds = p.n_theta > 0 && p.n_theta <= BLK_TH_CAP;
  StepParams pd = p;                  // what the dynamics see: theta / gradsum in LDS when they fit
  if (th_lds) { pd.theta = s_theta; pd.gradsum = s_grad; }

  if (!blk_handoff_begin(b)) return;
  if (th_lds)
    for (int idx = tid; idx < p.n_theta; idx += WG) { s_theta[idx] = p.theta[idx]; s_grad[idx] = p.gradsum[idx]; }
  if (!b.assemble) {
    for (int idx = tid; idx < RB * RB; idx += WG) sK[idx] = b.K[idx];
  } else {
    assemble_K<WG>(b, sK, sA, sKA, r, tid);
  }
  if (tid == 0) *errflag = 0;
  if (tid < RM) { s_mub[tid] = 0.0; s_h[tid] = 0.0; s_w[tid] = 0.0; s_f[tid] = 1.0; }
  for (int idx = tid; idx < (RM / 2) * RS; idx += WG) sF[idx] = 0.0;      // the MFMA form of F P F^T reads whole 32 x 32 images
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
  __syncthreads();
  // A_0 = [I; 0], K A_0 = first r columns of K, G_0 = K[0:r, 0:r] (exact Gram of the stored C)
  for (int idx = tid; idx < RB * r; idx += WG) {
    const int m = idx / r, c = idx - m * r;
    sA[m * RS + c] = (m == c) ? 1.0 : 0.0;
    sKA[m * RS + c] = sK[m * RB + c];
  }
#pragma unroll
  for (int m = 0; m < M; ++m) Gv[m] = val[m] ? sK[ii[m] * RB + j] : 0.0;
  __syncthreads();

  double s_last = 0.0, eta_last = 0.0, N_last = 0.0, phi = 1.0, omega = 1.0, ee_last = 0.0;
  BLK_T0();
  for (int jb = 0; jb < b.nb; ++jb) {
    const long long kstep = b.k0 + jb + 1;   // 1-based step index
    // ---- S1: mu_bar = f(theta, mu, k), F = df/dx (psmf.py:104-115; psmf_dyn.hip) ----
    dyn_forward<WG>(pd, (double)kstep, s_mu, s_mub, s_f, sF, RS, s_val, s_tp, sT, tid);     // ends with a barrier (sT: scratch here, P F^T below)
    BLK_T(0);
    // PSMFIter reads Q[k], R[k] of the step (psmf.py:115,123,141): scalar schedules (never with rPSMF's running Q, R)
    const double qs = p.q_sched ? p.q_sched[kstep - p.series_t0] : 1.0;
    if (p.rho_sched) rho = p.rho_sched[kstep - p.series_t0];
    // ---- S2: Pbar, w = V mu_bar, <G, Pbar>, s, eta, N, kappa ----
    double Pb[M];
    double part = 0.0, gp = 0.0;
    if (dense && p.pbar_predict && RPAD == 32) {
      // Pbar = F P F^T + Q on the float64 matrix cores (round 3; this instantiation serves r = 17 ... 32): the four waves each
      // form one 16 x 16 tile of T = P F^T and then of F T from zero-padded 32 x 32 LDS images, eight MFMAs per tile and
      // product, every operand of a tile loaded before the first MFMA.  (As two r-long LDS loops per element the two products
      // were 13 900 of a timestep's 53 500 cycles at r = 20, tools/blkgen_prof.hip.)
      const int wv_ = tid >> 6, ln_ = tid & 63, ti_ = wv_ >> 1, tj_ = wv_ & 1, lr_ = ln_ & 15, lk_ = ln_ >> 4;
#pragma unroll
      for (int m = 0; m < M; ++m) sPm[ii[m] * RS + j] = val[m] ? Pv[m] : 0.0;      // (the thread grid covers the padded matrix)
      __syncthreads();
      {
        double a[8], bq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          a[q] = sPm[(16 * ti_ + lr_) * RS + 4 * q + lk_];         // P[i][k]
          bq[q] = sF[(16 * tj_ + lr_) * RS + 4 * q + lk_];         // F^T[k][j] = F[j][k]  (F zero outside r x r)
        }
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bq[q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) sT[(16 * ti_ + lk_ + 4 * q) * RS + 16 * tj_ + lr_] = acc[q];
      }
      __syncthreads();
      {
        double a[8], bq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          a[q] = sF[(16 * ti_ + lr_) * RS + 4 * q + lk_];          // F[i][k]
          bq[q] = sT[(4 * q + lk_) * RS + 16 * tj_ + lr_];         // T[k][j]
        }
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bq[q], acc, 0, 0, 0);
        __syncthreads();          // every read of the T image is done: the product goes back into it
#pragma unroll
        for (int q = 0; q < 4; ++q) sT[(16 * ti_ + lk_ + 4 * q) * RS + 16 * tj_ + lr_] = acc[q];
      }
      __syncthreads();
      // back to the element-per-thread layout, symmetrised (F P F^T is symmetric up to round-off; the sweep assumes exact symmetry)
#pragma unroll
      for (int m = 0; m < M; ++m) Pb[m] = val[m] ? 0.5 * (sT[ii[m] * RS + j] + sT[j * RS + ii[m]]) + qs * Qv[m] : 0.0;
    } else if (dense && p.pbar_predict) {
      // Pbar = F P F^T + Q with a dense F: T = P F^T, then F T, through LDS images (odd row stride: conflict-free)
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (val[m]) sPm[ii[m] * RS + j] = Pv[m];
      __syncthreads();
      double tv[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        double a = 0.0;
        if (val[m])
          for (int q = 0; q < r; ++q) a += sPm[ii[m] * RS + q] * sF[j * RS + q];
        tv[m] = a;
      }
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (val[m]) sT[ii[m] * RS + j] = tv[m];
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; ++m) {
        double a = 0.0;
        if (val[m])
          for (int q = 0; q < r; ++q) a += sF[ii[m] * RS + q] * sT[q * RS + j];
        Pb[m] = val[m] ? a + qs * Qv[m] : 0.0;
      }
      // symmetrise (F P F^T is symmetric up to round-off; the sweep inversion assumes exact symmetry)
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (val[m]) sT[ii[m] * RS + j] = Pb[m];
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (val[m]) Pb[m] = 0.5 * (Pb[m] + sT[j * RS + ii[m]]);
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) Pb[m] = val[m] ? (p.pbar_predict ? s_f[ii[m]] * Pv[m] * s_f[j] + qs * Qv[m] : Pv[m]) : 0.0;
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
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
    BLK_T(1);
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
    BLK_T(2);
    // ---- S4: coefficient space: b = A mu_bar, Ka = K[:, r+jb] - KA mu_bar, a = u - b ----
    {
      const int mrow = tid & (RB - 1), qtr = tid >> 6;        // 4 quarter-row partial dots per row
      const int c0 = qtr * (RPAD / 4), c1 = min(c0 + RPAD / 4, r);
      double pb = 0.0, pk = 0.0;
      for (int c = c0; c < c1; ++c) {
        const double mu_c = s_mub[c];
        pb += sA[mrow * RS + c] * mu_c;
        pk += sKA[mrow * RS + c] * mu_c;
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
    BLK_T(3);
    // ---- h = A^T Ka (8 row groups x RPAD columns), ee = a . Ka ----
    {
      double ph = 0.0;
      if (j < r) {
        for (int m = ig; m < RB; m += RG) ph += sA[m * RS + j] * s_Ka[m];
      }
      col_reduce<RPAD>(ph, s_red, s_h);
    }
    double ee = (tid < RB) ? s_a[tid] * s_Ka[tid] : 0.0;
    ee = block_sum(ee, s4);
    BLK_T(4);
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
    BLK_T(5);
    // theta gradient at the pre-update state: g_f = d(incremental likelihood)/df (psmf.py:57-64, rpsmf.py:62-71),
    // then gradsum += J_theta^T g_f
    if (p.n_theta > 0) {
      if (tid < r) {
        const double wi = s_w[tid], hi = s_h[tid];
        double gf;
        if (p.robust) {
          const double D = lam * N;
          gf = dd * wi / N + 0.5 * (dd + lam) * (-2.0 * hi / D - 2.0 * lam * ee * wi / (D * D)) / (1.0 + ee / D);
        } else {
          gf = dd * wi * invN - hi * invN - ee * wi * invN * invN;
        }
        s_gf[tid] = gf;
      }
      __syncthreads();
      dyn_backward<WG>(pd, (double)kstep, s_mu, s_gf, s_val, s_tp, tid);        // ends with a barrier
    }
    BLK_T(6);
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
    {   // thread = row m (lane) x column class: no integer division, conflict-free (odd row stride)
      const int m = tid & (RB - 1);
      const double am = s_a[m], km = s_Ka[m];
      for (int c = tid >> 6; c < r; c += WG / RB) {
        const double wc = s_w[c] * invN;
        sA[m * RS + c] += am * wc;
        sKA[m * RS + c] += km * wc;
      }
    }
    __syncthreads();           // all reads of s_mu, s_w, s_h, s_a, s_Ka of this step are done
    if (tid < r) {
      s_mu[tid] = mu_new;
      if (p.mu_hist) p.mu_hist[(size_t)(kstep - p.series_t0) * r + tid] = mu_new;
    }
    s_last = s; eta_last = eta; N_last = N; ee_last = ee;
    __syncthreads();
    BLK_T(7);
    // PSMFRecursive: optimiser step on theta every update_every observations (psmf.py:299-304)
    if (p.recursive && p.n_theta > 0 && (kstep % p.update_every) == 0) dyn_adam_step<WG>(pd, kstep, tid);
    BLK_T(8);
  }
  BLK_TOUT();

  // ---- block end: coefficients and state back to memory ----
  for (int idx = tid; idx < RB * r; idx += WG) { const int m = idx / r; b.Acoef[idx] = sA[m * RS + (idx - m * r)]; }
  if (th_lds)
    for (int idx = tid; idx < p.n_theta; idx += WG) { p.gradsum[idx] = s_grad[idx]; if (p.recursive) p.theta[idx] = s_theta[idx]; }
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
  if (tid == 0) {
    st->k = b.k0 + b.nb;
    st->rho = rho; st->lam = lam; st->phi = phi; st->omega = omega; st->ee = ee_last;
    st->s_done = s_last; st->eta_done = eta_last; st->N_done = N_last;
    if (*errflag && st->err == 0) st->err = (int)(b.k0 + 1);
    st->ns_valid = 0;          // nothing the two-inversion kernels carry from block to block describes this state
  }
}

// ------------------------------------------------------------------------------------------
// Two-group variant for the common configuration (full filter, random-walk dynamics, Q = q I):
// 512 threads; half X owns the r x r state machine, half Y owns the coefficient-space matrices.
// The two r x r inversions of a step are made independent,
//     P+_k      = M_k^-1,                       M_k = Lbar_k + kappa_k G_{k-1}      (half X)
//     Lbar_{k+1} = Pbar_{k+1}^-1 = (1/omega_k) [ I/q - W_k / q^2 ],  W_k = (M_k / beta + I/q)^-1   (half Y)
// (Woodbury on Pbar_{k+1} = omega_k (beta M_k^-1 + q I)), and run in lockstep sharing the barriers,
// while the coefficient-space products of the step overlap with the r x r scalar work.
// ------------------------------------------------------------------------------------------
template <int RPAD>
__global__ __launch_bounds__(2 * WG) void psmf_blk_filter2(BlockParams b) {
  constexpr int RG = WG / RPAD;
  constexpr int M = (RPAD * RPAD) / WG > 0 ? (RPAD * RPAD) / WG : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const StepParams& p = b.sp;
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x, lt = tid & (WG - 1), j = lt % RPAD, ig = lt / RPAD;
  const bool X = tid < WG, Yg = !X;
  const int lw = (tid >> 6) & 3;
  const int r2 = r + (r & 1);
  const double dd = (double)p.d;
  // ---- LDS carve ----
  double* sK = sm;                    // RB x RB
  double* sA = sK + RB * RB;          // RB x r
  double* sKA = sA + RB * RS;         // RB x r
  double* sL = sKA + RB * RS;         // RPAD x RPAD: Lbar (precision of the predictive state), Y -> X
  double* s_redX = sL + RM * RM / 4;  // WG
  double* s_redY = s_redX + WG;       // WG
  double* s_mub = s_redY + WG;        // RM each below
  double* s_w = s_mub + RM;
  double* s_h = s_w + RM;
  double* s_vec = s_h + RM;
  double* s_mu = s_vec + RM;
  double* s_a = s_mu + RM;            // RB
  double* s_Ka = s_a + RB;            // RB
  double* s_p2 = s_Ka + RB;           // 4 x RB x 2
  double* rowbufX = s_p2 + 8 * RB;    // 4 * RM
  double* rowbufY = rowbufX + 4 * RM; // 4 * RM
  double* s4X = rowbufY + 4 * RM;     // 4
  double* s4Y = s4X + 4;              // 4
  double* s_sc = s4Y + 4;             // 16 scalars: 0 kappa 1 N 2 invN 3 ee 4 omega 5 phi 6 vscale 7 pscale
  double* s_nrm = s_sc + 16;          // 8: Newton-Schulz residual norms^2, [half][wave]
  double* nsM = s_nrm + 8 + (X ? 0 : 4 * NS_N * NS_S);   // per half: matrix, iterate (ping-pong), residual (NS_N x NS_S each)
  double* nsX0 = nsM + NS_N * NS_S;
  double* nsX1 = nsX0 + NS_N * NS_S;
  double* nsR = nsX1 + NS_N * NS_S;
  int* errflag = reinterpret_cast<int*>(s_nrm + 8 + 8 * NS_N * NS_S);
  constexpr int NT = RPAD > 16 ? 32 : 16;      // Newton-Schulz tile size
  const int nti = lw >> 1, ntj = lw & 1, lane = tid & 63;
  const bool ns_wave = (NT == 32) || lw == 0;

  if (!blk_handoff_begin(b)) return;
  if (!b.assemble) {
    for (int idx = tid; idx < RB * RB; idx += 2 * WG) sK[idx] = b.K[idx];
  } else {
    assemble_K<2 * WG>(b, sK, sA, sKA, r, tid);
  }
  if (tid == 0) *errflag = 0;
  for (int idx = lt; idx < NS_N * NS_S; idx += WG) {       // identity padding of the Newton-Schulz images
    const int i = idx / NS_S, c = idx - i * NS_S;
    nsM[idx] = (i == c) ? 1.0 : 0.0;
    nsX0[idx] = (i == c) ? 1.0 : 0.0;
    nsX1[idx] = (i == c) ? 1.0 : 0.0;
    nsR[idx] = 0.0;
  }
  if (tid < RM) { s_mub[tid] = 0.0; s_h[tid] = 0.0; s_w[tid] = 0.0; s_vec[tid] = 0.0; }
  if (tid < r) s_mu[tid] = st->mu[tid];
  double Vv[M], Pv[M], Gv[M], Lv[M];
  bool val[M];
  int ii[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    ii[m] = ig + m * RG;
    val[m] = (j < r) && (ii[m] < r);
    const int idx = val[m] ? ii[m] * r + j : 0;
    const double lv = st->V[idx], lp = st->P[idx];
    Vv[m] = val[m] ? lv : 0.0;
    Pv[m] = val[m] ? lp : 0.0;
    Lv[m] = 0.0;
  }
  double q = st->Q[0];                 // Q = q I (checked by the host)
  double rho = st->rho, lam = st->lam;
  __syncthreads();
  for (int idx = tid; idx < RB * r; idx += 2 * WG) {
    const int m = idx / r, c = idx - m * r;
    sA[m * RS + c] = (m == c) ? 1.0 : 0.0;
    sKA[m * RS + c] = sK[m * RB + c];
  }
#pragma unroll
  for (int m = 0; m < M; ++m) Gv[m] = val[m] ? sK[ii[m] * RB + j] : 0.0;
  int ns_skip = 0;                    // steps to wait before the next Newton-Schulz attempt after a failure
  int c_ns = 0, c_sw = 0, c_it = 0, c_fail = 0;   // diagnostics (uniform over the workgroup)
  float c_l0 = 0.f, c_l1 = 0.f, c_l2 = 0.f, c_m0 = -99.f;
  double Xp[M];                       // inverse found at the previous step (Newton-Schulz start)
  const bool carried = st->ns_valid == 1 || st->ns_valid == 3;     // uniform: the previous block left Lbar and both inverses behind
  if (carried) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int idx = val[m] ? ii[m] * r + j : 0;
      const double ll = st->Lbar[idx], xx = X ? st->XpX[idx] : st->XpY[idx];
      Lv[m] = val[m] ? ll : 0.0;
      Xp[m] = val[m] ? xx : 0.0;
      if (Yg && val[m]) sL[ii[m] * RPAD + j] = Lv[m];
    }
  } else {
    // Lbar_1 = (P + q I)^-1: both halves run the same sweep in lockstep (one result is kept)
    double A1[M];
#pragma unroll
    for (int m = 0; m < M; ++m) A1[m] = val[m] ? Pv[m] + (ii[m] == j ? q : 0.0) : ((ii[m] == j && j < r2) ? 1.0 : 0.0);
    sweep_all<RPAD>(A1, r2, j, ig, X ? rowbufX : rowbufY, errflag);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      Lv[m] = val[m] ? -A1[m] : 0.0;
      Xp[m] = 0.0;
      if (Yg && val[m]) sL[ii[m] * RPAD + j] = Lv[m];
    }
  }
  if (tid < r) s_mub[tid] = s_mu[tid];          // random walk: mu_bar_1 = mu_0
  __syncthreads();

  double s_last = 0.0, eta_last = 0.0, N_last = 0.0, phi = 1.0, omega = 1.0, ee_last = 0.0;
  BLK_T0();
  for (int jb = 0; jb < b.nb; ++jb) {
    // ---- P2: X: Pbar, partial w, <G, Pbar>;  Y: partial row dots of A mu_bar and KA mu_bar ----
    double Pb[M];
    if (X) {
      double part = 0.0, gp = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        Pb[m] = val[m] ? Pv[m] + (ii[m] == j ? q : 0.0) : 0.0;
        part += Vv[m] * s_mub[ii[m] & (RM - 1)];
        gp += Gv[m] * Pb[m];
      }
      s_redX[lt] = part;
      gp = wave_sum(gp);
      if ((tid & 63) == 0) s4X[lw] = gp;
    } else {
      const int mrow = lt & (RB - 1), qtr = lt >> 6;
      const int c0 = qtr * (RPAD / 4), c1 = min(c0 + RPAD / 4, r);
      double pb = 0.0, pk = 0.0;
      for (int c = c0; c < c1; ++c) {
        const double mu_c = s_mub[c];
        pb += sA[mrow * RS + c] * mu_c;
        pk += sKA[mrow * RS + c] * mu_c;
      }
      s_p2[(qtr * RB + mrow) * 2] = pb;
      s_p2[(qtr * RB + mrow) * 2 + 1] = pk;
    }
    BLK_T(0);
    __syncthreads();
    BLK_T(1);
    // ---- P3: X: w;  Y: b, Ka, a ----
    if (X) {
      if (lt < RPAD) {
        double a = 0.0;
#pragma unroll
        for (int gI = 0; gI < RG; ++gI) a += s_redX[gI * RPAD + lt];
        s_w[lt] = a;
      }
    } else if (lt < RB) {
      double bm = 0.0, km = 0.0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) { bm += s_p2[(qq * RB + lt) * 2]; km += s_p2[(qq * RB + lt) * 2 + 1]; }
      s_a[lt] = (lt == r + jb ? 1.0 : 0.0) - bm;
      s_Ka[lt] = sK[lt * RB + r + jb] - km;
      b.Bcoef[(size_t)jb * RB + lt] = bm;
    }
    BLK_T(2);
    __syncthreads();
    BLK_T(1);
    // ---- P4: X: s, eta, N, kappa;  Y: partial h, partial ee ----
    double kappa = 0.0, N = 0.0, invN = 0.0, s = 0.0, eta = 0.0;
    if (X) {
#pragma unroll 8
      for (int l = 0; l < RPAD; ++l) s += s_mub[l] * s_w[l];
      eta = rho + ((s4X[0] + s4X[1]) + (s4X[2] + s4X[3])) / dd;
      N = s + eta;
      invN = fast_rcp(N);
      kappa = fast_rcp(rho + s);
      if (lt == 0) { s_sc[0] = kappa; s_sc[1] = N; s_sc[2] = invN; }
    } else {
      double ph = 0.0;
      if (j < r)
        for (int m = ig; m < RB; m += RG) ph += sA[m * RS + j] * s_Ka[m];
      s_redY[lt] = ph;
      double e1 = (lt < RB) ? s_a[lt] * s_Ka[lt] : 0.0;
      e1 = wave_sum(e1);
      if ((tid & 63) == 0) s4Y[lw] = e1;
    }
    BLK_T(3);
    __syncthreads();
    BLK_T(1);
    // ---- P5: Y: h, ee (tiny) -- then both halves enter the lockstep inversion ----
    if (Yg) {
      if (lt < RPAD) {
        double a = 0.0;
#pragma unroll
        for (int gI = 0; gI < RG; ++gI) a += s_redY[gI * RPAD + lt];
        s_h[lt] = a;
      }
      if (lt == 0) s_sc[3] = (s4Y[0] + s4Y[1]) + (s4Y[2] + s4Y[3]);
      kappa = s_sc[0];
    }
    BLK_T(4);
    // ---- P6: M = Lbar + kappa G;  X: M^-1 = P+;  Y: (M / beta + I / q)^-1 = W ----
    double R1[M];
    {
      const double ib = X ? 1.0 : 1.0 / p.beta;
      const double dq = X ? 0.0 : 1.0 / q;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const double lb = X ? sL[(ii[m] & (RPAD - 1)) * RPAD + j] : Lv[m];
        R1[m] = val[m] ? (lb + kappa * Gv[m]) * ib + (ii[m] == j ? dq : 0.0) : ((ii[m] == j && j < r2) ? 1.0 : 0.0);
      }
      // Newton-Schulz from the previous step's inverse on the f64 matrix cores; the symmetric sweep is
      // the fallback (first step of the block, start too far, no convergence).  Control flow is uniform
      // over the whole workgroup: both halves see both residual norms.
      bool done = false;
      if ((jb > 0 || carried) && p.use_ns && ns_skip == 0) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          if (val[m]) {
            nsM[ii[m] * NS_S + j] = R1[m];
            nsX0[ii[m] * NS_S + j] = Xp[m];
          }
        }
        __syncthreads();
        // per iteration: [R = I - M X, norm] barrier [X' = X + X R into the other buffer] barrier; the
        // norms are read after the second barrier, so the decision costs no barrier of its own.  Once
        // ||R|| < 3e-7 the update that follows brings it below 1e-13 (R <- R^2): no further check.
        double* xc = nsX0;
        double* xn = nsX1;
        for (int it = 0; it < 8; ++it) {
          double nr = ns_wave ? ns_residual<NT>(nsM, xc, nsR, nti, ntj, lane) : 0.0;
          const float nw = wave_sum_f32_dpp((float)nr);
          if (lane == 0) s_nrm[tid >> 6] = (double)nw;
          __syncthreads();
          if (ns_wave) ns_store_tile(xn, ns_update_tile<NT>(xc, nsR, nti, ntj, lane), nti, ntj, lane);
          __syncthreads();
          const double nx = (s_nrm[0] + s_nrm[1]) + (s_nrm[2] + s_nrm[3]);
          const double ny = (s_nrm[4] + s_nrm[5]) + (s_nrm[6] + s_nrm[7]);
          const double worst = fmax(nx, ny);
          BLK_COUNT(2 + it, 1);
          ++c_it;
          { const float lg = 0.5f * __log10f((float)worst + 1e-38f); if (it == 0) { c_l0 += lg; c_m0 = fmaxf(c_m0, lg); } else if (it == 1) c_l1 += lg; else if (it == 2) c_l2 += lg; }
          if (it == 0) { BLK_COUNT(12, (unsigned long long)(1e6 * sqrt(worst))); }
          double* tsw = xc; xc = xn; xn = tsw;                     // xc now holds the updated iterate
          if (worst < p.ns_tol2) { done = true; break; }            // ||R|| below the tolerance before the update just made
          if (!(worst < 0.09) || it == 7) break;                   // too far (||R|| > 0.3) or not converging
        }
        if (done) {
#pragma unroll
          for (int m = 0; m < M; ++m)
            R1[m] = val[m] ? 0.5 * (xc[ii[m] * NS_S + j] + xc[j * NS_S + ii[m]]) : 0.0;
        } else {
          ns_skip = 3;        // far from the previous inverse (transient): do not pay for the attempt every step
          ++c_fail;
        }
      } else if (ns_skip > 0) {
        --ns_skip;
      }
      BLK_COUNT(done ? 0 : 1, 1);
      if (done) ++c_ns; else ++c_sw;
      if (!done) {
        sweep_all<RPAD>(R1, r2, j, ig, X ? rowbufX : rowbufY, errflag);    // R1 <- -(.)^-1; barriers shared by both halves
#pragma unroll
        for (int m = 0; m < M; ++m) R1[m] = val[m] ? -R1[m] : 0.0;
      }
#pragma unroll
      for (int m = 0; m < M; ++m) Xp[m] = R1[m];
    }
    // ---- P7: X: partial P+ h ----
    if (X) {
      double pt = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) pt += R1[m] * s_h[ii[m] & (RM - 1)];
      s_redX[lt] = pt;
    }
    __syncthreads();
    if (X && lt < RPAD) {
      double a = 0.0;
#pragma unroll
      for (int gI = 0; gI < RG; ++gI) a += s_redX[gI * RPAD + lt];
      s_vec[lt] = a;
    }
    __syncthreads();
    BLK_T(6);
    // ---- P8: X: mu, omega, phi ----
    const double ee = s_sc[3];
    double mu_new = 0.0;
    if (X) {
      double hPh = 0.0;
#pragma unroll 8
      for (int l = 0; l < RPAD; ++l) hPh += s_h[l] * s_vec[l];
      const double quad = kappa * ee - kappa * kappa * hPh;
      if (lt < r) mu_new = s_mub[lt] + kappa * s_vec[lt];
      double om = 1.0, ph = 1.0;
      if (p.robust) {
        const double ild = fast_rcp(lam + dd);
        ph = (lam + ee * invN) * ild;
        om = (lam + quad) * ild;
      }
      if (lt == 0) { s_sc[4] = om; s_sc[5] = ph; }
      // random walk: mu_bar_{k+1} = mu_k.  s_mub[lt] is read in this phase only by this thread (above);
      // its other readers (P2, P4) are behind the barrier below and want the NEW value.
      if (lt < r) {
        s_mu[lt] = mu_new;
        s_mub[lt] = mu_new;
        if (p.mu_hist) p.mu_hist[(size_t)(b.k0 + jb + 1 - p.series_t0) * r + lt] = mu_new;
      }
    }
    __syncthreads();
    BLK_T(7);
    // ---- P9: X: V, P, G;  Y: G, Lbar, A, KA ----
    omega = s_sc[4];
    phi = s_sc[5];
    invN = s_sc[2];
    {
      const double wj = s_w[j & (RM - 1)], hj = s_h[j & (RM - 1)];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (val[m]) {
          const double wi = s_w[ii[m]], hi = s_h[ii[m]];
          Gv[m] += (hi * wj + wi * hj) * invN + ee * (wi * wj) * (invN * invN);
          if (X) {
            Vv[m] = (p.robust ? p.alpha * phi : 1.0) * (Vv[m] - wi * wj * invN);
            Pv[m] = (p.robust ? p.beta * omega : 1.0) * R1[m];
          } else {
            // Lbar_{k+1} = (1/omega) (I/q - W/q^2)
            Lv[m] = ((ii[m] == j ? 1.0 / q : 0.0) - R1[m] / (q * q)) / omega;
            sL[ii[m] * RPAD + j] = Lv[m];
          }
        }
      }
    }
    {   // rank-1 updates, one matrix per half: thread = row m (lane) x column class (no integer
        // division, odd row stride: conflict-free)
      const int m = lt & (RB - 1);
      double* tgt = X ? sKA : sA;
      const double cm = X ? s_Ka[m] : s_a[m];
      for (int c = lt >> 6; c < r; c += WG / RB) tgt[m * RS + c] += cm * (s_w[c] * invN);
    }
    if (p.robust) {
      q *= omega;
      rho *= omega;
      if (!p.fixed_lambda) lam += dd;
    }
    s_last = s; eta_last = eta; N_last = s_sc[1]; ee_last = ee;
    BLK_T(8);
    __syncthreads();                       // every read of s_w, s_h, s_a, s_Ka of this step is done
    BLK_T(9);
  }
  BLK_TOUT();

  // ---- block end ----
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (val[m]) {
      const int idx = ii[m] * r + j;
      if (X) st->XpX[idx] = Xp[m];
      else { st->XpY[idx] = Xp[m]; st->Lbar[idx] = Lv[m]; }
    }
  }
  if (tid == 0) { st->ns_valid = 1; st->cnt[0] += c_ns; st->cnt[1] += c_sw; st->cnt[2] += c_it; st->cnt[3] += c_fail; }
  for (int idx = tid; idx < RB * r; idx += 2 * WG) { const int m = idx / r; b.Acoef[idx] = sA[m * RS + (idx - m * r)]; }
  if (X) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (val[m]) {
        const int idx = ii[m] * r + j;
        st->V[idx] = Vv[m];
        st->P[idx] = Pv[m];
        st->G[idx] = Gv[m];
        st->Q[idx] = (ii[m] == j) ? q : 0.0;
      }
    }
    if (lt < r) st->mu[lt] = s_mu[lt];
    if (lt == 0) {
      st->k = b.k0 + b.nb;
      st->rho = rho; st->lam = lam; st->phi = phi; st->omega = omega; st->ee = ee_last;
      st->s_done = s_last; st->eta_done = eta_last; st->N_done = N_last;
      if (*errflag && st->err == 0) st->err = (int)(b.k0 + 1);
    }
  }
}

inline size_t blk_filter2_lds_bytes() {
  const size_t doubles = (size_t)RB * RB + 2 * (size_t)RB * RS + RM * RM / 4 + 2 * WG + 5 * RM + 2 * RB + 8 * RB + 8 * RM + 8 + 16 + 8 + 8 * (size_t)NS_N * NS_S + 2;
  return (doubles * 8 + 15) & ~(size_t)15;
}

// theta and its summed gradient are kept in LDS for the duration of a block when they fit (n_theta <= BLK_TH_CAP): the dynamics
// read / accumulate them every timestep (FourierBasis N = 2, r = 10: 480 parameters; from global memory that was 24 000 of the
// 42 000 cycles of a timestep, tools/blkgen_prof.hip)
inline size_t blk_filter_lds_bytes() {
  const size_t doubles = (size_t)RB * RB + 2 * (size_t)RB * RS + WG + 6 * RM + 2 * RB + 8 * RB + 4 * RM + 4 + 2 +
                         3 * (size_t)(RM / 2) * RS + 2 * (size_t)DYN_MAX_TERMS * RM + 2 * RM + 2 * (size_t)BLK_TH_CAP;
  return (doubles * 8 + 15) & ~(size_t)15;
}


}  // namespace psmf
