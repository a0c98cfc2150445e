// Persistent per-step engine (gfx950): ONE launch advances a whole run of timesteps of the per-step filter
// (pypsmf/psmf/psmf.py:104-177, rpsmf.py:116-184 in the r x r form of SURVEY App. A), instead of two launches per timestep
// (psmf_sweep_solve + psmf_serial, psmf_kernels.hip).
//
//   block 0            the HUB: everything r-sized.  Waves 0-3 ("workers") keep V, P, G, Q, Lbar in registers across the
//                      steps and do the serial stage of a step; waves 6-7 run the step's r x r inversions as wave-local tile
//                      sweeps on the f64 matrix cores (psmf_ns.hip) while waves 0-5 collect the row workgroups' partial sums.
//                      33 <= r <= 48 ("BIG"): no wave can hold its share of five 64 x 64 matrices beside the 3 x 3 tile sweeps, so
//                      every r x r matrix lives in an LDS image (seven of 48 x 49) and waves 0-5 walk their elements through them.
//   blocks 1 .. n      ROW workgroups: each keeps its rows of C ON CHIP, as float64 in registers, for the whole launch (12.8 MB
//                      of float32 storage at d = 1e5, r = 32 = 100 KB of float64 per CU).  Per timestep a row workgroup reads
//                      y_k (prefetched one step ahead), forms y_hat = C mu_bar, e = y - y_hat, h += c e, C += e w^T / N, stores
//                      y_hat, and hands (h, ee) to the hub.  HBM traffic per timestep: y and y_hat only.
//
// Hand-offs inside the launch (MI355X guide, Guideline 16):
//   hub -> rows   mu_bar, w / N of the next step as 8-byte {tag = epoch, 32-bit value} granules, agent-scope relaxed atomic
//                 stores (write-through); ONE wave per row workgroup re-reads them until every tag matches -- the data is the flag.
//   rows -> hub   r + 1 partial sums per workgroup, agent-scope relaxed atomic stores, the storing wave's s_waitcnt vmcnt(0),
//                 then one epoch word per workgroup; the hub polls the words, then loads the sums with agent-scope atomic loads.
// Every polled word is zeroed by the host before each launch (epoch = step within the launch + 1); every spin is bounded by the
// 100 MHz clock and ends the launch with the sticky error flag set.
//
// The float64 copy of C is rounded to the storage type ONCE per launch (the two-launch engine rounds every step), all sums keep a
// fixed order (deterministic), and the launch starts from / leaves behind exactly the DevState the two-launch engine does, so the
// two are interchangeable between launches.
#include "psmf_pstep.h"
#include "psmf_ns.hip"

namespace psmf {

namespace {

constexpr long long PSTEP_SPIN_TICKS = 300000000LL;      // 3 s of the 100 MHz clock: a hand-off that has not come by then never will
constexpr unsigned PSTEP_ABORT_TAG = 0xFFFFFFFFu;

#define PS_DPP64(x, ctrl)                                                                                  \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, true),                \
                   __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, true))

// sum over the GS lanes (GS = 2 .. 16, aligned group) that share a row; every lane of the group gets the sum
template <int GS>
__device__ __forceinline__ double group_sum(double v) {      // (GS = 1: nothing to add)
  if (GS >= 2) v += PS_DPP64(v, 0xB1);     // quad_perm [1,0,3,2]
  if (GS >= 4) v += PS_DPP64(v, 0x4E);     // quad_perm [2,3,0,1]
  if (GS >= 8) v += PS_DPP64(v, 0x141);    // row_half_mirror
  if (GS >= 16) v += PS_DPP64(v, 0x140);   // row_mirror
  return v;
}

__device__ __forceinline__ double ps_xor16_sum(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(h2[0], l2[0]) + __hiloint2double(h2[1], l2[1]);
}
__device__ __forceinline__ double ps_xor32_sum(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(h2[0], l2[0]) + __hiloint2double(h2[1], l2[1]);
}

// sum over the lanes of a wave with equal (lane % GS): rotations inside the 16-lane rows, then the row swaps; fixed order
template <int GS>
__device__ __forceinline__ double cross_sum(double v) {
  if (GS <= 8) v += PS_DPP64(v, 0x128);    // row_ror:8
  if (GS <= 4) v += PS_DPP64(v, 0x124);    // row_ror:4
  if (GS <= 2) v += PS_DPP64(v, 0x122);    // row_ror:2
  v = ps_xor16_sum(v);
  return ps_xor32_sum(v);
}

// sum over the 64 lanes, float64, DPP row operations + four readlanes (a butterfly of __shfl_xor on doubles is two ds_bpermute and an LDS
// round trip per level); fixed order, same value in every lane
__device__ __forceinline__ double ps_wave_sum(double v) {
  v += PS_DPP64(v, 0xB1);
  v += PS_DPP64(v, 0x4E);
  v += PS_DPP64(v, 0x141);
  v += PS_DPP64(v, 0x140);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// sin / cos / pow of the cos-phase dynamics and the in-loop optimiser, OUT of line: inlined, their range reduction and polynomial
// temporaries (60-100 registers each) pushed the hub's loop over 256 registers and the spills landed on the common path -- 21 spilled
// registers cost 1.4 us per timestep (9.4 -> 10.8) for filters that never call them (random walk: n_theta = 0)
__device__ __attribute__((noinline)) double ps_sin(double x) { return sin(x); }
__device__ __attribute__((noinline)) double ps_cos(double x) { return cos(x); }
__device__ __attribute__((noinline)) double ps_pow(double a, double b) { return pow(a, b); }

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, and every wave here keeps global stores
// (y_hat, the mean history, hand-off words) in flight that nothing inside the launch reads back through the cache
__device__ __forceinline__ void ps_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// per-phase shader-clock sums of one lane (diagnostic builds only: -DPSTEP_PROF; tools/probe_pstep_time.py).  Stamps only in the
// UNMASKED instances: with stamps in them the masked instances die in hipcc 7.2's back end ("Illegal instruction detected: Operand has
// incorrect register class"; the error count changes with every stamp removed) -- the masked per-phase numbers of docs/MEASUREMENTS.md
// come from the build before the Gram slices became tagged granules, which still compiled.
#ifdef PSTEP_PROF
#define PS_PROF_DECL(n) long long pf_acc[n]; _Pragma("unroll") for (int pf_i = 0; pf_i < n; ++pf_i) pf_acc[pf_i] = 0; long long pf_prev = 0
#define PS_PROF_START() do { if constexpr (!MASKED) { __builtin_amdgcn_sched_barrier(0); pf_prev = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PS_PROF(i) do { if constexpr (!MASKED) { __builtin_amdgcn_sched_barrier(0); const long long pf_t = (long long)__builtin_amdgcn_s_memtime(); pf_acc[i] += pf_t - pf_prev; pf_prev = pf_t; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PS_PROF_OUT(base, n, cond) do { if constexpr (!MASKED) { if (q.prof && (cond)) { _Pragma("unroll") for (int pf_i = 0; pf_i < n; ++pf_i) q.prof[(base) + pf_i] = pf_acc[pf_i]; } } } while (0)
#else
#define PS_PROF_DECL(n) do { } while (0)
#define PS_PROF_START() do { } while (0)
#define PS_PROF(i) do { } while (0)
#define PS_PROF_OUT(base, n, cond) do { } while (0)
#endif
#define PS_PROFM(i) do { } while (0)

typedef unsigned long long u64;
__device__ __forceinline__ void gran_store(u64* g, unsigned tag, unsigned v) {
  __hip_atomic_store(g, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gran_load(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wt_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double wt_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte write-through-coherent load (buffer_load_dwordx4 ... sc1 through a raw buffer resource: aux bit 4 = sc1): the hub reads
// the row workgroups' partial sums two doubles at a time (8-byte sc1 accesses run at 0.54-0.70 x the 16-byte rate).  The builtin, not
// inline asm: the compiler tracks the load's wait count, so a spill of the destination registers lands behind the data.  (An inline
// asm global_load is invisible to that tracking -- with 256 registers in use hipcc spilled the destinations BEFORE the hand-written
// wait, and the fan-in summed stale scratch: found on the masked r = 32 instances.)
typedef unsigned ps_u32x4 __attribute__((ext_vector_type(4)));
struct ps_f64pair { double x, y; };
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const void* base, const unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ ps_f64pair wt_load2(const __amdgpu_buffer_rsrc_t rs, const int byte_off) {
  const ps_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
  return ps_f64pair{__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2])};
}

__device__ __forceinline__ int lds_word(const volatile int* p) { return *p; }

// ------------------------------------------------------------------------------------------------------------
// ROW workgroup
// ------------------------------------------------------------------------------------------------------------
template <typename T, int RPAD, int NT, int NPMAX, bool MASKED>
__device__ __forceinline__ void pstep_rows(const PstepParams& q, char* smem) {
  constexpr int GS = RPAD / 4;               // lanes per row; a lane owns 4 consecutive columns
  constexpr int RPW = NT / GS;               // rows per pass of the workgroup
  constexpr int NW = NT / 64;
  constexpr int NG = 4 * RPAD + 1;           // granules of a packet
  constexpr int NGL = (NG + 63) / 64;
  const StepParams& p = q.sp;
  const int wg = (int)blockIdx.x - 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = tid % GS, g = tid / GS;
  const int r = p.r, rp = p.rp, dl = p.d_local;
  const int row_begin = wg * q.rows_per_wg;
  const int row_end = min(row_begin + q.rows_per_wg, dl);

  const u64* pkt_mine = q.pkt + (size_t)(wg % PSTEP_PKT_REP) * (PSTEP_REP_STRIDE / 8);      // this workgroup's replica of the packet
  unsigned* s_pkt32 = reinterpret_cast<unsigned*>(smem);                   // NG words (8-byte aligned doubles inside)
  double* s_red = reinterpret_cast<double*>(smem + 2048);                  // NW x (RPAD + 1)
  int* s_ctl = reinterpret_cast<int*>(smem + 2048 + NW * (RPAD + 1) * 8);  // [0] stop

  T* __restrict__ Cg = reinterpret_cast<T*>(p.C);
  // ---- the rows of C, float64, in registers for the whole launch ----
  double c[NPMAX][4];
  unsigned okmask = 0;
#pragma unroll
  for (int ps = 0; ps < NPMAX; ++ps) {
#pragma unroll
    for (int v = 0; v < 4; ++v) c[ps][v] = 0.0;
    {
      const int row = row_begin + ps * RPW + g;
      if (row < row_end) {
        okmask |= 1u << ps;
        const T* src = Cg + (size_t)row * rp + 4 * j;
        if constexpr (sizeof(T) == 4) {
          if (4 * j < rp) {
            const float4 x = *reinterpret_cast<const float4*>(src);
            c[ps][0] = (double)x.x; c[ps][1] = (double)x.y; c[ps][2] = (double)x.z; c[ps][3] = (double)x.w;
          }
        } else {
          if (4 * j < rp) { const double2 x = *reinterpret_cast<const double2*>(src); c[ps][0] = x.x; c[ps][1] = x.y; }
          if (4 * j + 2 < rp) { const double2 x = *reinterpret_cast<const double2*>(src + 2); c[ps][2] = x.x; c[ps][3] = x.y; }
        }
      }
    }
  }
  if (tid == 0) s_ctl[0] = 0;
  const long long t_first = q.k_begin - p.series_t0;       // row of the series buffer holding the first step's y
  const T* __restrict__ Yg = reinterpret_cast<const T*>(p.Y);
  T* __restrict__ YPg = p.store_yp ? reinterpret_cast<T*>(p.YP) : nullptr;
  T ycur[NPMAX];
#pragma unroll
  for (int ps = 0; ps < NPMAX; ++ps) {
    ycur[ps] = Yg[(size_t)t_first * dl + min(row_begin + ps * RPW + g, row_end - 1)];
  }
  // masked handles: the observation mask of the current step (e) and of the next (its Gram is formed one step ahead), one bit per pass
  constexpr int NTG = RPAD > 16 ? 2 : 1;                 // 16-column tiles of the masked Gram
  constexpr int NTT = NTG * (NTG + 1) / 2;               // ... upper tiles
  constexpr int SG = NTG == 1 ? 16 : 48;                 // row stride of a wave's slab (16 mod 32: conflict-free operand reads)
  constexpr int RW = 64 / GS;                            // rows of a pass held by one wave
  constexpr int NSLAB = NPMAX * RW / 16;                 // 16-row slabs of a wave
  double* s_slab = reinterpret_cast<double*>(smem + 8192);                       // NW x 16 x SG  | later NW x NTT x 256 (wave sums)
  double* s_rs = reinterpret_cast<double*>(smem + 8192 + NW * NTT * 256 * 8);    // reduce-scatter staging: [element][source] (<= nge + n_row_wg)
  double* s_cnt = s_rs + 1280;
  const uint8_t* __restrict__ Mg = p.mask;
  auto mask_bits = [&](long long trow) -> unsigned {
    unsigned b = 0;
    if constexpr (MASKED) {
      if (trow > (long long)p.mask_rows - 1) trow = (long long)p.mask_rows - 1;      // (the step after the last: formed, never used)
      const uint8_t* mk = Mg + (size_t)trow * dl;
#pragma unroll
      for (int ps = 0; ps < NPMAX; ++ps) b |= (mk[min(row_begin + ps * RPW + g, row_end - 1)] != 0 ? 1u : 0u) << ps;
    }
    return b;
  };
  unsigned mcur = mask_bits(t_first), mnext = mask_bits(t_first + 1);
  if constexpr (MASKED) {
    for (int idx = tid; idx < NW * 16 * SG; idx += NT) s_slab[idx] = 0.0;      // columns beyond 4 GS of a slab stay zero
  }
  ps_bar();
  PS_PROF_DECL(8);
  PS_PROF_START();

  for (int s = 0; s < q.n_steps; ++s) {
    const unsigned epoch = (unsigned)s + 1u;
    // ---- wait for the step's packet (wave 0), spread it through LDS ----
    if (wv == 0) {
      unsigned val[NGL];
      const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
      int stop = 0;
      for (;;) {
        bool ok = true, ab = false;
#pragma unroll
        for (int m = 0; m < NGL; ++m) {
          const int k = lane + 64 * m;
          const u64 x = gran_load(pkt_mine + min(k, NG - 1));
          val[m] = (unsigned)x;
          const unsigned tag = (unsigned)(x >> 32);
          ok &= (k >= NG) || tag == epoch;
          ab |= (k == NG - 1) && tag == PSTEP_ABORT_TAG;
        }
        if (__any((int)ab)) { stop = 1; break; }
        if (__all((int)ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > PSTEP_SPIN_TICKS) { stop = 2; break; }
      }
#pragma unroll
      for (int m = 0; m < NGL; ++m) { const int k = lane + 64 * m; if (k < NG) s_pkt32[k] = val[m]; }
      if (lane == 0) { if (stop == 0 && s_pkt32[NG - 1] != 0u) stop = 1; }
      if (lane == 0 && stop) s_ctl[0] = stop;
    }
    PS_PROF(0);      // packet wait (wave 0) / idle (others)
    ps_bar();
    PS_PROF(1);
    if (lds_word(s_ctl) != 0) break;
    const double* s_mub = reinterpret_cast<const double*>(s_pkt32);
    const double* s_wn = s_mub + RPAD;
    double mub[4], wn[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { mub[v] = s_mub[4 * j + v]; wn[v] = s_wn[4 * j + v]; }
    const long long t = t_first + s;
    const bool more = s + 1 < q.n_steps;
    const T* __restrict__ ynx = Yg + (size_t)(t + (more ? 1 : 0)) * dl;      // next step's y
    T* __restrict__ yp = YPg ? YPg + (size_t)t * dl : nullptr;
    double hacc[4] = {0.0, 0.0, 0.0, 0.0};
    double eacc = 0.0;
#pragma unroll
    for (int ps = 0; ps < NPMAX; ++ps) {
      {
        double dot = (c[ps][0] * mub[0] + c[ps][1] * mub[1]) + (c[ps][2] * mub[2] + c[ps][3] * mub[3]);
        dot = group_sum<GS>(dot);
        const bool ok = (okmask >> ps) & 1u;
        const bool obs = ok && (!MASKED || ((mcur >> ps) & 1u));       // rows with m_i = 0: no residual, no update (y_hat stored all the same)
        const double e = obs ? (double)ycur[ps] - dot : 0.0;
        ycur[ps] = (T)dot;       // y_hat of the row: stored behind the hand-off below (no global memory traffic inside the passes)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          hacc[v] = fma(c[ps][v], e, hacc[v]);
          c[ps][v] = fma(e, wn[v], c[ps][v]);
        }
        eacc = fma(e, e, eacc);
      }
    }
    PS_PROF(2);      // row passes
    // ---- workgroup partial of (h, ee): lanes of equal j, then the waves through LDS, fixed order ----
#pragma unroll
    for (int v = 0; v < 4; ++v) hacc[v] = cross_sum<GS>(hacc[v]);
    eacc = cross_sum<GS>(eacc);
    if (lane < GS) {
#pragma unroll
      for (int v = 0; v < 4; ++v) s_red[wv * (RPAD + 1) + 4 * lane + v] = hacc[v];
      if (lane == 0) s_red[wv * (RPAD + 1) + RPAD] = eacc;
    }
    ps_bar();
    PS_PROF(3);      // wave sums + barrier
    if (wv == 0) {
      // lane i: elements 2 i, 2 i + 1 of the partial row [h_0 .. h_{r-1}, ee]
      double out[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = 2 * lane + u;
        const int src = e < r ? e : RPAD;        // (e == r: the ee slot; beyond: ee again, never stored)
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += s_red[w * (RPAD + 1) + min(src, RPAD)];
        out[u] = a;
      }
      double* dst = q.part + (size_t)wg * q.ncol2;
      if (2 * lane < r + 1) wt_store(dst + 2 * lane, out[0]);
      if (2 * lane + 1 < r + 1) wt_store(dst + 2 * lane + 1, out[1]);
      // (masked handles too: (h, ee) goes out NOW, before the Gram of the next step -- measured the other way round, with one drain and
      //  one flag for both, the hub's serial stage waited for h behind the Gram: 17.1 -> 21.8 us per masked timestep)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(q.flags + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PS_PROF(4);      // partial row out, drained, flag
    if constexpr (MASKED) {
      // ---------- masked Gram of the NEXT step from the updated rows: G_m = sum_i m_i c_i c_i^T (m of step k + 1) ----------
      // every wave by itself: 16-row slabs of its own rows (zeros where m_i = 0) in a wave-private LDS image, the upper 16 x 16 tiles
      // on the f64 matrix cores (psmf_masked.hip: mgram_body, with the rows coming from registers instead of HBM)
      f64x4 acc[NTT];
#pragma unroll
      for (int t_ = 0; t_ < NTT; ++t_) acc[t_] = f64x4{0.0, 0.0, 0.0, 0.0};
      double* slab = s_slab + wv * 16 * SG;
      const int gl = lane / GS, lr = lane & 15, lk = lane >> 4;
#pragma unroll
      for (int sb = 0; sb < NSLAB; ++sb) {
#pragma unroll
        for (int ps = 0; ps < NPMAX; ++ps) {
          if ((ps * RW) / 16 <= sb && sb <= (ps * RW + RW - 1) / 16) {      // (compile time) pass ps has rows in slab sb
            const int rr = ps * RW + gl;
            if (rr / 16 == sb) {
              const bool on = ((mnext & okmask) >> ps) & 1u;
              double* dstl = slab + (rr & 15) * SG + 4 * j;
              *reinterpret_cast<double2*>(dstl) = make_double2(on ? c[ps][0] : 0.0, on ? c[ps][1] : 0.0);
              *reinterpret_cast<double2*>(dstl + 2) = make_double2(on ? c[ps][2] : 0.0, on ? c[ps][3] : 0.0);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          double a[NTG];
#pragma unroll
          for (int t_ = 0; t_ < NTG; ++t_) a[t_] = slab[(4 * qq + lk) * SG + 16 * t_ + lr];
          int tt = 0;
#pragma unroll
          for (int ta = 0; ta < NTG; ++ta)
#pragma unroll
            for (int tb = ta; tb < NTG; ++tb, ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], a[tb], acc[tt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();        // every lane has read the image before the next slab overwrites it
      }
      double cnt = j == 0 ? (double)__builtin_popcount(mnext & okmask) : 0.0;
      cnt = cross_sum<GS>(cnt);
      ps_bar();                                  // the slabs are done: their memory takes the waves' tiles
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) s_slab[(wv * NTT + tt) * 256 + qq * 64 + lane] = acc[tt][qq];
      if (lane == 0) s_cnt[wv] = cnt;
      ps_bar();
      {
        double* gdst = q.gpart + (size_t)wg * q.nge;
        for (int idx = tid; idx < NTT * 256; idx += NT) {
          double a = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) a += s_slab[w * NTT * 256 + idx];      // fixed order
          wt_store(gdst + idx, a);
        }
        if (tid == 0) {
          double a = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) a += s_cnt[w];
          wt_store(gdst + NTT * 256, a);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave, then the barrier, then one flag
      ps_bar();
      if (tid < PSTEP_GF_REP) __hip_atomic_store(q.gflags + (size_t)tid * (PSTEP_REP_STRIDE / 4) + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every replica
      PS_PROFM(5);    // masked Gram of the next step: slabs, matrix cores, wave sums, partial out, drained, flag
      // ---------- reduce-scatter: this workgroup sums ITS elements of the Gram over all workgroups' partials (fixed order) ----------
      const int nwg = q.n_row_wg, SL = q.slice_len, e0 = wg * SL;
      const int nmine = max(0, min(SL, q.nge - e0));
      if (wv == 0) {
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        int stop = 0;
        for (;;) {
          bool ok = true;
          for (int b = lane; b < nwg; b += 64) ok &= __hip_atomic_load(q.gflags + (size_t)(wg % PSTEP_GF_REP) * (PSTEP_REP_STRIDE / 4) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
          if (__all((int)ok)) break;
          __builtin_amdgcn_s_sleep(1);
          if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > PSTEP_SPIN_TICKS) { stop = 2; break; }
        }
        if (lane == 0 && stop) s_ctl[0] = stop;
      }
      ps_bar();
      if (lds_word(s_ctl) != 0) break;
      // the slice leaves as 8-byte {tag = epoch, 32-bit half} granules (the data is the flag: no drain, no barrier, no flag word)
      u64* gsl = reinterpret_cast<u64*>(q.gslice);
      auto slice_out = [&](const int e, const double a) {
        const u64 bits = (u64)__double_as_longlong(a), tg = (u64)epoch << 32;
        __hip_atomic_store(gsl + 2 * e, tg | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gsl + 2 * e + 1, tg | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      // (the sources go through LDS, two loads per thread: with eight lanes per element loading 32 partials each straight into registers
      //  the one active wave issued 1 024 scattered requests by itself and the phase took 6.7 us instead of 2.2)
      for (int item = tid; item < nwg * nmine; item += NT) {
        const int src = item / nmine, el = item - src * nmine;
        s_rs[el * nwg + src] = wt_load(q.gpart + (size_t)src * q.nge + e0 + el);
      }
      ps_bar();
      if (8 * nmine <= NT) {       // eight lanes per element, then the eight
        const int el = tid >> 3, u = tid & 7;
        double a = 0.0;
        if (el < nmine) {
          const double* rowp = s_rs + el * nwg;
          for (int s0 = u; s0 < nwg; s0 += 32) {
            double v4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v4[k] = rowp[min(s0 + 8 * k, nwg - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) a += (s0 + 8 * k < nwg) ? v4[k] : 0.0;
          }
        }
        a = group_sum<8>(a);
        if (el < nmine && u == 0) slice_out(e0 + el, a);
      } else {                     // few workgroups, long slices: one lane per element
        for (int el = tid; el < nmine; el += NT) {
          double a = 0.0;
          for (int src = 0; src < nwg; ++src) a += s_rs[el * nwg + src];
          slice_out(e0 + el, a);
        }
      }
      ps_bar();                    // (s_rs is free again)
      PS_PROFM(6);    // partials of every workgroup seen, own slice summed and out (one stamp: with two in this region hipcc 7.2 fails with
                     // "Illegal instruction detected: Operand has incorrect register class" in the diagnostic build)
      mcur = mnext;
      mnext = mask_bits(t + 2);
    }
    // y_hat out and the next step's y in, BEHIND the hand-off: the drain above waits for every vector-memory operation of wave 0, and
    // a y load from HBM issued inside the passes would have been the longest of them
#pragma unroll
    for (int ps = 0; ps < NPMAX; ++ps) {
      const int row = row_begin + ps * RPW + g;
      if (yp && ((okmask >> ps) & 1u) && j == 0) yp[row] = ycur[ps];
      ycur[ps] = ynx[min(row, row_end - 1)];
    }
    // (no barrier: wave 0 rewrites s_pkt32 only after every wave has passed the barrier above, behind its own reads of it;
    //  s_red is rewritten behind the next step's first barrier, which wave 0 reaches after reading it)
  }

  PS_PROF_OUT(24, 8, wg == 0 && tid == 0);
  // ---- C back to its storage type: one rounding per launch ----
#pragma unroll
  for (int ps = 0; ps < NPMAX; ++ps) {
    if ((okmask >> ps) & 1u) {
      const int row = row_begin + ps * RPW + g;
      T* dstp = Cg + (size_t)row * rp + 4 * j;
      if constexpr (sizeof(T) == 4) {
        if (4 * j < rp) *reinterpret_cast<float4*>(dstp) = make_float4((float)c[ps][0], (float)c[ps][1], (float)c[ps][2], (float)c[ps][3]);
      } else {
        if (4 * j < rp) *reinterpret_cast<double2*>(dstp) = make_double2(c[ps][0], c[ps][1]);
        if (4 * j + 2 < rp) *reinterpret_cast<double2*>(dstp + 2) = make_double2(c[ps][2], c[ps][3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// HUB: the r x r inversions of a step on one or two waves, operands and results in LDS (row stride LS)
//   carried (StepParams.solve_dual, Lbar = Pbar^-1 known):  role 0  P+ = (Lbar + kappa G)^-1,  role 1  W = ((Lbar + kappa G) / beta + I / q)^-1
//   otherwise (role 0 only):  -Pbar^-1, M = Pbar^-1 + kappa G, P+ = M^-1 (and W from M when dual)
// (the per-step engine's solve block, psmf_wave16.hip: solve_block_wave_t, with LDS in place of DevState)
// ------------------------------------------------------------------------------------------------------------
template <int NTL>
__device__ __forceinline__ void hub_solve(const int role, const bool dual, const bool carried, const int r, const double kappa,
                                          const double iq, const double ib, const double* sL, const double* sG, double* sPp,
                                          double* sW, const int LS, int* s_bad) {
  const int lane = threadIdx.x & 63, lk = lane >> 4, lr = lane & 15;
  const int r2 = r + (r & 1);
  if (role > (carried ? 1 : 0)) return;
  Sw16K swk;
  sw16k_init(swk, lk, lr);
  double A[NTL][NTL][4];
  bool bad = false;
#define HS_FOR(body)                                                                   \
  _Pragma("unroll") for (int ti = 0; ti < NTL; ++ti)                                   \
    _Pragma("unroll") for (int tj = 0; tj < NTL; ++tj)                                 \
      _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) {                               \
        const int i = 16 * ti + lk + 4 * qq, cI = 16 * tj + lr;                        \
        const bool in = i < r && cI < r, pad = (i == cI) && i >= r;                    \
        const int ic = in ? i : 0, cc = in ? cI : 0;                                   \
        (void)pad; (void)ic; (void)cc;                                                 \
        body                                                                           \
      }
  if (carried) {
    HS_FOR({
      const double mv = sL[ic * LS + cc] + kappa * sG[ic * LS + cc];
      A[ti][tj][qq] = in ? (role == 0 ? mv : mv * ib + (i == cI ? iq : 0.0)) : (pad ? 1.0 : 0.0);
    })
    wave_invert_tiles<NTL>(A, r2, swk, bad);
    double* dst = role == 0 ? sPp : sW;
    HS_FOR({ if (in) dst[i * LS + cI] = -A[ti][tj][qq]; })
  } else {
    HS_FOR({ A[ti][tj][qq] = in ? 0.5 * (sL[ic * LS + cc] + sL[cc * LS + ic]) : (pad ? 1.0 : 0.0); })      // sL holds Pbar here
    wave_invert_tiles<NTL>(A, r2, swk, bad);                       // -Pbar^-1
    double Mx[NTL][NTL][4];
    HS_FOR({ Mx[ti][tj][qq] = in ? kappa * sG[ic * LS + cc] - A[ti][tj][qq] : (pad ? 1.0 : 0.0); A[ti][tj][qq] = Mx[ti][tj][qq]; })
    wave_invert_tiles<NTL>(A, r2, swk, bad);                       // -P+
    HS_FOR({ if (in) sPp[i * LS + cI] = -A[ti][tj][qq]; })
    if (dual) {
      HS_FOR({ A[ti][tj][qq] = in ? Mx[ti][tj][qq] * ib + (i == cI ? iq : 0.0) : (pad ? 1.0 : 0.0); })
      wave_invert_tiles<NTL>(A, r2, swk, bad);                     // -W
      HS_FOR({ if (in) sW[i * LS + cI] = -A[ti][tj][qq]; })
    }
  }
#undef HS_FOR
  if (__builtin_amdgcn_readfirstlane(__any((int)bad)) && lane == 0) *s_bad = 1;
}

// column sums over the worker threads' row groups: partial of thread (ig, j) -> out[j]   (two barriers; every wave of the hub calls it)
template <int RPAD, int NWK>
__device__ __forceinline__ void hub_col_reduce(const bool worker, const double partial, double* s_red, double* s_out) {
  constexpr int RG = NWK / RPAD;
  const int tid = threadIdx.x;
  if (worker) s_red[tid] = partial;
  ps_bar();
  if (tid < RPAD) {
    double a = 0.0;
#pragma unroll
    for (int gI = 0; gI < RG; ++gI) a += s_red[gI * RPAD + tid];
    s_out[tid] = a;
  }
  ps_bar();
}

template <int RPAD, int NT, bool MASKED>
__device__ __forceinline__ void pstep_hub(const PstepParams& q, char* smem) {
  constexpr int NWK = RPAD == 64 ? 384 : 256;     // worker threads: waves 0-3 (RPAD = 64: waves 0-5 -- six row groups of 8 rows each, all of the worker loop's threads)
  constexpr int NW = NT / 64;
  constexpr int RG = NWK / RPAD;
  constexpr int M = (RPAD * RPAD) / NWK > 0 ? (RPAD * RPAD) / NWK : 1;
  // BIG (RPAD = 64, 33 <= r <= 48): 16 elements of five matrices per worker do not fit beside the tile sweeps (2.7 KB of scratch per lane
  // when it was tried) -- V, Q and Pbar live in LDS images like G, P+, W and Lbar already do, 48 x 49 each; a worker walks its (row
  // group, column) elements through them, nothing r x r stays in registers across a timestep
  constexpr bool BIG = RPAD == 64;
  constexpr int MB = BIG ? 8 : 1;            // rows ig + 6 m < 48 of a worker's column
  constexpr int LS = BIG ? 49 : RPAD + 1;    // row stride of the r x r LDS images
  constexpr int IMG = BIG ? 48 * 49 : RPAD * (RPAD + 1);
  constexpr int MR = BIG ? 1 : M;            // (the register copies of the small layouts)
  constexpr int NG = 4 * RPAD + 1;
  // threads of the fan-in.  512-thread hub: waves 0, 1, 4, 5 -- the SIMDs (wave id mod 4) of the two solve waves, 6 and 7, are shared
  // with waves 2 and 3, which therefore spend phase A asleep in the workgroup barrier instead of polling LDS beside the tile sweeps
  constexpr int NFT = 256;
  constexpr int NLT = NT - 128;              // threads that run the worker loop (the two solve waves have a loop of their own)
  constexpr int LPC = RPAD == 64 ? 4 : (NT >= 512 ? 8 : 1);     // lanes per column of the fan-in's second level (LPC x (r + 1) <= the worker loop's NLT threads: 4 x 49 at r = 48)
  const StepParams& p = q.sp;
  DevState* st = p.st;
  const int r = p.r, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool worker = tid < NWK;
  const int j = tid % RPAD, ig = (tid % NWK) / RPAD;
  const double dd = (double)p.d;
  const int nwg = q.n_row_wg, ncol2 = q.ncol2, ncol = r + 1;

  double* sPp = reinterpret_cast<double*>(smem);
  double* sW = sPp + IMG;
  double* sL = sW + IMG;                     // Lbar (inversions side by side, carried) or Pbar
  double* sG = sL + IMG;
  double* sV = sG + IMG;                     // BIG only: V, Q, Pbar
  double* sQ = sV + (BIG ? IMG : 0);
  double* sPb = sQ + (BIG ? IMG : 0);
  double* s_seg = sPb + (BIG ? IMG : 0);     // fan-in: [segment][ncol2]
  double* s_red = s_seg + 2 * NT;
  double* s_he = s_red + NWK;                // h[0..r), ee at [r]
  double* s_w = s_he + 2 * (RM + 1);
  double* s_mub = s_w + RM;
  double* s_f = s_mub + RM;
  double* s_vec = s_f + RM;
  double* s_wn = s_vec + RM;
  double* s4 = s_wn + RM;                    // 8
  double* s_sc = s4 + 8;                     // [0] kappa, [1] 1 / q, [2] 1 / beta (solve operands)
  int* s_ctl = reinterpret_cast<int*>(s_sc + 8);      // [0] stop, [1] fan-in epoch seen by wave 0, [2] bad pivot, [3] carried
  double* s_gm = s_sc + 16;                  // masked handles: the reduced Gram of the next step, upper tiles in the MFMA layout | observed count
  constexpr int NTG = RPAD > 16 ? 2 : 1;
  constexpr int NTT = NTG * (NTG + 1) / 2;
  // element (i, c) of the symmetric Gram in s_gm (tiles (ta, tb), ta <= tb; lane = 16 (row & 3) + column, register = (row & 15) >> 2)
  auto gm_index = [](int i, int cI) -> int {
    if ((i >> 4) > (cI >> 4)) { const int t_ = i; i = cI; cI = t_; }
    const int tt = NTG == 1 ? 0 : ((i >> 4) == 0 ? (cI >> 4) : 2);
    const int i16 = i & 15, c16 = cI & 15;
    return tt * 256 + (i16 >> 2) * 64 + (i16 & 3) * 16 + c16;
  };

  // ---------------- the state the two-launch engine left in DevState ----------------
  const bool dual = p.solve_dual != 0;
  const int nsv = st->ns_valid;
  bool carried = dual && nsv == 7;
#define PS_BIG_FOR(...)                                                                                    \
  _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                                         \
    const int i_ = ig + m * RG;                                                                            \
    const bool v_ = worker && j < r && i_ < r;                                                             \
    const int a_ = v_ ? i_ * LS + j : 0, at_ = v_ ? j * LS + i_ : 0;                                       \
    (void)a_; (void)at_;                                                                                   \
    __VA_ARGS__                                                                                            \
  }
  double Vv[MR], Pv[MR], Gv[MR], Qv[MR], Pbv[MR];
  bool val[MR];
  int ii[MR];
  double pscale_last = 1.0;
  if constexpr (BIG) {
    PS_BIG_FOR({
      const int idx = v_ ? i_ * r + j : 0;
      const double lv = st->V[idx], lg = st->G[idx], lq = st->Q[idx], lp = st->Pbar[idx], lpt = st->Pbar[v_ ? j * r + i_ : 0], ll = st->Lbar[idx];
      if (v_) {
        const double pb = 0.5 * (lp + lpt);
        sV[a_] = lv; sG[a_] = lg; sQ[a_] = lq; sPb[a_] = pb;
        sL[a_] = carried ? ll : pb;
      }
    })
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if constexpr (BIG) { ii[m] = 0; val[m] = false; Vv[m] = Gv[m] = Qv[m] = Pbv[m] = Pv[m] = 0.0; continue; }
    ii[m] = ig + m * RG;
    val[m] = worker && (j < r) && (ii[m] < r);
    const int idx = val[m] ? ii[m] * r + j : 0;
    const double lv = st->V[idx], lg = st->G[idx], lq = st->Q[idx], lp = st->Pbar[idx], lpt = st->Pbar[val[m] ? j * r + ii[m] : 0], ll = st->Lbar[idx];
    Vv[m] = val[m] ? lv : 0.0;
    Gv[m] = val[m] ? lg : 0.0;
    Qv[m] = val[m] ? lq : 0.0;
    Pbv[m] = val[m] ? 0.5 * (lp + lpt) : 0.0;
    Pv[m] = 0.0;
    if (val[m]) {
      sG[ii[m] * LS + j] = Gv[m];
      sL[ii[m] * LS + j] = carried ? ll : Pbv[m];
    }
  }
  double rho = st->rho, lam = st->lam;
  double N0 = st->N, kappa0 = st->kappa, s0 = st->s, eta0 = st->eta;
  long long k0 = st->k;
  double nobs = 0.0;
  if constexpr (MASKED) {
    // masked handle: G is the masked Gram of the CURRENT step, reduced by the previous launch (psmf_mgram_reduce or this kernel) into
    // the handle's buffer with the shares of <G_m, Pbar>; eta, N, kappa of the step are formed from it (ExperimentImpute/PSMF.py:77-78)
    const double* mgp = q.mg_out;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const double a = mgp[val[m] ? ii[m] * r + j : 0], b = mgp[val[m] ? j * r + ii[m] : 0];
      Gv[m] = val[m] ? 0.5 * (a + b) : 0.0;
      if (val[m]) sG[ii[m] * LS + j] = Gv[m];
    }
    nobs = mgp[r * r];
    double tr = 0.0;
    for (int w = 0; w < q.mg_ntr; ++w) tr += mgp[r * r + 2 + w];
    eta0 = (rho * nobs + tr) / dd;
    N0 = s0 + eta0;
    kappa0 = fast_rcp(rho + s0);
  }
  const bool vl = tid < r;
  const bool tl = tid < p.n_theta && p.dyn_kind == 1;
  const int tc = tid & (RM - 1);
  const double l_mu = st->mu[tc], l_w = st->w[tc], l_mub = st->mu_bar[tc], l_wn = st->wN[tc], l_th = p.theta[tc], l_gs = p.gradsum[tc],
               l_am = p.adam_m[tc], l_av = p.adam_v[tc];
  double mu_new = vl ? l_mu : 0.0;
  double w_t = vl ? l_w : 0.0;
  double mub_t = vl ? l_mub : 0.0;
  double theta = tl ? l_th : 0.0;
  double gsum = tl ? l_gs : 0.0;
  double am = tl ? l_am : 0.0;
  double av = tl ? l_av : 0.0;
  double phi = 1.0, omega = 1.0, ee_last = 0.0, s_done = s0, eta_done = eta0, N_done = N0;
  if (tid < RM) { s_w[tid] = vl ? l_w : 0.0; s_mub[tid] = vl ? l_mub : 0.0; s_wn[tid] = vl ? (MASKED ? l_w * fast_rcp(N0) : l_wn) : 0.0; s_f[tid] = 0.0; s_vec[tid] = 0.0; }
  if (MASKED && tid == 0 && p.sc_hist) { p.sc_hist[2 * (k0 - p.series_t0)] = s0; p.sc_hist[2 * (k0 - p.series_t0) + 1] = eta0; }
  if (tid < 2 * (RM + 1)) s_he[tid] = 0.0;
  if (tid == 0) {
    s_ctl[0] = 0; s_ctl[1] = 0; s_ctl[2] = 0;
    s_sc[0] = kappa0;
    s_sc[1] = 1.0 / st->Q[0];
    s_sc[2] = p.robust ? 1.0 / p.beta : 1.0;
  }
  ps_bar();
  // packet of the launch's first step
  for (int gI = tid; gI < NG; gI += NT) {
    const int e = gI >> 1;
    const double v = gI == NG - 1 ? 0.0 : (e < RPAD ? s_mub[e] : s_wn[e - RPAD]);
    const unsigned half = gI == NG - 1 ? 0u : ((gI & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v));
#pragma unroll
    for (int rep = 0; rep < PSTEP_PKT_REP; ++rep) gran_store(q.pkt + (size_t)rep * (PSTEP_REP_STRIDE / 8) + gI, 1u, half);
  }

  // fan-in geometry: thread t of waves 0 .. NW-3 sums elements (2 pi, 2 pi + 1) of the partial rows seg, seg + S, ...
  const int npair = ncol2 >> 1;
  const int S = min(NFT / npair, 3 * LPC);   // segments: the second level below sums three of them per lane, LPC lanes per column
  const int f_tid = wv < 2 ? tid : (wv >= 4 && wv < 6 ? tid - 128 : NFT);      // index among the fan-in threads
  const bool f_wave = f_tid < NFT;
  const int f_pi = f_tid % npair, f_seg = f_tid / npair;
  const bool f_on = f_wave && f_seg < S;

  // ---- the solve waves run a loop of their own (same barriers, none of the workers' registers): the kernel's register allocation
  //      is the larger of the two roles, not their sum ----
  if (wv >= NW - 2) {
    PS_PROF_DECL(8);
    PS_PROF_START();
    for (int s = 0; s < q.n_steps; ++s) {
      if (p.coef_update) {
        const int role = wv - (NW - 2);
        if constexpr (RPAD <= 16) hub_solve<1>(role, dual, carried, r, s_sc[0], s_sc[1], s_sc[2], sL, sG, sPp, sW, LS, s_ctl + 2);
        else if constexpr (RPAD == 32) hub_solve<2>(role, dual, carried, r, s_sc[0], s_sc[1], s_sc[2], sL, sG, sPp, sW, LS, s_ctl + 2);
        else hub_solve<3>(role, dual, carried, r, s_sc[0], s_sc[1], s_sc[2], sL, sG, sPp, sW, LS, s_ctl + 2);      // (RPAD = 64: r <= 48, pstep_plan)
      }
      PS_PROF(0);                                                 // the inversion
      ps_bar();                                                   // end of phase A
      PS_PROF(1);                                                 // waiting for the fan-in
      if (lds_word(s_ctl) != 0 || lds_word(s_ctl + 2) != 0) break;
      ps_bar();                                                   // phase B: the workers' barriers, one for one
      if (p.coef_update) { ps_bar(); ps_bar(); }
      ps_bar();
      ps_bar(); ps_bar();
      if constexpr (MASKED) {                                     // the Gram hand-off of the next step: two barriers, one exit
        ps_bar();
        if (lds_word(s_ctl) != 0) break;
        ps_bar();
      }
      ps_bar();
      carried = dual;
      PS_PROF(2);                                                 // the workers' phase B
    }
    PS_PROF_OUT(16, 8, tid == (NW - 2) * 64);
    return;
  }

  int n_done = 0;
  PS_PROF_DECL(16);
  PS_PROF_START();
  for (int s = 0; s < q.n_steps; ++s) {
    const unsigned epoch = (unsigned)s + 1u;
    // =========================== phase A: the fan-in (beside the solve waves' inversions) ===========================
    {
      if (wv == 0) {
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        int stop = 0;
        for (;;) {
          bool ok = true;
          for (int b = lane; b < nwg; b += 64) ok &= __hip_atomic_load(q.flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
          if (__all((int)ok)) break;
          __builtin_amdgcn_s_sleep(1);
          if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > PSTEP_SPIN_TICKS) { stop = 2; break; }
        }
        if (lane == 0) {
          if (stop) *reinterpret_cast<volatile int*>(s_ctl) = stop;
          *reinterpret_cast<volatile int*>(s_ctl + 1) = (int)epoch;
        }
      } else if (f_wave) {
        while (lds_word(s_ctl + 1) < (int)epoch) __builtin_amdgcn_s_sleep(1);      // (wave 0 always sets it, bounded by its own timeout)
      }
      asm volatile("" ::: "memory");
      PS_PROF(0);      // flags of all row workgroups seen
      if (f_on && lds_word(s_ctl) == 0) {
        double a0 = 0.0, a1 = 0.0;
        // every load of the thread in flight at once: ONE memory round trip (rows f_seg, f_seg + S, ...: at most PSTEP_FANIN_ROWS of them,
        // which pstep_plan guarantees), summed in fixed order
        constexpr int FR = BIG ? PSTEP_FANIN_ROWS_BIG : PSTEP_FANIN_ROWS;
        ps_f64pair x[FR];
        const __amdgpu_buffer_rsrc_t rs = wt_rsrc(q.part, (unsigned)nwg * (unsigned)ncol2 * 8u);
#pragma unroll
        for (int u = 0; u < FR; ++u) x[u] = wt_load2(rs, (min(f_seg + u * S, nwg - 1) * ncol2 + 2 * f_pi) * 8);
#pragma unroll
        for (int u = 0; u < FR; ++u) {
          const bool in = f_seg + u * S < nwg;
          a0 += in ? x[u].x : 0.0;
          a1 += in ? x[u].y : 0.0;
        }
        s_seg[f_seg * ncol2 + 2 * f_pi] = a0;
        s_seg[f_seg * ncol2 + 2 * f_pi + 1] = a1;
      }
    }
    PS_PROF(1);        // partial rows loaded, segment sums in LDS
    ps_bar();
    PS_PROF(2);        // (waiting for the solve waves)
    if ((lds_word(s_ctl) | lds_word(s_ctl + 2)) != 0) break;
    // =========================== phase B: the serial stage of step k0 (psmf_kernels.hip: serial_body) ===========================
    if (tid < LPC * ncol) {      // column tid / LPC: lane u of its LPC sums segments u, u + LPC, u + 2 LPC, then the LPC lanes (fixed order)
      const int col = tid / LPC, u = tid % LPC;
      double v[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = s_seg[min(u + LPC * k, S - 1) * ncol2 + col];
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) a += (u + LPC * k < S) ? v[k] : 0.0;
      a = group_sum<LPC>(a);
      if (u == 0) s_he[col] = a;
    }
    // P+ and W of the solve waves, symmetrised
    if constexpr (!BIG) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (p.coef_update) Pv[m] = val[m] ? 0.5 * (sPp[ii[m] * LS + j] + sPp[j * LS + ii[m]]) : 0.0;
        else Pv[m] = Pbv[m];
      }
    }
    ps_bar();
    PS_PROF(3);        // h, ee summed; P+, W read
    const double N = N0, kappa = kappa0;
    const double ee = s_he[r];
    const double wj = j < r ? s_w[j] : 0.0;
    double quad = kappa * ee;
    const double mu_old = mu_new;
    if (p.coef_update) {
      double part = 0.0;
      if constexpr (BIG) {
        PS_BIG_FOR({ part += v_ ? 0.5 * (sPp[a_] + sPp[at_]) * s_he[i_] : 0.0; })
      } else {
#pragma unroll
        for (int m = 0; m < M; ++m) part += val[m] ? Pv[m] * s_he[min(ii[m], r - 1)] : 0.0;
      }
      hub_col_reduce<RPAD, NWK>(worker, part, s_red, s_vec);      // s_vec = P+ h
      const double bPb = ps_wave_sum(lane < r ? s_he[lane] * s_vec[lane] : 0.0);
      quad -= kappa * kappa * bPb;
      if (vl) mu_new = mub_t + kappa * s_vec[tid];
    } else {
      if (vl) mu_new = mub_t;
    }
    PS_PROF(4);        // P+ h, mu
    // theta gradient at the pre-update state   psmf.py:48-66,167-177; rpsmf.py:53-73
    if (tl) {
      const double tk = (double)(k0 + 1);
      const double arg = 2.0 * M_PI * theta * tk + mu_old;
      const double jt = -ps_sin(arg) * (2.0 * M_PI * tk);
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
    // robust scalars   rpsmf.py:133-171
    double vscale = 1.0, pscale = 1.0, qscale = 1.0;
    phi = 1.0; omega = 1.0;
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
    const double iq_old = s_sc[1];
    const double iom = p.robust ? fast_rcp(omega) : 1.0;      // (a float64 division per element is ~30 instructions: one reciprocal instead)
    pscale_last = pscale;
    if constexpr (BIG) {
      PS_BIG_FOR({
        if (v_) {
          const double wi = s_w[i_], hi = s_he[i_], hj = s_he[j];
          sV[a_] = vscale * (sV[a_] - wi * wj * invN);
          if (p.track_g) sG[a_] += (hi * wj + wi * hj) * invN + ee * (wi * wj) * (invN * invN);
          if (qscale != 1.0) sQ[a_] *= qscale;
          if (dual) {
            const double wsym = 0.5 * (sW[a_] + sW[at_]);
            sL[a_] = ((i_ == j ? iq_old : 0.0) - wsym * iq_old * iq_old) * iom;
          }
        }
      })
    }
#pragma unroll
    for (int m = 0; m < (BIG ? 0 : M); ++m) {
      if (val[m]) {
        const double wi = s_w[ii[m]], hi = s_he[ii[m]], hj = s_he[j];
        Vv[m] = vscale * (Vv[m] - wi * wj * invN);
        Pv[m] *= pscale;
        if (p.track_g) Gv[m] += (hi * wj + wi * hj) * invN + ee * (wi * wj) * (invN * invN);
        if (qscale != 1.0) Qv[m] *= qscale;
        // operands of the next step's inversions (the solve waves only read them in phase A): the tracked Gram, and -- inversions side by
        // side -- Pbar'^-1 = (I / q - W / q^2) / omega from the W the solve waves left (symmetrised)
        sG[ii[m] * LS + j] = Gv[m];
        if (dual) {
          const double wsym = 0.5 * (sW[ii[m] * LS + j] + sW[j * LS + ii[m]]);
          sL[ii[m] * LS + j] = ((ii[m] == j ? iq_old : 0.0) - wsym * iq_old * iq_old) * iom;
        }
      }
    }
    const long long knext = k0 + 1;
    // Adam on theta inside the time loop (PSMFRecursive, psmf.py:299-304,224-242)
    if (tl) {
      if (p.recursive && (knext % p.update_every) == 0) {
        const double kk = (double)knext;
        const double lr = p.lr_steps > 0.0 ? p.lr * ps_pow(p.lr_end / p.lr, kk / p.lr_steps) : p.lr;
        if (p.recursive == 2) {           // plain SGD (psmf.py:244-248)
          theta = fmax(theta - lr * gsum, 0.0);
        } else {
          am = p.b1 * am + (1.0 - p.b1) * gsum;
          av = p.b2 * av + (1.0 - p.b2) * gsum * gsum;
          const double mh = am / (1.0 - ps_pow(p.b1, kk));
          const double vh = av / (1.0 - ps_pow(p.b2, kk));
          theta = fmax(theta - lr * mh / (sqrt(vh) + 1e-8), 0.0);
        }
        gsum = 0.0;
      }
    }
    if (vl && p.mu_hist) p.mu_hist[(size_t)(knext - p.series_t0) * r + tid] = mu_new;
    ee_last = ee; s_done = s0; eta_done = eta0; N_done = N;
    // ---- everything the NEXT step needs (index knext + 1) ----
    const double qs = p.q_sched ? p.q_sched[knext + 1 - p.series_t0] : 1.0;
    if (p.rho_sched) rho = p.rho_sched[knext + 1 - p.series_t0];
    if (vl) {
      double mb = mu_new, f = 1.0;
      if (p.dyn_kind == 1) {   // cos(2 pi theta t + x)
        const double arg = 2.0 * M_PI * theta * (double)(knext + 1) + mu_new;
        mb = ps_cos(arg);
        f = -ps_sin(arg);
      }
      s_mub[tid] = mb;
      s_f[tid] = f;
      mub_t = mb;
    }
    ps_bar();
    PS_PROF(5);        // gradient, scalars, r x r updates, Adam, mu_bar
    double part = 0.0, gp = 0.0;
    if constexpr (BIG) {
      PS_BIG_FOR({
        if (v_) {
          const double pv = p.coef_update ? pscale * (0.5 * (sPp[a_] + sPp[at_])) : sPb[a_];      // (sL is being rewritten: Pbar of the step from its own image)
          const double pb = p.pbar_predict ? s_f[i_] * pv * s_f[j] + qs * sQ[a_] : pv;
          sPb[a_] = pb;
          part += sV[a_] * s_mub[i_];
          gp += sG[a_] * pb;
        }
      })
    }
#pragma unroll
    for (int m = 0; m < (BIG ? 0 : M); ++m) {
      if (val[m]) {
        const double pb = p.pbar_predict ? s_f[ii[m]] * Pv[m] * s_f[j] + qs * Qv[m] : Pv[m];
        Pbv[m] = pb;
        part += Vv[m] * s_mub[ii[m]];
        if (!MASKED) gp += Gv[m] * pb;
      }
    }
    if (!MASKED && p.eta_full) {      // <G, Pbar>: the waves' shares go out with the column reduction's own barriers
      const double x = ps_wave_sum(worker ? gp : 0.0);
      if (lane == 0) s4[wv] = x;
    }
    hub_col_reduce<RPAD, NWK>(worker, part, s_red, s_vec);        // s_vec = V mu_bar
    const double sN = ps_wave_sum(lane < r ? s_mub[lane] * s_vec[lane] : 0.0);
    double eta = rho * p.rho_mean;
    if constexpr (MASKED) {
      // ---- phase C: the masked Gram of the NEXT step, summed by the row workgroups slice by slice (pstep_rows); it arrives as
      //      tagged granules: every wave of the loop re-reads its elements until their tags carry the epoch ----
      {
        const u64* gsl = reinterpret_cast<const u64*>(q.gslice);
        constexpr int NEL = (NTT * 256 + 1 + NLT - 1) / NLT;      // elements per thread
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        int stop = 0;
        for (;;) {
          bool ok = true;
          u64 lo[NEL], hi[NEL];
#pragma unroll
          for (int k = 0; k < NEL; ++k) {
            const int e = min(tid + k * NLT, q.nge - 1);
            lo[k] = gran_load(gsl + 2 * e);
            hi[k] = gran_load(gsl + 2 * e + 1);
          }
#pragma unroll
          for (int k = 0; k < NEL; ++k) {
            ok &= (unsigned)(lo[k] >> 32) == epoch && (unsigned)(hi[k] >> 32) == epoch;
            const int e = tid + k * NLT;
            if (e < q.nge) s_gm[e] = __hiloint2double((int)(unsigned)hi[k], (int)(unsigned)lo[k]);
          }
          if (__all((int)ok)) break;
          __builtin_amdgcn_s_sleep(2);
          if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > PSTEP_SPIN_TICKS) { stop = 2; break; }
        }
        if (stop && lane == 0) *reinterpret_cast<volatile int*>(s_ctl) = stop;
      }
      ps_bar();
      PS_PROFM(9);      // masked: the reduced Gram of the next step is in LDS
      if (lds_word(s_ctl) != 0) break;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (val[m]) {
          Gv[m] = s_gm[gm_index(ii[m], j)];
          gp += Gv[m] * Pbv[m];
          sG[ii[m] * LS + j] = Gv[m];
        }
      }
      nobs = s_gm[NTT * 256];
      double x = ps_wave_sum(worker ? gp : 0.0);
      if (lane == 0) s4[wv] = x;
      ps_bar();
      eta = (rho * nobs + ((s4[0] + s4[1]) + (s4[2] + s4[3]))) / dd;      // divided by d, not by the observed count (PSMF.py:77)
      PS_PROFM(10);     // masked: reduced Gram gathered, eta
      if (tid == 0 && p.sc_hist) {
        long long tr_ = knext - p.series_t0;
        if (tr_ > (long long)p.mask_rows - 1) tr_ = (long long)p.mask_rows - 1;      // (the step after the last of the series: never run)
        if (knext - p.series_t0 <= (long long)p.mask_rows - 1) { p.sc_hist[2 * tr_] = sN; p.sc_hist[2 * tr_ + 1] = eta; }
      }
    } else if (p.eta_full) {
      eta += (((s4[0] + s4[1]) + (s4[2] + s4[3])) + (BIG ? s4[4] + s4[5] : 0.0)) / dd;
    }
    PS_PROF(6);        // V mu_bar, s, eta
    const double Nn = sN + eta;
    const double iNn = fast_rcp(Nn);
    const double kap_n = fast_rcp(rho + sN);
    if (vl) {
      w_t = s_vec[tid];
      s_w[tid] = w_t;
      s_wn[tid] = w_t * iNn;
    }
    // packet of the next step, by wave 0 alone and at once: mu_bar and w / N were written by this wave (r <= 32 < 64: LDS operations
    // of one wave complete in order), so nothing on the way to the row workgroups waits for a workgroup barrier
    if (wv == 0 && s + 1 < q.n_steps) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int gm_ = 0; gm_ < (NG + 63) / 64; ++gm_) {
        const int gI = lane + 64 * gm_;
        if (gI < NG) {
          const int e = gI >> 1;
          const double v = gI == NG - 1 ? 0.0 : (e < RPAD ? s_mub[e] : s_wn[e - RPAD]);
          const unsigned half = gI == NG - 1 ? 0u : ((gI & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v));
#pragma unroll
          for (int rep = 0; rep < PSTEP_PKT_REP; ++rep) gran_store(q.pkt + (size_t)rep * (PSTEP_REP_STRIDE / 8) + gI, epoch + 1u, half);
        }
      }
    }
    if (!dual) {      // inversions one after the other: the solve wave starts from Pbar
      if constexpr (BIG) {
        PS_BIG_FOR({ if (v_) sL[a_] = sPb[a_]; })
      } else {
#pragma unroll
        for (int m = 0; m < M; ++m) if (val[m]) sL[ii[m] * LS + j] = Pbv[m];
      }
    }
    if (tid == 0) {
      s_sc[0] = kap_n;
      s_sc[1] = fast_rcp(BIG ? sQ[0] : Qv[0]);
    }
    carried = dual;
    N0 = Nn; kappa0 = kap_n; s0 = sN; eta0 = eta; k0 = knext;
    n_done = s + 1;
    ps_bar();
    PS_PROF(7);        // w / N, solve operands, barrier
    PS_PROF(8);        // packet out
  }
  PS_PROF_OUT(0, 16, tid == 0);

  // ---------------- end of the launch: DevState as the two-launch engine leaves it ----------------
  const int stop = lds_word(s_ctl), badp = lds_word(s_ctl + 2);
  if ((stop != 0 || badp != 0) && tid == 0) {
    for (int rep = 0; rep < PSTEP_PKT_REP; ++rep) gran_store(q.pkt + (size_t)rep * (PSTEP_REP_STRIDE / 8) + (NG - 1), PSTEP_ABORT_TAG, 1u);        // the row workgroups leave at their next poll
    if (st->err == 0) st->err = badp ? (int)(k0 + 1) : -8;
  }
  if constexpr (BIG) {
    PS_BIG_FOR({
      if (v_) {
        const int idx = i_ * r + j;
        st->V[idx] = sV[a_];
        if (n_done > 0) st->P[idx] = p.coef_update ? pscale_last * (0.5 * (sPp[a_] + sPp[at_])) : sPb[a_];
        st->G[idx] = sG[a_];
        st->Q[idx] = sQ[a_];
        st->Pbar[idx] = sPb[a_];
        if (n_done > 0 && dual) { st->Lbar[idx] = sL[a_]; st->XpY[idx] = 0.5 * (sW[a_] + sW[at_]); }
        if (n_done > 0 && p.coef_update) st->Pplus[idx] = sPp[a_];
      }
    })
  }
#pragma unroll
  for (int m = 0; m < (BIG ? 0 : M); ++m) {
    if (val[m]) {
      const int idx = ii[m] * r + j;
      st->V[idx] = Vv[m];
      if (n_done > 0) st->P[idx] = Pv[m];
      if (!MASKED) st->G[idx] = Gv[m];
      else if (n_done > 0) { q.mg_out[idx] = Gv[m]; st->GR[idx] = Gv[m]; }      // the reduced masked Gram of the next step, where the next launch looks for it (GR: debug copy)
      st->Q[idx] = Qv[m];
      st->Pbar[idx] = Pbv[m];
      if (n_done > 0 && dual) { st->Lbar[idx] = sL[ii[m] * LS + j]; st->XpY[idx] = 0.5 * (sW[ii[m] * LS + j] + sW[j * LS + ii[m]]); }
      if (n_done > 0 && p.coef_update) st->Pplus[idx] = sPp[ii[m] * LS + j];
    }
  }
  if (vl) {
    st->mu[tid] = mu_new;
    st->mu_bar[tid] = mub_t;
    st->w[tid] = w_t;
    st->wN[tid] = s_wn[tid];
  }
  if (tl) {
    p.theta[tid] = theta;
    p.gradsum[tid] = gsum;
    p.adam_m[tid] = am;
    p.adam_v[tid] = av;
  }
  if (MASKED && tid == 0 && n_done > 0) {
    q.mg_out[r * r] = nobs;
    q.mg_out[r * r + 2] = eta0 * dd - rho * nobs;          // <G_m, Pbar> as ONE share (the consumer sums mg_ntr of them in order)
    for (int w = 1; w < q.mg_ntr; ++w) q.mg_out[r * r + 2 + w] = 0.0;
  }
  if (tid == 0) {
    st->k = k0;
    st->kq = k0;
    if (n_done > 0) st->ns_valid = dual ? 7 : 0;
    st->rho = rho;
    st->lam = lam;
    st->s = s0; st->eta = eta0; st->N = N0; st->kappa = kappa0;
    if (n_done > 0) {
      st->phi = phi; st->omega = omega; st->ee = ee_last;
      st->s_done = s_done; st->eta_done = eta_done; st->N_done = N_done;
    }
  }
}

// NP: row passes of a row workgroup, all of them unrolled and executed (rows beyond the workgroup's share are masked): no branch
// between the passes, so their dot products, lane sums and updates interleave.  NT: threads per workgroup -- 512 in every instance
// (RPAD = 64, 33 <= r <= 48: the hub's r x r matrices live in LDS, pstep_hub BIG; a 256-thread form with everything in registers
// spilled 2.7 KB per lane and was removed).
template <typename T, int RPAD, int NP, int NT, bool MASKED>
__global__ __launch_bounds__(NT) void psmf_pstep_k(PstepParams q) {
  extern __shared__ __attribute__((aligned(16))) char ps_smem[];
  if (blockIdx.x == 0) pstep_hub<RPAD, NT, MASKED>(q, ps_smem);
  else pstep_rows<T, RPAD, NT, NP, MASKED>(q, ps_smem);
}

typedef void (*pstep_fn_t)(PstepParams);
constexpr int pstep_nt(int rpad) { return 512; }
// RPAD = 64 (16 lanes per row, 32 rows per pass): a fourth variant of 16 passes = 512 rows per workgroup, so that d_local = 1e5 fits the
// 204 partial rows the hub's fan-in takes at r >= 40
constexpr int PSTEP_NPMAX_BIG = 16;
int pstep_np_variant(int rpad, int np) { return np <= 4 ? 4 : (np <= 8 ? 8 : (np <= PSTEP_NPMAX || rpad <= 32 ? PSTEP_NPMAX : PSTEP_NPMAX_BIG)); }

template <int RPAD, bool MASKED>
pstep_fn_t pstep_kernel_r(bool f64, int np) {
  constexpr int NT = pstep_nt(RPAD);
  if (f64) return np <= 4 ? psmf_pstep_k<double, RPAD, 4, NT, MASKED> : (np <= 8 ? psmf_pstep_k<double, RPAD, 8, NT, MASKED> : psmf_pstep_k<double, RPAD, PSTEP_NPMAX, NT, MASKED>);
  return np <= 4 ? psmf_pstep_k<float, RPAD, 4, NT, MASKED> : (np <= 8 ? psmf_pstep_k<float, RPAD, 8, NT, MASKED> : psmf_pstep_k<float, RPAD, PSTEP_NPMAX, NT, MASKED>);
}
pstep_fn_t pstep_kernel(int rpad, bool f64, int np, bool masked) {
  switch (rpad) {
    case 8: return masked ? pstep_kernel_r<8, true>(f64, np) : pstep_kernel_r<8, false>(f64, np);
    case 16: return masked ? pstep_kernel_r<16, true>(f64, np) : pstep_kernel_r<16, false>(f64, np);
    case 32: return masked ? pstep_kernel_r<32, true>(f64, np) : pstep_kernel_r<32, false>(f64, np);
    case 64:      // 33 <= r <= 48, unmasked (the hub's LDS-resident layout)
      if (masked) return nullptr;
      if (np > PSTEP_NPMAX) return f64 ? psmf_pstep_k<double, 64, PSTEP_NPMAX_BIG, 512, false> : psmf_pstep_k<float, 64, PSTEP_NPMAX_BIG, 512, false>;
      return pstep_kernel_r<64, false>(f64, np);
  }
  return nullptr;
}
// dynamic LDS of a launch: more than half of a CU's 160 KB (one workgroup per compute unit); RPAD = 64: the hub's four 64 x 65 images
size_t pstep_lds_bytes(int rpad) { return rpad > 32 ? (size_t)148 * 1024 : (size_t)84 * 1024; }

int pstep_rpad(int r) { return r <= 8 ? 8 : (r <= 16 ? 16 : (r <= 32 ? 32 : 64)); }

}  // namespace

// 33 <= r <= 48: the hub keeps every r x r matrix in LDS (pstep_hub, BIG) -- with 16 elements of five 64 x 64 matrices per worker thread in
// registers beside the 3 x 3 tile sweeps it needed more than a wave can hold (hipcc: 2.7 KB of scratch per lane, measured).  r > 48 (4 x 4
// tiles, seven 64 x 65 images = 233 KB of LDS) and masked handles at r > 32 keep the launches per timestep (psmf_kernels.hip).
static pstep_fn_t pstep_kernel_any(int rpad, bool f64, int np, bool masked) { return pstep_kernel(rpad, f64, np, masked); }

bool pstep_plan(int d_local, int r, int n_cu, bool storage_f64, bool masked, PstepPlan* out) {
  if (r < 1 || r > 48 || (r > 32 && masked) || d_local < 1 || n_cu < 2) return false;
  const int rpad = pstep_rpad(r);
  const int rpw = pstep_nt(rpad) / (rpad / 4);
  int nwg = (d_local + rpw - 1) / rpw;
  if (nwg > n_cu - 1) nwg = n_cu - 1;
  int rows = (d_local + nwg - 1) / nwg;
  int np = (rows + rpw - 1) / rpw;
  if (np > (rpad > 32 ? PSTEP_NPMAX_BIG : PSTEP_NPMAX)) return false;
  (void)storage_f64;
  np = pstep_np_variant(rpad, np);       // the kernel instance runs exactly this many passes: fewer, fuller workgroups
  rows = np * rpw;
  nwg = (d_local + rows - 1) / rows;
  {      // the hub's fan-in: (threads of its four fan-in waves / pairs of columns, at most 24) segments x PSTEP_FANIN_ROWS rows each
    const int npair = ((r + 1 + 1) & ~1) / 2;
    int S = (4 * 64) / npair;
    const int cap = rpad > 32 ? 12 : 24;      // 3 x the lanes per column of the second level (pstep_hub: LPC)
    if (S > cap) S = cap;
    if (nwg > S * (rpad > 32 ? PSTEP_FANIN_ROWS_BIG : PSTEP_FANIN_ROWS)) return false;
  }
  out->n_row_wg = nwg;
  out->rows_per_wg = rows;
  out->np = np;
  out->ncol2 = (r + 1 + 1) & ~1;
  const size_t fl = (((size_t)nwg * sizeof(unsigned)) + 15) & ~(size_t)15;
  // the zeroed block: | flags of the (h, ee) hand-off | packet x PSTEP_PKT_REP | flags of the Gram partials x PSTEP_GF_REP | granules of the reduced Gram |
  static_assert(PSTEP_PKT_MAX * 8 <= PSTEP_REP_STRIDE, "a packet fits its replica slot");
  const int ntg = rpad > 16 ? 2 : 1;
  const int nge = masked ? ntg * (ntg + 1) / 2 * 256 + 1 : 0;
  const size_t gs = ((size_t)2 * nge * sizeof(unsigned long long) + 15) & ~(size_t)15;
  const size_t fl4k = (fl + 4095) & ~(size_t)4095;
  if ((size_t)nwg * sizeof(unsigned) > (size_t)PSTEP_REP_STRIDE) return false;
  out->off_pkt = fl4k;
  out->off_gflags = fl4k + (size_t)PSTEP_PKT_REP * PSTEP_REP_STRIDE;
  out->off_sflags = 0;
  out->off_gslice = out->off_gflags + (masked ? (size_t)PSTEP_GF_REP * PSTEP_REP_STRIDE : 0);
  out->zero_bytes = out->off_gslice + gs;
  out->off_part = (out->zero_bytes + 255) & ~(size_t)255;
  size_t end = out->off_part + (size_t)nwg * out->ncol2 * sizeof(double);
  out->nge = nge; out->slice_len = 0; out->off_gpart = 0;
  if (masked) {
    out->slice_len = (nge + nwg - 1) / nwg;
    out->off_gpart = (end + 255) & ~(size_t)255;
    end = out->off_gpart + (size_t)nwg * nge * sizeof(double);
  }
  out->total_bytes = end;
  return true;
}

hipError_t pstep_init() {
  for (int rpad = 8; rpad <= 64; rpad *= 2)
    for (int f = 0; f < 4; ++f)
      for (int np = 4; np <= 16; np += 4) {      // (every instance is reached by one of these)
        if (rpad == 64 && (f & 2)) continue;
        const hipError_t e = hipFuncSetAttribute((const void*)pstep_kernel_any(rpad, (f & 1) != 0, np, (f & 2) != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pstep_lds_bytes(rpad));
        if (e != hipSuccess) return e;
      }
  return hipSuccess;
}

hipError_t pstep_launch(const PstepParams& q, bool storage_f64, hipStream_t stream) {
  const int rpad = pstep_rpad(q.sp.r);
  pstep_fn_t fn = pstep_kernel_any(rpad, storage_f64, q.np, q.masked != 0);
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(q.n_row_wg + 1), dim3(pstep_nt(rpad)), pstep_lds_bytes(rpad), stream, q);
  return hipGetLastError();
}

}  // namespace psmf
