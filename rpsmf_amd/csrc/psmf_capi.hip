// C ABI of libpsmf_hip.so (include/psmf_hip.h): host-side orchestration of the large-d engine.
// One handle = one HIP stream, one device-resident filter (or row shard), one hipGraph of
// per-step launches replayed over the series.  No torch, no hipBLAS: plain HIP + RCCL.
#include "../../include/psmf_hip.h"
#include "psmf_kernels.hip"
#include "psmf_masked.hip"
#include <chrono>
#include "psmf_block.hip"
#include "psmf_blk3.hip"
#include "psmf_blk16.hip"
#include "psmf_blk32.hip"
#include "psmf_bulk.hip"
#include "psmf_rotate.hip"
#include "psmf_pstep.h"       // persistent per-step engine: its kernels are a translation unit of their own (psmf_pstep.hip)

#include <rccl/rccl.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

using psmf::DevState;
using psmf::StepParams;

namespace {

thread_local std::string g_create_error;

struct Geometry {
  int vec, nv, rp, gs, rpp, rpad, nt;
  int n_sweep_wg, rows_per_wg, ps;
  size_t sweep_lds;
};

#ifndef PSMF_SWEEP_UNROLL
#define PSMF_SWEEP_UNROLL 4
#endif
constexpr int kUnroll = PSMF_SWEEP_UNROLL;      // row passes (16-byte loads per lane) in flight in the sweep
constexpr int kGramWG = 128;

}  // namespace

// Environment switches ("Switches"): read ONCE per handle, at psmf_create -- tests flip them between
// handles of one process; nothing on the per-block host path calls getenv.
struct Switches {
  bool bulk2 = true, filter3 = true, filter4 = true, filter6 = true, filter7 = true, filter6_dual = true, block_dual = true, block_flags = true, block_chain = true, block_pipe = true;
  bool force_collective = false;
  bool host_comm_flags = false;     // PSMF_HOST_COMM_FLAGS=1: device-flag hand-off (and chained filter launches) under a host-mediated communicator too
  bool serial_wide = true, step_dual = true, tail_reduce = true, wgram_mfma = true, pstep_big = true;
  int sweep_threads = 512;
  double ns_far4 = 0.6;            // filter4 / filter4s: residual at which a Newton-Schulz start is given up (PSMF_NS_FAR4; PSMF_NS_FAR, when set, rules both)
  bool ns_far_set = false;
  bool step_persistent = true;      // per-step engine: one persistent launch per run (psmf_pstep.hip) where it applies; PSMF_STEP_PERSISTENT=0: two launches per timestep
  static bool off(const char* name) { const char* e = getenv(name); return e && atoi(e) == 0; }
  void read() {
    bulk2 = !off("PSMF_BULK2"); filter3 = !off("PSMF_FILTER3"); filter4 = !off("PSMF_FILTER4"); filter6 = !off("PSMF_FILTER6"); filter7 = !off("PSMF_FILTER7"); filter6_dual = !off("PSMF_FILTER6_DUAL"); block_dual = !off("PSMF_BLOCK_DUAL");
    block_flags = !off("PSMF_BLOCK_FLAGS"); block_chain = !off("PSMF_BLOCK_CHAIN"); block_pipe = !off("PSMF_BLOCK_PIPE");
    force_collective = getenv("PSMF_FORCE_COLLECTIVE") != nullptr;
    { const char* e = getenv("PSMF_HOST_COMM_FLAGS"); host_comm_flags = e && atoi(e) != 0; }
    step_persistent = !off("PSMF_STEP_PERSISTENT");
    serial_wide = !off("PSMF_SERIAL_WIDE"); step_dual = !off("PSMF_STEP_DUAL"); tail_reduce = !off("PSMF_TAIL_REDUCE"); wgram_mfma = !off("PSMF_WGRAM_MFMA"); pstep_big = !off("PSMF_PSTEP_BIG");
    { const char* e = getenv("PSMF_SWEEP_THREADS"); sweep_threads = (e && atoi(e) == 256) ? 256 : 512; }
    if (const char* e = getenv("PSMF_NS_FAR4")) ns_far4 = atof(e);
    ns_far_set = getenv("PSMF_NS_FAR") != nullptr;
  }
};

struct psmf_filter {
  psmf_config cfg;
  Geometry geo;
  Switches sw;
  hipStream_t stream = nullptr;
  DevState* st = nullptr;
  void* C = nullptr;
  void* Y = nullptr;
  void* YP = nullptr;
  double* partials = nullptr;
  double* gpart = nullptr;
  double* thbuf = nullptr;     // theta | gradsum | adam_m | adam_v, th_cap doubles each
  size_t th_cap = 0;
  double* rho_rows = nullptr;  // d_local per-row diag(R) (cfg.nonuniform_R)
  double* rotU = nullptr;      // d x d: eigenvectors of a non-diagonal R in its columns (psmf_set_noise_rotation); series, C, y_hat are kept rotated
  void* rot_tmp = nullptr;     // staging of a rotation (the GEMM is out of place)
  size_t rot_tmp_bytes = 0;
  // masked filter (cfg.masked, psmf_masked.hip)
  uint8_t* mask = nullptr;     // T_cap x d_local observation mask (psmf_upload_mask)
  uint8_t* mmiss = nullptr;    // staging of the held-out mask for psmf_masked_metrics (mmiss_cap bytes)
  size_t mmiss_cap = 0;
  double* mg = nullptr;        // r*r + 1: masked Gram and observed count of the current step, summed over workgroups (and ranks)
  double* sc_hist = nullptr;   // T_cap x 2: (s_k, eta_k) of every step -- the bands are formed from them
  bool have_mask = false;
  double* sched = nullptr;     // rho_k | q_k schedules, sched_n doubles each (psmf_set_schedules)
  int64_t sched_n = 0;
  double* qmat = nullptr;      // Q_k matrices, (qmat_n + 1) x r x r (psmf_set_q_matrix_schedule)
  int64_t qmat_n = 0;
  double* mu_hist = nullptr;   // (T_cap + 1) x r
  hipStream_t fstream = nullptr;   // blocked engine, pipelined: the filter chain's own stream, pinned to reserved CUs (or nullptr)
  bool streams_concurrent = false;           // the filter stream's kernels run concurrently with the bulk stream's (probed at creation)
  // HIP-event timing of the chained filter launches (one per run): a ring of event pairs, read out at the next sync
  static constexpr int kTimedRuns = 256;
  hipEvent_t evK0[kTimedRuns] = {}, evK1[kTimedRuns] = {};
  hipEvent_t evC = nullptr;        // end of the chained filter launch: orders the handle's main stream (host reads of DevState) after it
  int evk_pending = 0;
  double kernel_ms_sum = 0.0;
  long long kernel_launches = 0;
  int reserved_cus = 0;
  int bulk_wgs = 256;          // workgroups of the streaming bulk kernels (one per CU of the bulk stream: a 257th would wait for a whole round)
  double* scratch = nullptr;   // sq-error partials / predict staging
  // blocked engine
  int engine = 1;              // 1 per-step, 2 blocked
  int block_steps = 0;         // B = RB - r
  bool q_iso = false;          // Q = q I with q > 0 as last uploaded (two-group block filter applies)
  double q_last = 0.0, p_diag_max = 0.0;      // Q[0][0] and max_i P[i][i] as last uploaded: the give-up policy of the Newton-Schulz starts (update_ns_policy)
  bool ns_far_env = false, ns_skip_env = false;
  double* Kpart = nullptr;
  double* Kmat = nullptr;
  double* Acoef = nullptr;     // 2 x RB x RM   (ping-pong across pipelined blocks)
  double* Bcoef = nullptr;     // 2 x RB x RB
  double* XGpart = nullptr;    // BLK_GRAM_WG x (RB + XGB) x XGB
  double* XG = nullptr;        // 2 x (RB + XGB) x XGB
  long long* flags = nullptr;  // device-flag hand-off of the pipelined blocks (psmf_block.hip): xg_seq, filt_seq, abort
  long long seq_next = 1;      // sequence number of the next block to be enqueued
  hipStream_t bulk = nullptr;  // Gram / cross-Gram / apply of the pipelined blocked engine
  hipEvent_t evF[4] = {}, evA[4] = {}, evX[4] = {}, evS = nullptr;
  size_t scratch_bytes = 0;
  int64_t T_cap = 0;
  StepParams sp;
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  int chunk = 0;
  // persistent per-step engine (psmf_pstep.hip): geometry of a launch and its communication block (flags | packet | partial rows)
  psmf::PstepPlan ps_plan = {};
  bool ps_ok = false;
  void* ps_comm = nullptr;
  long long* ps_prof = nullptr;    // PSMF_PSTEP_PROF=1 with a -DPSTEP_PROF build: per-phase clock sums of the last launch, printed at psmf_destroy
  long long ps_prof_steps = 0;
  long long ps_launches = 0;
  bool have_state = false;
  bool need_prep = true;
  int64_t k_done = 0;
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  bool use_coll = false;   // per-step all-reduce on (nranks > 1, or forced for single-GPU testing)
  // host-mediated collective (psmf_comm_init_host): the sum-all-reduce goes through a caller-supplied function
  psmf_allreduce_fn host_fn = nullptr;
  void* host_ctx = nullptr;
  double* host_buf = nullptr;      // pinned staging buffer, kHostBufElems doubles
  static constexpr size_t kHostBufElems = 8192;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int* err_host = nullptr;     // pinned, device-mapped: a one-thread kernel publishes the device error flag here
  int* err_host_dev = nullptr;
  std::string err;
  size_t elem() const { return cfg.storage == PSMF_F64 ? 8 : 4; }
};

namespace {

int fail(psmf_handle h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(h, PSMF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

#define NCCL_TRY(h, expr)                                                                  \
  do {                                                                                     \
    ncclResult_t e_ = (expr);                                                              \
    if (e_ != ncclSuccess)                                                                 \
      return fail(h, PSMF_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(e_));  \
  } while (0)

int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// Completion waits by polling: hipStreamSynchronize / hipEventSynchronize fall back to an interrupt wait that, on this
// stack, now and then returns ~70 ms after the work is done (seen as wall time without matching event time).
hipError_t spin_stream(hipStream_t s) {
  hipError_t e;
  while ((e = hipStreamQuery(s)) == hipErrorNotReady) { __builtin_ia32_pause(); }
  return e;
}
hipError_t spin_event(hipEvent_t ev) {
  hipError_t e;
  while ((e = hipEventQuery(ev)) == hipErrorNotReady) { __builtin_ia32_pause(); }
  return e;
}

// The one exchange of the sharded engines: in-place sum of `count` float64 on the device over all ranks.  RCCL on the given
// stream, or -- host-mediated communicator -- device -> pinned host -> caller's function -> device, synchronously.
int all_reduce_sum(psmf_filter* h, double* buf, size_t count, hipStream_t s) {
  if (h->host_fn) {
    if (count > psmf_filter::kHostBufElems) return fail(h, PSMF_ERR_ARG, "host all-reduce: message larger than the staging buffer");
    HIP_TRY(h, hipMemcpyAsync(h->host_buf, buf, count * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, spin_stream(s));
    if (h->host_fn(h->host_ctx, h->host_buf, (int64_t)count) != 0) return fail(h, PSMF_ERR_RCCL, "host all-reduce callback reported a failure");
    HIP_TRY(h, hipMemcpyAsync(buf, h->host_buf, count * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, spin_stream(s));        // the staging buffer is reused by the next call
    return PSMF_OK;
  }
  NCCL_TRY(h, ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, h->comm, s));
  return PSMF_OK;
}

typedef void (*sweep_fn_t)(StepParams);
typedef void (*serial_fn_t)(StepParams, int);

// loads in flight per lane of the row sweep.  The 256-thread instances of r > 32 (GS lanes x VEC elements > 32 columns) run ONE wave
// per SIMD -- their solve block keeps 3 x 3 / 4 x 4 tile arrays in a wave's 512 registers, and a kernel has one register allocation --
// so the row workgroups make up in depth what they lack in occupancy: 16 passes in flight instead of 4 (rows alone at d = 2e4, r = 40:
// 17.5 us -> see docs/MEASUREMENTS.md).
template <typename T, int GS, int NT>
constexpr int sweep_unroll() { return (NT == 256 && GS * (int)(16 / sizeof(T)) > 32) ? 16 : kUnroll; }

template <typename T, int NT>
sweep_fn_t sweep_for_gs(int gs) {
  switch (gs) {
    case 1: return psmf::psmf_sweep_solve<T, 1, kUnroll, NT>;
    case 2: return psmf::psmf_sweep_solve<T, 2, kUnroll, NT>;
    case 4: return psmf::psmf_sweep_solve<T, 4, kUnroll, NT>;
    case 8: return psmf::psmf_sweep_solve<T, 8, kUnroll, NT>;
    case 16: return psmf::psmf_sweep_solve<T, 16, sweep_unroll<T, 16, NT>(), NT>;
    case 32: return psmf::psmf_sweep_solve<T, 32, sweep_unroll<T, 32, NT>(), NT>;
  }
  return nullptr;
}

// threads per sweep workgroup (tuning knob PSMF_SWEEP_THREADS=256|512): 512 halves the number of
// per-workgroup partial rows the serial stage has to read at the same number of waves per CU
sweep_fn_t sweep_kernel(const psmf_filter* h) {
  const int gs = h->geo.gs;
  if (h->geo.nt == 512)
    return h->cfg.storage == PSMF_F64 ? sweep_for_gs<double, 512>(gs) : sweep_for_gs<float, 512>(gs);
  return h->cfg.storage == PSMF_F64 ? sweep_for_gs<double, 256>(gs) : sweep_for_gs<float, 256>(gs);
}

bool serial_wide(const psmf_filter* h) { return h->geo.rpad >= 64 && h->sw.serial_wide; }

serial_fn_t serial_kernel(const psmf_filter* h) {
  switch (h->geo.rpad) {
    case 8: return psmf::psmf_serial<8>;
    case 16: return psmf::psmf_serial<16>;
    case 32: return psmf::psmf_serial<32>;
    default: return serial_wide(h) ? psmf::psmf_serial_wide : psmf::psmf_serial<64>;
  }
}

void launch_sweep(psmf_filter* h) {
  const int grid = h->geo.n_sweep_wg + (h->cfg.coef_update ? 1 : 0);
  hipLaunchKernelGGL(sweep_kernel(h), dim3(grid), dim3(h->geo.nt), h->geo.sweep_lds, h->stream, h->sp);
}

void launch_serial(psmf_filter* h, int first) {
  hipLaunchKernelGGL(serial_kernel(h), dim3(1), dim3(serial_wide(h) ? psmf::SERIAL_WIDE_NT : psmf::serial_threads(h->geo.rpad)), 0, h->stream, h->sp, first);
}

// one filter step on the stream (captured into the graph or launched eagerly)
int enqueue_weighted_gram(psmf_filter* h);

int enqueue_serial_mgram(psmf_filter* h, int first);

int enqueue_step(psmf_filter* h) {
  if (h->sp.rho_rows && h->cfg.coef_update) {      // non-uniform diagonal R: this step's weighted Gram, before the sweep rewrites C
    const int rc = enqueue_weighted_gram(h);
    if (rc) return rc;
  }
  launch_sweep(h);
  if (h->use_coll) {
    if (!h->sp.tail_reduce) hipLaunchKernelGGL(psmf::psmf_reduce_partials, dim3(1), dim3(psmf::WG), 0, h->stream, h->sp);
    const int rc = all_reduce_sum(h, h->st->red, h->geo.ps, h->stream);
    if (rc) return rc;
  }
  if (h->cfg.masked) return enqueue_serial_mgram(h, 0);      // serial stage of this step beside the masked Gram of the next (psmf_masked.hip)
  launch_serial(h, 0);
  return PSMF_OK;
}

// one block of nb steps of the blocked engine: Gram, reduction, (all-reduce), coefficient-space filter, apply
void fill_block_params(psmf_filter* h, psmf::BlockParams& b, int64_t k0, int nb, int slot = 0) {
  memset(&b, 0, sizeof(b));
  b.last = 1;            // standalone block; the pipelined loop clears it for all but a run's last block
  b.sp = h->sp;
  b.Kpart = h->Kpart; b.K = h->Kmat;
  b.Acoef = h->Acoef + (size_t)slot * psmf::RB * psmf::RM;
  b.Bcoef = h->Bcoef + (size_t)slot * psmf::RB * psmf::RB;
  b.XGpart = h->XGpart;
  b.k0 = k0; b.nb = nb;
  b.gram_rows = (h->cfg.d_local + psmf::BLK_GRAM_WG - 1) / psmf::BLK_GRAM_WG;
}

void launch_blk_gram(psmf_filter* h, const psmf::BlockParams& b, hipStream_t stream = nullptr) {
  if (!stream) stream = h->stream;
  if (h->cfg.storage == PSMF_F64)
    hipLaunchKernelGGL(psmf::psmf_blk_gram_mfma<double>, dim3(psmf::BLK_GRAM_WG), dim3(psmf::WG), 0, stream, b);
  else
    hipLaunchKernelGGL(psmf::psmf_blk_gram_mfma<float>, dim3(psmf::BLK_GRAM_WG), dim3(psmf::WG), 0, stream, b);
  hipLaunchKernelGGL(psmf::psmf_blk_reduce, dim3(psmf::RB * psmf::RB / 128), dim3(128), 0, stream, b, (int)psmf::BLK_GRAM_WG);
}

// streaming bulk kernels (psmf_bulk.hip): float32 storage, d_local a multiple of 4, 16 <= r <= 32
bool blk_bulk2_ok(const psmf_filter* h) {
  return h->sw.bulk2 && h->cfg.storage == PSMF_F32 && (h->cfg.d_local % 4) == 0 && h->cfg.r <= 32 && (h->geo.rp % 4) == 0;
}

void launch_blk_xgram(psmf_filter* h, const psmf::BlockParams& x, double* xg, hipStream_t stream) {
  const size_t xg_elems = (size_t)(psmf::RB + psmf::XGB) * psmf::XGB;
  if (blk_bulk2_ok(h)) {
    const int nct = (h->block_steps + 15) / 16;
    const size_t lds = psmf::blk_xgram2_lds_bytes();
    if (nct <= 2) {
      hipLaunchKernelGGL(psmf::psmf_blk_xgram2<2>, dim3(h->bulk_wgs), dim3(psmf::BK_NT), lds, stream, x);
      hipLaunchKernelGGL(psmf::psmf_blk_xreduce2<2>, dim3(6 * 2 * 256 / 32), dim3(256), 0, stream, (const double*)x.XGpart, xg, h->bulk_wgs);
    } else {
      hipLaunchKernelGGL(psmf::psmf_blk_xgram2<3>, dim3(h->bulk_wgs), dim3(psmf::BK_NT), lds, stream, x);
      hipLaunchKernelGGL(psmf::psmf_blk_xreduce2<3>, dim3(7 * 3 * 256 / 32), dim3(256), 0, stream, (const double*)x.XGpart, xg, h->bulk_wgs);
    }
    return;
  }
  if (h->cfg.storage == PSMF_F64)
    hipLaunchKernelGGL(psmf::psmf_blk_xgram_mfma<double>, dim3(psmf::BLK_GRAM_WG), dim3(psmf::WG), 0, stream, x);
  else
    hipLaunchKernelGGL(psmf::psmf_blk_xgram_mfma<float>, dim3(psmf::BLK_GRAM_WG), dim3(psmf::WG), 0, stream, x);
  hipLaunchKernelGGL(psmf::psmf_blk_xreduce, dim3((int)(xg_elems / 128)), dim3(128), 0, stream, x, xg, (int)psmf::BLK_GRAM_WG);
}

bool blk_small_dual(const psmf_filter* h);
bool blk_use_filter3(const psmf_filter* h) { return h->sw.filter3 && !blk_small_dual(h); }

bool blk_dual_ok(const psmf_filter* h) {
  // The two-inversion kernels (filter3, filter3s, filter2) read rho and q ONCE per block: per-step R_k / Q_k schedules
  // (psmf_set_schedules; the reference reads R[k], Q[k] every step, psmf.py:115,123,141) go to the general kernel.
  return h->sw.block_dual && h->q_iso && h->cfg.coef_update && h->cfg.pbar_predict && h->cfg.eta_full &&
         h->cfg.dyn_kind == PSMF_DYN_RANDOM_WALK && !h->sp.rho_sched && !h->sp.q_sched;
}

// filter4 (psmf_blk4.hip): the role-specialised kernel for diagonal-Jacobian dynamics -- cos-phase, unscaled sinusoid, and the
// random walk when R_k / Q_k schedules keep it off filter3 -- full filter, Q = q I, r <= 32; the recursive classes included
// filter6 (psmf_blk16.hip): the general block filter for r <= 16, role-specialised -- whatever filter3s / filter5 do not take,
// INCLUDING what filter4s would (measured at r = 10, d = 2e4: cos-phase full filter 117 k timesteps/s on filter4s, 316 k on
// filter6; its recursive form 196 k against 214 k)
bool blk_small_ok(const psmf_filter* h) { return h->sw.filter6 && h->cfg.r <= psmf::F6_RMAX; }

// ... and the default model too (random walk, Q = q I; PSMF_FILTER6_DUAL=0: filter3s), psmf_blk_filter6d
bool blk_small_dual(const psmf_filter* h) { return h->sw.filter6_dual && blk_small_ok(h) && blk_dual_ok(h); }

bool blk_seq_ok(const psmf_filter* h) {
  if (blk_small_ok(h)) return false;
  const int kd = h->cfg.dyn_kind;
  const bool diag_dyn = kd == PSMF_DYN_RANDOM_WALK || kd == PSMF_DYN_COS_PHASE || (kd == PSMF_DYN_SINUSOID && !(h->cfg.dyn_flags & 1));
  return h->sw.filter4 && h->sw.filter3 && h->sw.block_dual && h->q_iso && h->cfg.coef_update && h->cfg.pbar_predict && h->cfg.eta_full && diag_dyn &&
         h->cfg.r <= 32 && h->cfg.recursive != 2;      // (in-loop SGD: the kernels with dyn_adam_step carry it, filter4 / filter5 have an Adam step of their own)
}

// filter5: the simplified hook configuration (no coefficient update, eta = tr(R) / d, P_bar = P) with diagonal-Jacobian dynamics
bool blk_simpl_ok(const psmf_filter* h) {
  const int kd = h->cfg.dyn_kind;
  const bool diag_dyn = kd == PSMF_DYN_RANDOM_WALK || kd == PSMF_DYN_COS_PHASE || (kd == PSMF_DYN_SINUSOID && !(h->cfg.dyn_flags & 1));
  return h->sw.filter4 && h->sw.filter3 && !h->cfg.coef_update && !h->cfg.eta_full && !h->cfg.pbar_predict && diag_dyn && h->cfg.r <= 32 &&
         !h->sp.q_sched && h->cfg.recursive != 2;
}

// The ONE place that decides which kernel advances the coefficient-space state of a block: launch_blk_filter switches on it
// and psmf_filter_kernel reports it (tests and bench.py quote that name as evidence of what ran).  Values = the codes of
// psmf_filter_kernel in include/psmf_hip.h.
enum FilterKernel { FK_STEP = 0, FK_GENERAL = 1, FK_FILTER2 = 2, FK_FILTER3 = 3, FK_FILTER3S = 4, FK_FILTER4 = 5, FK_FILTER4S = 6,
                    FK_FILTER5 = 7, FK_FILTER6 = 8, FK_FILTER6D = 9, FK_FILTER7 = 10, FK_PSTEP = 11 };

// Can the handle's next run go through the persistent per-step kernel?  (one rank, uniform diagonal R, random walk or cos-phase
// dynamics, r <= 32, rows that fit the row workgroups' registers, unmasked or the masked PSMF / rPSMF filter; everything else keeps
// the two -- masked: three -- launches per timestep)
bool pstep_usable(const psmf_filter* h) {
  return h->engine == 1 && h->ps_ok && h->sw.step_persistent && !h->use_coll && !h->host_fn && !h->sp.rho_rows &&
         (h->cfg.masked == 0 || (h->cfg.masked == 1 && h->have_mask)) &&        // masked PSMF / rPSMF; MLE-SMF and TMF keep the two launches
         !h->cfg.nonuniform_R && h->cfg.dyn_kind <= PSMF_DYN_COS_PHASE && !h->sp.solve_lds && !h->sp.q_mat;
}

FilterKernel select_filter_kernel(const psmf_filter* h) {
  if (h->engine != 2) return pstep_usable(h) ? FK_PSTEP : FK_STEP;
  if (blk_simpl_ok(h)) return FK_FILTER5;
  const bool dual3 = blk_dual_ok(h) && blk_use_filter3(h);
  if (!dual3 && blk_seq_ok(h)) return h->cfg.r > 16 ? FK_FILTER4 : FK_FILTER4S;
  if (blk_small_dual(h)) return FK_FILTER6D;      // random walk, Q = q I at r <= 16: filter6 with the two inversions side by side
  if (dual3) return h->cfg.r > 16 ? FK_FILTER3 : FK_FILTER3S;
  if (blk_dual_ok(h)) return FK_FILTER2;
  if (blk_small_ok(h)) return FK_FILTER6;
  // 17 <= r <= 32, whatever is left (dense Jacobians, a general Q, ...): filter6's design on 2 x 2 tiles (psmf_blk32.hip)
  return (h->sw.filter7 && h->cfg.r > psmf::F6_RMAX && h->cfg.r <= 32) ? FK_FILTER7 : FK_GENERAL;
}

void launch_blk_filter(psmf_filter* h, const psmf::BlockParams& b, hipStream_t stream = nullptr) {
  if (!stream) stream = h->stream;
  const size_t lds3 = psmf::blk_filter3_lds_bytes(), lds = psmf::blk_filter_lds_bytes();
  const FilterKernel fk = select_filter_kernel(h);
  switch (fk) {
    case FK_FILTER5: hipLaunchKernelGGL(psmf::psmf_blk_filter5, dim3(1), dim3(psmf::F3_NT), lds3, stream, b); return;
    case FK_FILTER4:
    case FK_FILTER4S: {
      // filter4's fallback is the wave-local sweep (~5 us, five to six iterations' worth; filter3's LDS sweep: 15 us): a start
      // beyond ||R||_F = 0.6 is cheaper swept than iterated (PSMF_NS_FAR4)
      psmf::BlockParams b4 = b;
      if (!h->sw.ns_far_set) b4.sp.ns_far2 = h->sw.ns_far4 * h->sw.ns_far4;
      if (fk == FK_FILTER4) hipLaunchKernelGGL(psmf::psmf_blk_filter4, dim3(1), dim3(psmf::F3_NT), lds3, stream, b4);
      else hipLaunchKernelGGL(psmf::psmf_blk_filter4s, dim3(1), dim3(psmf::F3_NT), lds3, stream, b4);
      return;
    }
    case FK_FILTER6D: {
      psmf::BlockParams b2 = b;
      b2.dual6 = 1;
      hipLaunchKernelGGL(psmf::psmf_blk_filter6d, dim3(1), dim3(psmf::WG), lds, stream, b2);
      return;
    }
    case FK_FILTER3: hipLaunchKernelGGL(psmf::psmf_blk_filter3, dim3(1), dim3(psmf::F3_NT), lds3, stream, b); return;
    case FK_FILTER3S: hipLaunchKernelGGL(psmf::psmf_blk_filter3s, dim3(1), dim3(psmf::F3_NT), lds3, stream, b); return;
    case FK_FILTER2: {
      const size_t lds2 = psmf::blk_filter2_lds_bytes();
      switch (h->geo.rpad) {
        case 8: hipLaunchKernelGGL(psmf::psmf_blk_filter2<8>, dim3(1), dim3(2 * psmf::WG), lds2, stream, b); break;
        case 16: hipLaunchKernelGGL(psmf::psmf_blk_filter2<16>, dim3(1), dim3(2 * psmf::WG), lds2, stream, b); break;
        default: hipLaunchKernelGGL(psmf::psmf_blk_filter2<32>, dim3(1), dim3(2 * psmf::WG), lds2, stream, b); break;
      }
      return;
    }
    case FK_FILTER6: hipLaunchKernelGGL(psmf::psmf_blk_filter6, dim3(1), dim3(psmf::WG), lds, stream, b); return;
    case FK_FILTER7: hipLaunchKernelGGL(psmf::psmf_blk_filter7, dim3(1), dim3(psmf::WG), lds, stream, b); return;
    case FK_GENERAL:
    case FK_STEP:
    case FK_PSTEP:
      break;
  }
  switch (h->geo.rpad) {
    case 8: hipLaunchKernelGGL(psmf::psmf_blk_filter<8>, dim3(1), dim3(psmf::WG), lds, stream, b); break;
    case 16: hipLaunchKernelGGL(psmf::psmf_blk_filter<16>, dim3(1), dim3(psmf::WG), lds, stream, b); break;
    default: hipLaunchKernelGGL(psmf::psmf_blk_filter<32>, dim3(1), dim3(psmf::WG), lds, stream, b); break;
  }
}

void launch_blk_apply(psmf_filter* h, const psmf::BlockParams& b, hipStream_t stream = nullptr) {
  if (!stream) stream = h->stream;
  if (blk_bulk2_ok(h)) {
    const int nyc = (h->block_steps + 15) / 16;
    const size_t lds = psmf::blk_apply2_lds_bytes();
    if (nyc <= 2) hipLaunchKernelGGL(psmf::psmf_blk_apply2<2>, dim3(h->bulk_wgs), dim3(psmf::BK_NT), lds, stream, b);
    else hipLaunchKernelGGL(psmf::psmf_blk_apply2<3>, dim3(h->bulk_wgs), dim3(psmf::BK_NT), lds, stream, b);
    return;
  }
  const int nslab = (h->cfg.d_local + 15) / 16;
  int g = (nslab + 3) / 4;
  if (g > 1024) g = 1024;
  const size_t lds = psmf::blk_apply_lds_bytes();
  if (h->cfg.storage == PSMF_F64)
    hipLaunchKernelGGL(psmf::psmf_blk_apply_mfma<double>, dim3(g), dim3(psmf::WG), lds, stream, b);
  else
    hipLaunchKernelGGL(psmf::psmf_blk_apply_mfma<float>, dim3(g), dim3(psmf::WG), lds, stream, b);
}

int enqueue_block(psmf_filter* h, int64_t k0, int nb) {
  psmf::BlockParams b;
  fill_block_params(h, b, k0, nb);
  launch_blk_gram(h, b);
  if (h->use_coll) { const int rc = all_reduce_sum(h, h->Kmat, psmf::RB * psmf::RB, h->stream); if (rc) return rc; }
  launch_blk_filter(h, b);
  launch_blk_apply(h, b);
  return PSMF_OK;
}

// Pipelined blocks: the filter kernels chain back to back on the main stream; Gram of the first block,
// cross-Grams (one block ahead) and applies run on the bulk stream, synchronised with events:
//   bulk:  gram(0) | xgram(1) | [filter(0)] apply(0) | xgram(2) | [filter(1)] apply(1) | ...
//   main:  [gram(0)] filter(0) | [xgram(1)] filter(1) | [xgram(2)] filter(2) | ...
// xgram(b+1) reads C before apply(b) rewrites it (stream order on bulk); the ping-pong coefficient
// buffers of block b are rewritten by filter(b+2), which waits for xgram(b+2), enqueued after apply(b).
double host_now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
const bool g_host_timing = getenv("PSMF_HOST_TIMING") && atoi(getenv("PSMF_HOST_TIMING")) != 0;

int enqueue_blocks_pipelined(psmf_filter* h, int64_t k_begin, int64_t k_end) {
  const int B = h->block_steps;
  const double t_enq0 = g_host_timing ? host_now_ms() : 0.0;
  double t_prev = t_enq0, t_worst = 0.0;
  long long worst_blk = -1;
  const int64_t nblk = (k_end - k_begin + B - 1) / B;
  auto k0_of = [&](int64_t b) { return k_begin + b * B; };
  auto nb_of = [&](int64_t b) { const int64_t left = k_end - k0_of(b); return (int)(left < B ? left : B); };
  const size_t xg_elems = (size_t)(psmf::RB + psmf::XGB) * psmf::XGB;
  HIP_TRY(h, hipEventRecord(h->evS, h->stream));            // everything enqueued so far (state uploads) is visible to bulk
  HIP_TRY(h, hipStreamWaitEvent(h->bulk, h->evS, 0));
  hipStream_t fs = h->fstream ? h->fstream : h->stream;    // the filter chain (its own CUs when the mask streams exist)
  if (h->fstream) HIP_TRY(h, hipStreamWaitEvent(fs, h->evS, 0));
  psmf::BlockParams b;
  // hand-off by device flags when the filter chain has a stream (hardware queue) of its own; by events otherwise
  const bool flags_off = !h->sw.block_flags;
  // (a tool that serialises dispatches: events.  A host-mediated communicator synchronises the bulk stream at every exchange
  //  anyway, and several such handles usually share one process and one GPU -- shards of a test -- where kernels that spin on
  //  flags could end up behind each other in a shared hardware queue: events there, too -- unless PSMF_HOST_COMM_FLAGS=1 asks for the
  //  flags, which is how tests/test_hip_multishard.py runs the flag hand-off and the chained launch with more than one shard)
  const bool use_flags = h->fstream != nullptr && h->flags != nullptr && !flags_off && h->streams_concurrent && (!h->host_fn || h->sw.host_comm_flags);
  const long long s0 = h->seq_next;
  h->seq_next += nblk;
  // chain: the filter kernels of the whole run as ONE launch (psmf_blk_filter3; the bulk stream is driven as before)
  const bool chain_off = !h->sw.block_chain;
  const bool chain = use_flags && !chain_off && nblk > 1 && ((blk_dual_ok(h) && blk_use_filter3(h)) || blk_seq_ok(h) || blk_simpl_ok(h));
  // first block: plain Gram of the stored C
  fill_block_params(h, b, k0_of(0), nb_of(0), 0);
  launch_blk_gram(h, b, h->bulk);
  if (h->use_coll) { const int rc = all_reduce_sum(h, h->Kmat, psmf::RB * psmf::RB, h->bulk); if (rc) return rc; }
  if (use_flags) hipLaunchKernelGGL(psmf::psmf_flag_set_k, dim3(1), dim3(1), 0, h->bulk, h->flags + 0, s0);
  else HIP_TRY(h, hipEventRecord(h->evX[0], h->bulk));
  if (chain) {
    psmf::BlockParams c;
    fill_block_params(h, c, k0_of(0), nb_of(0), 0);
    c.flags = h->flags;
    c.seq = s0;
    c.last = 1;
    c.chain = (int)nblk;
    c.chain_B = B;
    c.chain_kend = k_end;
    c.Acoef0 = h->Acoef;
    c.Bcoef0 = h->Bcoef;
    c.XG0 = h->XG;
    const int slot_ev = h->evk_pending < psmf_filter::kTimedRuns ? h->evk_pending : -1;
    if (slot_ev >= 0) HIP_TRY(h, hipEventRecord(h->evK0[slot_ev], fs));
    launch_blk_filter(h, c, fs);
    if (slot_ev >= 0) { HIP_TRY(h, hipEventRecord(h->evK1[slot_ev], fs)); ++h->evk_pending; }
    HIP_TRY(h, hipEventRecord(h->evC, fs));
  }
  for (int64_t bi = 0; bi < nblk; ++bi) {
    const int slot = (int)(bi & 1);
    // bulk: cross-Gram for block bi + 1 (needs C as of the start of block bi)
    if (bi + 1 < nblk) {
      psmf::BlockParams x;
      fill_block_params(h, x, k0_of(bi), nb_of(bi), slot);
      x.k1 = k0_of(bi + 1);
      x.nb1 = nb_of(bi + 1);
      double* xg = h->XG + (size_t)((bi + 1) & 1) * xg_elems;
      launch_blk_xgram(h, x, xg, h->bulk);
      if (h->use_coll) { const int rc = all_reduce_sum(h, xg, xg_elems, h->bulk); if (rc) return rc; }
      if (use_flags) hipLaunchKernelGGL(psmf::psmf_flag_set_k, dim3(1), dim3(1), 0, h->bulk, h->flags + 0, s0 + bi + 1);
      else HIP_TRY(h, hipEventRecord(h->evX[(bi + 1) & 3], h->bulk));
    }
    // filter stream: filter of block bi
    fill_block_params(h, b, k0_of(bi), nb_of(bi), slot);
    if (bi > 0) {
      b.assemble = 1;
      b.XG = h->XG + (size_t)(bi & 1) * xg_elems;
      b.Aprev = h->Acoef + (size_t)(slot ^ 1) * psmf::RB * psmf::RM;
    }
    b.last = (bi + 1 == nblk) ? 1 : 0;
    if (use_flags) {
      b.flags = h->flags;
      b.seq = s0 + bi;
      if (!chain) {
        launch_blk_filter(h, b, fs);
        if (bi + 1 == nblk) hipLaunchKernelGGL(psmf::psmf_flag_set_k, dim3(1), dim3(1), 0, fs, h->flags + 1, s0 + nblk);   // the last block has no successor to announce it
      }
      hipLaunchKernelGGL(psmf::psmf_flag_wait_k, dim3(1), dim3(64), 0, h->bulk, h->flags, s0 + bi + 1, h->st);
    } else {
      HIP_TRY(h, hipStreamWaitEvent(fs, h->evX[bi & 3], 0));
      launch_blk_filter(h, b, fs);
      HIP_TRY(h, hipEventRecord(h->evF[bi & 3], fs));
      HIP_TRY(h, hipStreamWaitEvent(h->bulk, h->evF[bi & 3], 0));
    }
    // bulk: apply of block bi
    launch_blk_apply(h, b, h->bulk);
    HIP_TRY(h, hipEventRecord(h->evA[bi & 3], h->bulk));
    if (g_host_timing) { const double t = host_now_ms(); if (t - t_prev > t_worst) { t_worst = t - t_prev; worst_blk = bi; } t_prev = t; }
  }
  if (g_host_timing) {
    const double t = host_now_ms();
    if (t - t_enq0 > 20.0 || t_worst > 5.0)
      fprintf(stderr, "[psmf host timing] enqueue of %lld blocks took %.1f ms, slowest block's calls %.1f ms (block %lld)\n", (long long)nblk, t - t_enq0, t_worst, worst_blk);
  }
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->evA[(nblk - 1) & 3], 0));   // the main stream sees the final C / y_hat
  // ... and the r x r state: the chained kernel writes DevState in its tail, after it has released the last apply, so the end of
  // that launch (not the apply alone) is what host reads / the next run's preparation on the main stream have to follow
  if (chain) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->evC, 0));
  HIP_TRY(h, hipGetLastError());
  return PSMF_OK;
}

int enqueue_gram_into(psmf_filter* h, double* Gout, const DevState* wst, const double* rho_rows) {
  const int r = h->cfg.r;
  const int rows = (h->cfg.d_local + kGramWG - 1) / kGramWG;
  if (h->cfg.storage == PSMF_F64)
    hipLaunchKernelGGL(psmf::psmf_gram_partial<double>, dim3(kGramWG), dim3(psmf::WG), 0, h->stream,
                       (const double*)h->C, h->cfg.d_local, r, h->geo.rp, rows, h->gpart, wst, rho_rows);
  else
    hipLaunchKernelGGL(psmf::psmf_gram_partial<float>, dim3(kGramWG), dim3(psmf::WG), 0, h->stream,
                       (const float*)h->C, h->cfg.d_local, r, h->geo.rp, rows, h->gpart, wst, rho_rows);
  hipLaunchKernelGGL(psmf::psmf_gram_reduce, dim3((r * r + 255) / 256), dim3(256), 0, h->stream,
                     (const double*)h->gpart, kGramWG, r * r, Gout);
  if (h->use_coll) { const int rc = all_reduce_sum(h, Gout, (size_t)r * r, h->stream); if (rc) return rc; }
  return PSMF_OK;
}
constexpr int kMGramWG = 256;      // workgroups of the masked Gram (one partial each)

// psmf_serial_mgram: block 0 = the serial stage, blocks 1 .. kMGramWG = the masked Gram of the next step (psmf_masked.hip)
template <typename T>
int launch_serial_mgram_t(psmf_filter* h, int first) {
  const int r = h->cfg.r, rpad = h->geo.rpad;
  const dim3 grid(1 + kMGramWG);
#define PSMF_SM_LAUNCH(RPAD_, NT_, NW_)                                                                                     \
  do {                                                                                                                      \
    const size_t lds_ = (size_t)psmf::mgram_lds_doubles(NT_, NW_) * sizeof(double);                                         \
    static bool attr_[2] = {false, false};                                                                                  \
    if (!attr_[sizeof(T) == 8]) {                                                                                           \
      HIP_TRY(h, hipFuncSetAttribute((const void*)psmf::psmf_serial_mgram<RPAD_, T, NT_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
      attr_[sizeof(T) == 8] = true;                                                                                         \
    }                                                                                                                       \
    hipLaunchKernelGGL((psmf::psmf_serial_mgram<RPAD_, T, NT_, NW_>), grid, dim3(NW_ * 64), lds_, h->stream, h->sp, first, h->gpart); \
  } while (0)
  if (rpad == 8) PSMF_SM_LAUNCH(8, 1, 16);
  else if (rpad == 16) PSMF_SM_LAUNCH(16, 1, 16);
  else if (rpad == 32) PSMF_SM_LAUNCH(32, 2, 8);
  else if (r <= 48) PSMF_SM_LAUNCH(64, 3, 4);
  else PSMF_SM_LAUNCH(64, 4, 4);
#undef PSMF_SM_LAUNCH
  return PSMF_OK;
}

// serial stage of the step + masked Gram of the next, then the Gram's fixed-order reduction (and its all-reduce over the shards)
int enqueue_serial_mgram(psmf_filter* h, int first) {
  int rc = h->cfg.storage == PSMF_F64 ? launch_serial_mgram_t<double>(h, first) : launch_serial_mgram_t<float>(h, first);
  if (rc) return rc;
  const int ne = h->cfg.r * h->cfg.r + 1;
  // the shares of <G_m, Pbar> for the next sweep's eta: by the reduction itself, or -- row shards -- behind the all-reduce of the Gram
  const int ntr = (ne + 63) / 64;
  double* tpart = h->mg + ne + 1;
  hipLaunchKernelGGL(psmf::psmf_mgram_reduce, dim3(ntr), dim3(512), 0, h->stream, (const double*)h->gpart, (int)kMGramWG, ne, h->mg,
                     h->use_coll ? (const double*)nullptr : (const double*)h->st->Pbar, h->cfg.r, tpart);
  if (h->use_coll) {
    const int rc2 = all_reduce_sum(h, h->mg, (size_t)ne, h->stream);
    if (rc2) return rc2;
    hipLaunchKernelGGL(psmf::psmf_mgram_trace, dim3(ntr), dim3(64), 0, h->stream, (const double*)h->mg, (const double*)h->st->Pbar, h->cfg.r, tpart);
  }
  return PSMF_OK;
}

int enqueue_gram(psmf_filter* h) { return enqueue_gram_into(h, h->st->G, nullptr, nullptr); }

// the weighted Gram of the current step (non-uniform diagonal R) on the matrix cores: psmf_wgram_mfma + the masked Gram's reduction
template <typename T>
int launch_wgram_t(psmf_filter* h) {
  const int r = h->cfg.r, rpad = h->geo.rpad;
#define PSMF_WG_LAUNCH(NT_)                                                                                                 \
  do {                                                                                                                      \
    const size_t lds_ = (size_t)psmf::mgram_lds_doubles(NT_, 8) * sizeof(double);                                           \
    static bool attr_[2] = {false, false};                                                                                  \
    if (!attr_[sizeof(T) == 8]) {                                                                                           \
      HIP_TRY(h, hipFuncSetAttribute((const void*)psmf::psmf_wgram_mfma<T, NT_, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
      attr_[sizeof(T) == 8] = true;                                                                                         \
    }                                                                                                                       \
    hipLaunchKernelGGL((psmf::psmf_wgram_mfma<T, NT_, 8>), dim3(kMGramWG), dim3(8 * 64), lds_, h->stream, h->sp, h->gpart);  \
  } while (0)
  if (rpad <= 16) PSMF_WG_LAUNCH(1);
  else if (rpad == 32) PSMF_WG_LAUNCH(2);
  else if (r <= 48) PSMF_WG_LAUNCH(3);
  else PSMF_WG_LAUNCH(4);
#undef PSMF_WG_LAUNCH
  return PSMF_OK;
}

int enqueue_weighted_gram(psmf_filter* h) {
  if (!h->sw.wgram_mfma) return enqueue_gram_into(h, h->st->GR, h->st, h->sp.rho_rows);      // PSMF_WGRAM_MFMA=0: the vector-unit Gram
  const int r = h->cfg.r;
  const int rc = h->cfg.storage == PSMF_F64 ? launch_wgram_t<double>(h) : launch_wgram_t<float>(h);
  if (rc) return rc;
  hipLaunchKernelGGL(psmf::psmf_mgram_reduce, dim3((r * r + 63) / 64), dim3(512), 0, h->stream, (const double*)h->gpart, (int)kMGramWG, r * r,
                     h->st->GR, (const double*)nullptr, r, (double*)nullptr);
  if (h->use_coll) { const int rc2 = all_reduce_sum(h, h->st->GR, (size_t)r * r, h->stream); if (rc2) return rc2; }
  return PSMF_OK;
}

void destroy_graph(psmf_filter* h) {
  if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
  if (h->graph) { hipGraphDestroy(h->graph); h->graph = nullptr; }
  h->chunk = 0;
}

int build_graph(psmf_filter* h, int chunk) {
  destroy_graph(h);
  HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = PSMF_OK;
  for (int i = 0; i < chunk && rc == PSMF_OK; ++i) rc = enqueue_step(h);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(h->stream, &g);
  if (rc != PSMF_OK) { if (g) hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return fail(h, PSMF_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  h->graph = g;
  HIP_TRY(h, hipGraphInstantiate(&h->gexec, h->graph, nullptr, nullptr, 0));
  h->chunk = chunk;
  return PSMF_OK;
}

int ensure_scratch(psmf_filter* h, size_t bytes) {
  if (h->scratch_bytes >= bytes) return PSMF_OK;
  if (h->scratch) HIP_TRY(h, hipFree(h->scratch));
  h->scratch = nullptr;
  h->scratch_bytes = 0;
  HIP_TRY(h, hipMalloc((void**)&h->scratch, bytes));
  h->scratch_bytes = bytes;
  return PSMF_OK;
}

int ensure_rot_tmp(psmf_filter* h, size_t bytes) {
  if (h->rot_tmp_bytes >= bytes) return PSMF_OK;
  if (h->rot_tmp) HIP_TRY(h, hipFree(h->rot_tmp));
  h->rot_tmp = nullptr;
  h->rot_tmp_bytes = 0;
  HIP_TRY(h, hipMalloc(&h->rot_tmp, bytes));
  h->rot_tmp_bytes = bytes;
  return PSMF_OK;
}

// O[M x N] = A[M x K] B[K x N] on the handle's stream (psmf_rotate.hip); the caller synchronises
template <typename TA, typename TB, typename TO>
int rot_gemm(psmf_filter* h, const TA* A, long long a_i, long long a_k, const TB* B, long long b_k, long long b_j, TO* O, long long o_i,
             long long M, long long N, long long K) {
  if (M <= 0 || N <= 0 || K <= 0) return PSMF_OK;
  if (!A || !B || !O || (M + psmf::ROT_T - 1) / psmf::ROT_T > 65535) return fail(h, PSMF_ERR_ARG, "rotation: bad operand");
  dim3 grid((unsigned)((N + psmf::ROT_T - 1) / psmf::ROT_T), (unsigned)((M + psmf::ROT_T - 1) / psmf::ROT_T));
  hipLaunchKernelGGL((psmf::psmf_rot_gemm<TA, TB, TO>), grid, dim3(256), 0, h->stream, A, a_i, a_k, B, b_k, b_j, O, o_i, (int)M, (int)N, (int)K);
  HIP_TRY(h, hipGetLastError());
  return PSMF_OK;
}

// rows of X (n x d, storage type, row stride d) times U (fwd: into rotated coordinates) or U^T (back), out of place: src -> dst
int rot_rows(psmf_filter* h, const void* src, void* dst, long long n, bool fwd) {
  const long long d = h->cfg.d_local;
  const long long bk = fwd ? d : 1, bj = fwd ? 1 : d;
  if (h->cfg.storage == PSMF_F64) return rot_gemm(h, (const double*)src, d, 1LL, (const double*)h->rotU, bk, bj, (double*)dst, d, n, d, d);
  return rot_gemm(h, (const float*)src, d, 1LL, (const double*)h->rotU, bk, bj, (float*)dst, d, n, d, d);
}

// the dictionary (d x rp, storage type): dst = U^T src (fwd) or U src (back)
int rot_dict(psmf_filter* h, const void* src, void* dst, bool fwd) {
  const long long d = h->cfg.d_local, rp = h->geo.rp, r = h->cfg.r;
  const long long ai = fwd ? 1 : d, ak = fwd ? d : 1;
  if (h->cfg.storage == PSMF_F64) return rot_gemm(h, (const double*)h->rotU, ai, ak, (const double*)src, rp, 1LL, (double*)dst, rp, d, r, d);
  return rot_gemm(h, (const double*)h->rotU, ai, ak, (const float*)src, rp, 1LL, (float*)dst, rp, d, r, d);
}

void compute_geometry(const psmf_config& c, Geometry& g, const int sweep_nt = 512) {
  g.vec = c.storage == PSMF_F64 ? 2 : 4;
  g.nv = (c.r + g.vec - 1) / g.vec;
  g.rp = g.nv * g.vec;
  g.gs = next_pow2(g.nv);
  g.nt = c.r > 32 ? 256 : sweep_nt;      // r > 32: the solve block (3 x 3 / 4 x 4 tiles of 16 x 16 in ONE wave's registers) needs a 256-thread kernel's register budget
  g.rpp = g.nt / g.gs;
  g.rpad = next_pow2(c.r < 8 ? 8 : c.r);
  const size_t solve_lds = c.coef_update ? (size_t)(4 * psmf::RM + 2) * 8 : 0;
  const size_t red_lds = (size_t)(g.nt / 64) * (g.gs * g.vec + 1) * 8;
  g.sweep_lds = ((solve_lds > red_lds ? solve_lds : red_lds) + 15) & ~(size_t)15;
  // (r > 32: one 256-thread workgroup per CU -- one wave per SIMD, see sweep_unroll -- so ONE round of workgroups, a few CUs left to the solve block)
  int target = c.n_workgroups > 0 ? c.n_workgroups : (c.r > 32 ? 248 : (g.nt == 512 ? 256 : 512));
  int rows = (c.d_local + target - 1) / target;
  rows = ((rows + g.rpp - 1) / g.rpp) * g.rpp;
  if (rows < g.rpp) rows = g.rpp;
  g.rows_per_wg = rows;
  g.n_sweep_wg = (c.d_local + rows - 1) / rows;
  g.ps = c.nonuniform_R ? 2 * (c.r + 1) : c.r + 1;      // partial row: h, ee (+ the weighted b, q)
}

// StepParams.solve_dual: can the per-step solve block run its two inversions side by side?  (random walk, Q = q I as last uploaded,
// full filter, uniform R, wave-local solve, no Q_k schedule: q of the next step is q -- or omega q -- of this one)
void update_solve_dual(psmf_filter* h) {
  const psmf_config& c = h->cfg;
  // Lbar' = (I / q - W / q^2) / omega loses log10((p + q) / q) digits to cancellation (p: the scale of P): with float64 storage, whose
  // results are otherwise good to 1e-11, the side-by-side form is left where that is more than five (q = 1e-8 against P0 = I: 3e-8
  // measured, 3e-11 with the inversions one after the other -- tools/probe_step_tinyq.py); float32 storage has its 1e-7 anyway
  const bool cancels = c.storage == PSMF_F64 && h->q_last > 0.0 && h->p_diag_max > 1e5 * h->q_last;
  const int v = (h->sw.step_dual && h->engine == 1 && h->q_iso && c.masked < 2 && c.dyn_kind == PSMF_DYN_RANDOM_WALK && c.coef_update && c.pbar_predict && !c.nonuniform_R &&
                 !h->sp.solve_lds && !h->sp.q_sched && !h->sp.q_mat && !cancels) ? 1 : 0;
  if (v != h->sp.solve_dual) {
    h->sp.solve_dual = v;
    destroy_graph(h);        // the captured launches carry the old parameter block
  }
}

// Persistent kernels spin on device flags: two of them resident at once (two handles of one process on one device) could each
// hold compute units the other needs.  They are serialised per device with an event chain.
std::mutex g_ps_mutex;
hipEvent_t g_ps_event[64] = {};

int launch_pstep(psmf_filter* h, int64_t k_begin, int64_t n) {
  while (n > 0) {
    const int64_t chunk = n > (int64_t)1 << 30 ? (int64_t)1 << 30 : n;
    psmf::PstepParams q;
    memset(&q, 0, sizeof(q));
    q.sp = h->sp;
    q.k_begin = k_begin;
    q.n_steps = (int)chunk;
    q.n_row_wg = h->ps_plan.n_row_wg;
    q.rows_per_wg = h->ps_plan.rows_per_wg;
    q.np = h->ps_plan.np;
    q.ncol2 = h->ps_plan.ncol2;
    q.flags = reinterpret_cast<unsigned*>(h->ps_comm);
    q.pkt = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(h->ps_comm) + h->ps_plan.off_pkt);
    q.part = reinterpret_cast<double*>(reinterpret_cast<char*>(h->ps_comm) + h->ps_plan.off_part);
    if (h->cfg.masked) {
      char* base = reinterpret_cast<char*>(h->ps_comm);
      q.masked = 1;
      q.nge = h->ps_plan.nge;
      q.slice_len = h->ps_plan.slice_len;
      q.gflags = reinterpret_cast<unsigned*>(base + h->ps_plan.off_gflags);
      q.sflags = reinterpret_cast<unsigned*>(base + h->ps_plan.off_sflags);
      q.gpart = reinterpret_cast<double*>(base + h->ps_plan.off_gpart);
      q.gslice = reinterpret_cast<double*>(base + h->ps_plan.off_gslice);
      q.mg_out = h->mg;
      q.mg_ntr = h->sp.mg_ntr;
    }
    q.prof = h->ps_prof;
    h->ps_prof_steps = chunk;
    std::lock_guard<std::mutex> lock(g_ps_mutex);
    const int dev = h->cfg.device & 63;
    if (!g_ps_event[dev]) HIP_TRY(h, hipEventCreateWithFlags(&g_ps_event[dev], hipEventDisableTiming));
    else HIP_TRY(h, hipStreamWaitEvent(h->stream, g_ps_event[dev], 0));
    HIP_TRY(h, hipMemsetAsync(h->ps_comm, 0, h->ps_plan.zero_bytes, h->stream));      // every polled word, before every launch
    HIP_TRY(h, psmf::pstep_launch(q, h->cfg.storage == PSMF_F64, h->stream));
    HIP_TRY(h, hipEventRecord(g_ps_event[dev], h->stream));
    ++h->ps_launches;
    k_begin += chunk;
    n -= chunk;
  }
  return PSMF_OK;
}

// Give-up policy of filter3's Newton-Schulz starts (docs/MEASUREMENTS.md round 5).  A start whose residual exceeds
// `far` is abandoned for the pivot-exact LDS sweep (15 us) and the next `skip` steps sweep unasked.  Where Lbar' = (I / q - W / q^2) / omega
// cancels -- eigenvalues of Pbar far above q: the adversarial Q = 1e-8 -- every early step that iterates instead of sweeping costs
// accuracy, so the conservative 0.3 / 3 stays; everywhere else (max diag P / q <= 1e3: at most three digits cancel) 0.9 / 1 takes
// config E's cold pass from 36.5 to 35.5 ms with unchanged errors (full-size and adversarial suites under it: profiles/r5_ns_far_policy.txt).
// PSMF_NS_FAR / PSMF_NS_SKIP override both.
void update_ns_policy(psmf_filter* h) {
  const bool benign = h->q_iso && h->q_last > 0.0 && h->p_diag_max <= 1e3 * h->q_last;
  if (!h->ns_far_env) { const double f = benign ? 0.9 : 0.3; h->sp.ns_far2 = f * f; }
  if (!h->ns_skip_env) h->sp.ns_skip_n = benign ? 1 : 3;
}

int set_device(psmf_handle h) {
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  return PSMF_OK;
}

template <typename T>
void pack_rows(const double* src, T* dst, int d_local, int r, int rp) {
  for (int i = 0; i < d_local; ++i) {
    for (int c = 0; c < r; ++c) dst[(size_t)i * rp + c] = (T)src[(size_t)i * r + c];
    for (int c = r; c < rp; ++c) dst[(size_t)i * rp + c] = (T)0;
  }
}

// start-of-run preparation: step counter, exact Gram, then everything the first sweep needs
int prepare(psmf_filter* h, int64_t k_begin) {
  // step counter and error flag by a one-thread kernel (its arguments travel with the launch): no host buffer to keep
  // alive, so no synchronisation here -- consecutive passes over the series queue up back to back
  hipLaunchKernelGGL(psmf::psmf_prepare_k, dim3(1), dim3(1), 0, h->stream, h->st, (long long)k_begin);
  if (h->flags) hipLaunchKernelGGL(psmf::psmf_flag_set_k, dim3(1), dim3(1), 0, h->stream, h->flags + 2, 0LL);
  if (h->mu_hist)
    HIP_TRY(h, hipMemcpyAsync(h->mu_hist + (size_t)(k_begin - h->sp.series_t0) * h->cfg.r, h->st->mu, h->cfg.r * sizeof(double),
                              hipMemcpyDeviceToDevice, h->stream));
  if (h->engine == 2) {   // the blocked engine derives everything it needs from the block Gram
    h->need_prep = false;
    h->k_done = k_begin;
    return PSMF_OK;
  }
  if (h->sp.track_g) {
    int rc = enqueue_gram(h);
    if (rc != PSMF_OK) return rc;
  }
  if (h->cfg.masked) {      // + the masked Gram of the run's first step (psmf_prepare_k set kq = k_begin)
    const int rc = enqueue_serial_mgram(h, 1);
    if (rc) return rc;
  } else {
    launch_serial(h, 1);
  }
  HIP_TRY(h, hipGetLastError());
  h->need_prep = false;
  h->k_done = k_begin;
  return PSMF_OK;
}

// f(theta, x, t) on the host, for the predict roll-out (psmf.py:182-188); same term structure as psmf_dyn.hip
void dyn_f_host(const psmf_config& c, const double* th, const double* x, double t, double* out) {
  const int r = c.r, kind = c.dyn_kind, flags = c.dyn_flags, N = c.dyn_terms;
  if (kind == PSMF_DYN_RANDOM_WALK) { for (int i = 0; i < r; ++i) out[i] = x[i]; return; }
  if (kind == PSMF_DYN_SCALED_WALK) {
    for (int i = 0; i < r; ++i) {
      double a = (flags & 1) ? th[r * r + i] : 0.0;
      for (int j = 0; j < r; ++j) a += th[i * r + j] * x[j];
      out[i] = a;
    }
    return;
  }
  for (int i = 0; i < r; ++i) out[i] = 0.0;
  std::vector<double> val(r);
  const int nt = psmf::dyn_n_terms(kind, N);
  for (int tI = 0; tI < nt; ++tI) {
    const psmf::DynTerm d = psmf::dyn_term(kind, flags, N, r, tI);
    for (int j = 0; j < r; ++j) {
      const double arg = 2.0 * M_PI * th[d.b_off + j] * t + (d.c_off >= 0 ? th[d.c_off + j] : 1.0) * x[j];
      val[j] = d.is_cos ? cos(arg) : sin(arg);
    }
    for (int i = 0; i < r; ++i) {
      if (d.m_off >= 0) { double a = 0.0; for (int j = 0; j < r; ++j) a += th[d.m_off + i * r + j] * val[j]; out[i] += a; }
      else out[i] += val[i];
    }
  }
}

}  // namespace

extern "C" {

int psmf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* psmf_last_error(psmf_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int psmf_create(psmf_handle* out, const psmf_config* cfg) {
  if (!out || !cfg) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: null argument");
  *out = nullptr;
  if (cfg->abi_version != PSMF_ABI_VERSION) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: ABI version mismatch");
  if (cfg->r < 1 || cfg->r > PSMF_RMAX) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: need 1 <= r <= 64");
  if (cfg->d < 1 || cfg->d_local < 1 || cfg->row0 < 0 || cfg->row0 + cfg->d_local > cfg->d)
    return fail(nullptr, PSMF_ERR_ARG, "psmf_create: bad d / row0 / d_local");
  if (cfg->storage != PSMF_F32 && cfg->storage != PSMF_F64) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: storage must be f32 or f64");
  if (cfg->dyn_kind < PSMF_DYN_RANDOM_WALK || cfg->dyn_kind > PSMF_DYN_HOST) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: unknown dyn_kind");
  if (cfg->dyn_kind == PSMF_DYN_FOURIER && (cfg->dyn_terms < 1 || 2 * cfg->dyn_terms > psmf::DYN_MAX_TERMS))
    return fail(nullptr, PSMF_ERR_ARG, "psmf_create: Fourier dynamics need 1 <= dyn_terms <= 4");
  if (cfg->n_theta != psmf::dyn_n_theta(cfg->dyn_kind, cfg->dyn_flags, cfg->dyn_terms, cfg->r))
    return fail(nullptr, PSMF_ERR_ARG, "psmf_create: n_theta does not match dyn_kind / dyn_flags / dyn_terms (see psmf_dyn_kind)");
  if (cfg->dyn_kind == PSMF_DYN_HOST && cfg->recursive) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: host-stepped dynamics keep theta (and its optimiser) on the host");
  if (cfg->recursive && cfg->update_every < 1) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: update_every must be >= 1");
  if (cfg->recursive < 0 || cfg->recursive > 2) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: recursive must be 0, 1 (in-loop Adam) or 2 (in-loop SGD)");
  if (cfg->masked < 0 || cfg->masked > 3) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: masked must be 0 .. 3");
  if (cfg->masked >= 2 && (cfg->robust || cfg->dyn_kind != PSMF_DYN_RANDOM_WALK))
    return fail(nullptr, PSMF_ERR_ARG, "psmf_create: masked = 2 (MLE-SMF) / 3 (TMF) are random-walk, non-robust filters");
  if (cfg->masked) {
    if (cfg->dyn_kind != PSMF_DYN_RANDOM_WALK)
      return fail(nullptr, PSMF_ERR_ARG, "psmf_create: masked handles are random-walk filters (ExperimentImpute/PSMF.py:65-66: Pbar = P + Q)");
    if (!cfg->coef_update || !cfg->eta_full || !cfg->pbar_predict || cfg->nonuniform_R || cfg->engine == 2 || !cfg->store_y_pred)
      return fail(nullptr, PSMF_ERR_ARG, "psmf_create: masked = 1 needs the full filter (coef_update, eta_full, pbar_predict), a uniform diagonal R, "
                                         "store_y_pred = 1 and the per-step engine");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(nullptr, PSMF_ERR_NO_DEVICE, "psmf_create: no HIP device visible (the MI355X path has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, PSMF_ERR_ARG, "psmf_create: bad device ordinal");

  psmf_filter* h = new psmf_filter();
  h->cfg = *cfg;
  h->sw.read();
  compute_geometry(h->cfg, h->geo, h->sw.sweep_threads);
  auto bail = [&](int code) { g_create_error = h->err; psmf_destroy(h); return code; };
#define CREATE_TRY(expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(PSMF_ERR_HIP); } \
  } while (0)
  CREATE_TRY(hipSetDevice(cfg->device));
  CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CREATE_TRY(hipEventCreate(&h->ev0));
  CREATE_TRY(hipEventCreate(&h->ev1));
  CREATE_TRY(hipHostMalloc((void**)&h->err_host, sizeof(int), hipHostMallocMapped));
  CREATE_TRY(hipHostGetDevicePointer((void**)&h->err_host_dev, h->err_host, 0));
  CREATE_TRY(hipMalloc((void**)&h->st, sizeof(DevState)));
  CREATE_TRY(hipMemset(h->st, 0, sizeof(DevState)));
  CREATE_TRY(hipMalloc(&h->C, (size_t)cfg->d_local * h->geo.rp * h->elem()));
  CREATE_TRY(hipMalloc((void**)&h->partials, (size_t)h->geo.n_sweep_wg * h->geo.ps * sizeof(double)));
  CREATE_TRY(hipMalloc((void**)&h->gpart, (size_t)((cfg->masked || cfg->nonuniform_R) ? 256 : kGramWG) * (cfg->r * cfg->r + 1) * sizeof(double)));
  if (cfg->masked) CREATE_TRY(hipMalloc((void**)&h->mg, (size_t)(cfg->r * cfg->r + 2 + (cfg->r * cfg->r + 64) / 64 + 1) * sizeof(double)));     // Gram, count | trace shares
  {
    const bool can_block = cfg->r <= psmf::RM / 2 && cfg->dyn_kind != PSMF_DYN_HOST && !cfg->nonuniform_R && !cfg->masked;
    if (cfg->engine == 2 && !can_block) { h->err = "psmf_create: the blocked engine needs r <= 32, device-evaluated dynamics and a uniform diagonal R"; return bail(PSMF_ERR_ARG); }
    if (cfg->engine < 0 || cfg->engine > 2) { h->err = "psmf_create: engine must be 0 (auto), 1 (per-step) or 2 (blocked)"; return bail(PSMF_ERR_ARG); }
    // auto: blocked whenever it applies -- it is exact and removes the per-step launches and row sweeps
    h->engine = cfg->engine == 0 ? (can_block ? 2 : 1) : cfg->engine;
    if (const char* e = getenv("PSMF_ENGINE")) { const int v = atoi(e); if (v == 1 || (v == 2 && can_block)) h->engine = v; }
    // (scaled-walk / sinusoid / Fourier dynamics on the per-step engine -- r > 32, a non-uniform R, engine = 1: the launched form's
    //  serial stage evaluates them through psmf_dyn.hip like the blocked engine's general kernel)
  }
  h->th_cap = (size_t)(cfg->n_theta > psmf::RM ? cfg->n_theta : psmf::RM);
  CREATE_TRY(hipMalloc((void**)&h->thbuf, 4 * h->th_cap * sizeof(double)));
  CREATE_TRY(hipMemset(h->thbuf, 0, 4 * h->th_cap * sizeof(double)));
  if (h->engine == 2) {
    // B = 64 - r timesteps per block, at most 48: the role-specialised filter kernel and the streaming bulk kernels stage
    // up to three 16-column tiles of a series block (r < 16 would otherwise give blocks of 49..63)
    h->block_steps = psmf::RB - cfg->r < 48 ? psmf::RB - cfg->r : 48;
    CREATE_TRY(hipMalloc((void**)&h->Kpart, (size_t)psmf::BLK_GRAM_WG * psmf::RB * psmf::RB * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->Kmat, (size_t)psmf::RB * psmf::RB * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->Acoef, (size_t)2 * psmf::RB * psmf::RM * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->Bcoef, (size_t)2 * psmf::RB * psmf::RB * sizeof(double)));
    CREATE_TRY(hipMemset(h->Bcoef, 0, (size_t)2 * psmf::RB * psmf::RB * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->XGpart, (size_t)psmf::BLK_GRAM_WG * (psmf::RB + psmf::XGB) * psmf::XGB * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->XG, (size_t)2 * (psmf::RB + psmf::XGB) * psmf::XGB * sizeof(double)));
    CREATE_TRY(hipMemset(h->XG, 0, (size_t)2 * (psmf::RB + psmf::XGB) * psmf::XGB * sizeof(double)));   // the all-reduce covers entries no kernel writes
    CREATE_TRY(hipMalloc((void**)&h->flags, 8 * sizeof(long long)));
    CREATE_TRY(hipMemset(h->flags, 0, 8 * sizeof(long long)));
    {
      // The filter chain is one workgroup on the critical path; the bulk kernels (cross-Gram, apply) run
      // beside it and would be co-scheduled onto its CU, stretching it by 10-17 % (measured).  Partition the
      // chip with CU masks: the filter's stream owns `nres` CUs, the bulk stream the others.
      int nres = 8;
      if (const char* e = getenv("PSMF_RESERVED_CUS")) nres = atoi(e);
      hipDeviceProp_t prop;
      CREATE_TRY(hipGetDeviceProperties(&prop, cfg->device));
      const int ncu = prop.multiProcessorCount;
      const int words = (ncu + 31) / 32;
      // The split below is written for the unpartitioned MI355X: 256 CUs = 8 XCDs x 4 shader engines x 8 CUs, mask bit =
      // 32 cu + 8 se + xcc (tools/xcc_probe.hip).  On any other device (a CPX / NPS partition, another part) the bit layout and the
      // engine count are not known here: no CU masks, plain streams -- the filter chain then shares CUs with the bulk kernels
      // (10-17 % slower, measured), which is a speed matter only.
      const bool known_layout = ncu == 256;
      if (known_layout && nres > 0 && nres < ncu / 2 && words <= 16) {
        uint32_t mf[16] = {0}, mb[16] = {0};
        for (int i = 0; i < ncu; ++i) (i < nres ? mf : mb)[i >> 5] |= 1u << (i & 31);
        hipStream_t fs = nullptr, bs = nullptr;
        if (hipExtStreamCreateWithCUMask(&fs, words, mf) == hipSuccess && hipExtStreamCreateWithCUMask(&bs, words, mb) == hipSuccess) {
          h->fstream = fs;
          h->bulk = bs;
          h->reserved_cus = nres;
          // The streaming kernels hold one 512-thread workgroup per CU (86-131 KB of LDS), and the dispatcher deals workgroups
          // to the 32 shader engines (8 XCDs x 4) in equal shares whatever the mask has left each of them.  The filter's
          // 8 CUs are CU 0 of engine 0 of every XCD (mask bit = 32 cu + 8 se + xcc, tools/xcc_probe.hip): those engines
          // keep 7 CUs, so with more than 7 workgroups per engine one CU gets a second one and the kernel takes two
          // rounds -- 231 / 317 us per block at d = 1e6 with 248 or 256 workgroups against 138 / 193 us with 224
          // (tools/bulk_stream.hip; 124 / 172 us on the unmasked chip).  Hence (CUs per engine - 1) x 32.
          {
            const int n_engines = 32, per_engine = ncu / n_engines - (nres + n_engines - 1) / n_engines;
            h->bulk_wgs = per_engine >= 1 ? per_engine * n_engines : 8;
            if (h->bulk_wgs > 256) h->bulk_wgs = 256;
            if (const char* e = getenv("PSMF_BULK_WGS")) { const int v = atoi(e); if (v >= 8 && v <= 256) h->bulk_wgs = (v / 8) * 8; }
          }
        } else {
          (void)hipGetLastError();
          if (fs) hipStreamDestroy(fs);
          if (bs) hipStreamDestroy(bs);
        }
      }
      if (!h->bulk) CREATE_TRY(hipStreamCreateWithFlags(&h->bulk, hipStreamNonBlocking));
    }
    for (int i = 0; i < 4; ++i) {
      CREATE_TRY(hipEventCreateWithFlags(&h->evF[i], hipEventDisableTiming));
      CREATE_TRY(hipEventCreateWithFlags(&h->evA[i], hipEventDisableTiming));
      CREATE_TRY(hipEventCreateWithFlags(&h->evX[i], hipEventDisableTiming));
    }
    CREATE_TRY(hipEventCreateWithFlags(&h->evS, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->evC, hipEventDisableTiming));
    for (int i = 0; i < psmf_filter::kTimedRuns; ++i) { CREATE_TRY(hipEventCreate(&h->evK0[i])); CREATE_TRY(hipEventCreate(&h->evK1[i])); }
    if (h->fstream && h->flags) {
      // the device-flag hand-off and the chained filter launches need the two streams to run concurrently: probe it (a waiter on the filter stream, then the
      // setter on the bulk stream; the waiter gives up after 50 ms)
      int* dres = nullptr;
      struct FreeOnExit { int** p; ~FreeOnExit() { if (*p) { hipFree(*p); *p = nullptr; } } } dres_guard{&dres};      // also on the CREATE_TRY failure paths below
      CREATE_TRY(hipMalloc((void**)&dres, sizeof(int)));
      CREATE_TRY(hipMemset(dres, 0, sizeof(int)));
      CREATE_TRY(hipDeviceSynchronize());      // hipMemset is asynchronous on the null stream, the probe's streams are non-blocking: the zeroes (of dres and of h->flags above) first
      hipLaunchKernelGGL(psmf::psmf_probe_wait_k, dim3(1), dim3(1), 0, h->fstream, h->flags + 7, 1LL, 5000000LL, dres);
      hipLaunchKernelGGL(psmf::psmf_flag_set_k, dim3(1), dim3(1), 0, h->bulk, h->flags + 7, 1LL);
      CREATE_TRY(hipStreamSynchronize(h->fstream));
      CREATE_TRY(hipStreamSynchronize(h->bulk));
      int res = 0;
      CREATE_TRY(hipMemcpy(&res, dres, sizeof(int), hipMemcpyDeviceToHost));
      h->streams_concurrent = res == 1;
    }
    const size_t flds = psmf::blk_filter_lds_bytes();
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter6d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter7, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    const size_t alds = psmf::blk_apply_lds_bytes();
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_apply_mfma<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)alds));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_apply_mfma<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)alds));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_filter3_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter3s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_filter3_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_filter3_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter4s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_filter3_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_filter3_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_xgram2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_xgram2_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_xgram2<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_xgram2_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_apply2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_apply2_lds_bytes()));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_apply2<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmf::blk_apply2_lds_bytes()));
    const size_t flds2 = psmf::blk_filter2_lds_bytes();
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter2<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds2));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter2<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds2));
    CREATE_TRY(hipFuncSetAttribute((const void*)psmf::psmf_blk_filter2<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds2));
  }
  if (h->geo.sweep_lds > 48 * 1024)
    CREATE_TRY(hipFuncSetAttribute((const void*)sweep_kernel(h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->geo.sweep_lds));
  if (h->engine == 1 && h->sw.step_persistent && (cfg->r <= 32 || (h->sw.pstep_big && cfg->r <= 48 && cfg->masked == 0)) && cfg->masked <= 1 && !cfg->nonuniform_R &&
      cfg->dyn_kind <= PSMF_DYN_COS_PHASE) {
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (psmf::pstep_plan(cfg->d_local, cfg->r, prop.multiProcessorCount, cfg->storage == PSMF_F64, cfg->masked == 1, &h->ps_plan)) {
      CREATE_TRY(psmf::pstep_init());
      CREATE_TRY(hipMalloc(&h->ps_comm, h->ps_plan.total_bytes));
      CREATE_TRY(hipMemset(h->ps_comm, 0, h->ps_plan.total_bytes));
      if (getenv("PSMF_PSTEP_PROF")) { CREATE_TRY(hipMalloc((void**)&h->ps_prof, 64 * sizeof(long long))); CREATE_TRY(hipMemset(h->ps_prof, 0, 64 * sizeof(long long))); }
      h->ps_ok = true;
    }
  }
#undef CREATE_TRY

  StepParams& sp = h->sp;
  memset(&sp, 0, sizeof(sp));
  sp.st = h->st;
  sp.C = h->C;
  sp.partials = h->partials;
  sp.d = cfg->d; sp.d_local = cfg->d_local; sp.r = cfg->r; sp.rp = h->geo.rp; sp.nv = h->geo.nv;
  sp.n_sweep_wg = h->geo.n_sweep_wg; sp.rows_per_wg = h->geo.rows_per_wg; sp.ps = h->geo.ps;
  sp.robust = cfg->robust; sp.coef_update = cfg->coef_update; sp.eta_full = cfg->eta_full;
  sp.pbar_predict = cfg->pbar_predict; sp.fixed_lambda = cfg->fixed_lambda;
  sp.dyn_kind = cfg->dyn_kind; sp.n_theta = cfg->n_theta; sp.store_yp = 0;
  sp.dyn_flags = cfg->dyn_flags; sp.dyn_terms = cfg->dyn_terms;
  sp.theta = h->thbuf; sp.gradsum = h->thbuf + h->th_cap; sp.adam_m = h->thbuf + 2 * h->th_cap; sp.adam_v = h->thbuf + 3 * h->th_cap;
  sp.rho_sched = nullptr; sp.q_sched = nullptr; sp.q_mat = nullptr;
  sp.rho_rows = nullptr; sp.rho_mean = 1.0;
  sp.recursive = cfg->recursive; sp.update_every = cfg->update_every > 0 ? cfg->update_every : 1;
  sp.track_g = ((cfg->eta_full || cfg->coef_update) && !cfg->masked) ? 1 : 0;     // masked: G is this step's masked Gram, recomputed every step
  sp.mask = nullptr; sp.mg = nullptr; sp.mg_tr = nullptr; sp.mg_ntr = 0; sp.sc_hist = nullptr; sp.mask_rows = 0;
  sp.masked_method = cfg->masked >= 2 ? cfg->masked : 0;
  sp.solve_lds = (Switches::off("PSMF_STEP_WAVE_SOLVE") || (cfg->r > 32 && Switches::off("PSMF_STEP_WAVE_BIG"))) ? 1 : 0;
  {
    // The last row workgroup of a sweep sums the partial rows (tail_reduce_partials) where the solve block outlasts the row blocks
    // by more than that tail -- r > 32 at moderate d_local, small shards: 31.6 -> 30.0 us per timestep at r = 40, d = 2e4, 50.3 ->
    // 47.4 at r = 64 -- and the serial stage does where the rows are the longer part (d = 1e5, r = 32: 19.5 against 21.2 with the
    // tail; tools/probe_tail.py).  PSMF_TAIL_REDUCE=1 / 0 forces it on / off.
    const double rows_us = 2.0 * (double)cfg->d_local * h->geo.rp * (double)h->elem() / 3.0e6;
    const double solve_us = cfg->r <= 32 ? 0.3 * cfg->r : (cfg->r <= 48 ? 13.0 : 24.0);
    const char* e = getenv("PSMF_TAIL_REDUCE");
    const bool forced = e && atoi(e) == 1;
    sp.tail_reduce = (h->engine == 1 && h->sw.tail_reduce && (forced || (cfg->coef_update && rows_us + 3.0 < solve_us))) ? 1 : 0;
  }
  sp.external_reduce = sp.tail_reduce;
  sp.use_ns = (getenv("PSMF_NS") && atoi(getenv("PSMF_NS")) == 0) ? 0 : 1;
  sp.ns_predict = getenv("PSMF_NS_PREDICT") ? atoi(getenv("PSMF_NS_PREDICT")) : 7;      // bits: 1 a / b (phase F), 2 core (wave 7), 4 applied
  // Newton-Schulz acceptance: ||I - M X||_F below the tolerance BEFORE the last update (which squares it).  float64
  // storage: 3e-7 (-> 1e-13).  float32 storage: 3e-4 (-> ~1e-7, of the order of the rounding of C and y to float32; errors
  // against the float64 oracle measured at 1e-4 / 3e-4 / 1e-3 in DESIGN section 5: unchanged up to 3e-4).  PSMF_NS_TOL overrides.
  const double ns_tol = getenv("PSMF_NS_TOL") ? atof(getenv("PSMF_NS_TOL")) : (cfg->storage == PSMF_F64 ? 3e-7 : 3e-4);
  sp.ns_tol2 = ns_tol * ns_tol;
  // A start with ||I - M X0||_F >= 0.3 is given up for the direct sweep, and the next three steps sweep unasked (PSMF_NS_FAR,
  // PSMF_NS_SKIP).  Below 1 the iteration would converge -- from 0.9 in seven iterations of 0.9 us against a 15 us sweep -- and
  // 0.9 / 0 takes config E's cold pass from 36.4 to 35.5 ms (111 -> 10 sweeps in its first 480 timesteps, tools/probe_cold.py);
  // NOT taken: the iteration stops at a residual (1e-7), the sweep is pivot-exact, and where Lbar' = (I / q - W / q^2) / omega
  // cancels (q = 1e-8, tests/adversarial_cases.py:tiny_Q) every early step iterated instead of swept costs accuracy -- y_hat error
  // 6.4e-7 (0.3 / 3), 4.0e-6 (0.3 / 0), 9.2e-6 (0.6 / 1), 1.26e-5 (0.6 / 0) against the 1e-5 bar (profiles/r4_adversarial_ns_far.txt).
  const double ns_far = getenv("PSMF_NS_FAR") ? atof(getenv("PSMF_NS_FAR")) : 0.3;
  sp.ns_far2 = ns_far * ns_far;
  sp.ns_skip_n = getenv("PSMF_NS_SKIP") ? atoi(getenv("PSMF_NS_SKIP")) : 3;
  h->ns_far_env = getenv("PSMF_NS_FAR") != nullptr;          // (otherwise psmf_set_state picks 0.9 / 1 where nothing cancels: update_ns_policy)
  h->ns_skip_env = getenv("PSMF_NS_SKIP") != nullptr;
  sp.alpha = cfg->alpha; sp.beta = cfg->beta;
  sp.lr = cfg->adam_lr; sp.lr_end = cfg->adam_lr_end; sp.lr_steps = cfg->adam_lr_steps;
  sp.b1 = cfg->adam_b1; sp.b2 = cfg->adam_b2;
  // the zero-fills above ran on the null stream; the handle's own streams are non-blocking
  if (hipDeviceSynchronize() != hipSuccess) { h->err = "hipDeviceSynchronize at the end of psmf_create failed"; return bail(PSMF_ERR_HIP); }
  *out = h;
  return PSMF_OK;
}

void psmf_destroy(psmf_handle h) {
  if (!h) return;
  hipSetDevice(h->cfg.device);
  if (h->stream) hipStreamSynchronize(h->stream);
  destroy_graph(h);
  if (h->comm) ncclCommDestroy(h->comm);
  if (h->st) hipFree(h->st);
  if (h->C) hipFree(h->C);
  if (h->Y) hipFree(h->Y);
  if (h->YP) hipFree(h->YP);
  if (h->partials) hipFree(h->partials);
  if (h->ps_prof) {
    long long pf[64];
    if (hipMemcpy(pf, h->ps_prof, sizeof(pf), hipMemcpyDeviceToHost) == hipSuccess && h->ps_prof_steps > 0) {
      fprintf(stderr, "[pstep prof] cycles per timestep over the last launch (%lld steps), d_local %d r %d:\n", h->ps_prof_steps, h->cfg.d_local, h->cfg.r);
      const char* grp[3] = {"hub workers", "solve wave ", "row wg 0   "};
      const int base[3] = {0, 16, 24}, cnt[3] = {11, 3, 8};
      for (int g = 0; g < 3; ++g) {
        fprintf(stderr, "  %s:", grp[g]);
        for (int i = 0; i < cnt[g]; ++i) fprintf(stderr, " %7.0f", (double)pf[base[g] + i] / (double)h->ps_prof_steps);
        fprintf(stderr, "\n");
      }
    }
    hipFree(h->ps_prof);
  }
  if (h->ps_comm) hipFree(h->ps_comm);
  if (h->gpart) hipFree(h->gpart);
  if (h->mu_hist) hipFree(h->mu_hist);
  if (h->thbuf) hipFree(h->thbuf);
  if (h->sched) hipFree(h->sched);
  if (h->qmat) hipFree(h->qmat);
  if (h->rho_rows) hipFree(h->rho_rows);
  if (h->rotU) hipFree(h->rotU);
  if (h->rot_tmp) hipFree(h->rot_tmp);
  if (h->mask) hipFree(h->mask);
  if (h->mmiss) hipFree(h->mmiss);
  if (h->mg) hipFree(h->mg);
  if (h->sc_hist) hipFree(h->sc_hist);
  if (h->Kpart) hipFree(h->Kpart);
  if (h->Kmat) hipFree(h->Kmat);
  if (h->Acoef) hipFree(h->Acoef);
  if (h->Bcoef) hipFree(h->Bcoef);
  if (h->XGpart) hipFree(h->XGpart);
  if (h->XG) hipFree(h->XG);
  if (h->flags) hipFree(h->flags);
  for (int i = 0; i < 4; ++i) { if (h->evF[i]) hipEventDestroy(h->evF[i]); if (h->evA[i]) hipEventDestroy(h->evA[i]); if (h->evX[i]) hipEventDestroy(h->evX[i]); }
  if (h->evS) hipEventDestroy(h->evS);
  if (h->evC) hipEventDestroy(h->evC);
  if (h->host_buf) hipHostFree(h->host_buf);
  for (int i = 0; i < psmf_filter::kTimedRuns; ++i) { if (h->evK0[i]) hipEventDestroy(h->evK0[i]); if (h->evK1[i]) hipEventDestroy(h->evK1[i]); }
  if (h->bulk) { hipStreamSynchronize(h->bulk); hipStreamDestroy(h->bulk); }
  if (h->fstream) { hipStreamSynchronize(h->fstream); hipStreamDestroy(h->fstream); }
  if (h->scratch) hipFree(h->scratch);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->err_host) hipHostFree(h->err_host);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int psmf_set_state(psmf_handle h, const double* C, const double* V, const double* P, const double* Q,
                   const double* mu, double rho, double lambda0, const double* theta) {
  if (!h) return PSMF_ERR_ARG;
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const int r = h->cfg.r, dl = h->cfg.d_local, rp = h->geo.rp;
  if (C) {
    if (h->cfg.storage == PSMF_F64) {
      std::vector<double> buf((size_t)dl * rp);
      pack_rows<double>(C, buf.data(), dl, r, rp);
      HIP_TRY(h, hipMemcpy(h->C, buf.data(), buf.size() * 8, hipMemcpyHostToDevice));
    } else {
      std::vector<float> buf((size_t)dl * rp);
      pack_rows<float>(C, buf.data(), dl, r, rp);
      HIP_TRY(h, hipMemcpy(h->C, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
    }
    if (h->rotU) {                       // non-diagonal R: the handle keeps U^T C
      const size_t cb = (size_t)dl * rp * h->elem();
      rc = ensure_rot_tmp(h, cb);
      if (rc) return rc;
      HIP_TRY(h, hipMemcpyAsync(h->rot_tmp, h->C, cb, hipMemcpyDeviceToDevice, h->stream));     // (stream-ordered with the GEMM: a plain D2D hipMemcpy does not wait on the host side)
      rc = rot_dict(h, h->rot_tmp, h->C, true);
      if (rc) return rc;
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  const size_t rr = (size_t)r * r * sizeof(double);
  if (V) HIP_TRY(h, hipMemcpy(h->st->V, V, rr, hipMemcpyHostToDevice));
  if (P) {
    HIP_TRY(h, hipMemcpy(h->st->P, P, rr, hipMemcpyHostToDevice));
    double m = 0.0;
    for (int i = 0; i < r; ++i) m = std::fmax(m, std::fabs(P[(size_t)i * r + i]));
    h->p_diag_max = m;
    if (!Q) update_solve_dual(h);
  }
  if (Q) {
    HIP_TRY(h, hipMemcpy(h->st->Q, Q, rr, hipMemcpyHostToDevice));
    bool iso = Q[0] > 0.0;
    for (int i = 0; i < r && iso; ++i)
      for (int c = 0; c < r; ++c)
        if (Q[i * r + c] != (i == c ? Q[0] : 0.0)) { iso = false; break; }
    h->q_iso = iso;
    h->q_last = Q[0];
    update_solve_dual(h);
  }
  if (mu) HIP_TRY(h, hipMemcpy(h->st->mu, mu, r * sizeof(double), hipMemcpyHostToDevice));
  if (theta && h->cfg.n_theta > 0)
    HIP_TRY(h, hipMemcpy(h->sp.theta, theta, h->cfg.n_theta * sizeof(double), hipMemcpyHostToDevice));
  if (!std::isnan(rho)) HIP_TRY(h, hipMemcpy(&h->st->rho, &rho, sizeof(double), hipMemcpyHostToDevice));
  if (!std::isnan(lambda0)) HIP_TRY(h, hipMemcpy(&h->st->lam, &lambda0, sizeof(double), hipMemcpyHostToDevice));
  { const int zero = 0; HIP_TRY(h, hipMemcpy(&h->st->ns_valid, &zero, sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(&h->st->err, &zero, sizeof(int), hipMemcpyHostToDevice)); }   // a new state clears a sticky numeric error
  if (P || Q) update_ns_policy(h);
  if (C && V && P && mu) h->have_state = true;
  h->need_prep = true;
  return PSMF_OK;
}

int psmf_zero_gradsum(psmf_handle h) {
  if (!h) return PSMF_ERR_ARG;
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipMemsetAsync(h->sp.gradsum, 0, sizeof(double) * h->th_cap, h->stream));
  return PSMF_OK;
}

int psmf_set_adam(psmf_handle h, const double* m, const double* v) {
  if (!h) return PSMF_ERR_ARG;
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const size_t nb = (size_t)h->cfg.n_theta * sizeof(double);
  if (m && nb) HIP_TRY(h, hipMemcpy(h->sp.adam_m, m, nb, hipMemcpyHostToDevice));
  if (v && nb) HIP_TRY(h, hipMemcpy(h->sp.adam_v, v, nb, hipMemcpyHostToDevice));
  return PSMF_OK;
}

int psmf_get_state(psmf_handle h, double* C, double* V, double* P, double* Q, double* mu, double* theta,
                   double* gradsum, double* scalars) {
  if (!h) return PSMF_ERR_ARG;
  int rc = psmf_sync(h);
  if (rc) return rc;
  const int r = h->cfg.r, dl = h->cfg.d_local, rp = h->geo.rp;
  if (C) {
    const void* Csrc = h->C;
    if (h->rotU) {                       // non-diagonal R: back to the caller's coordinates, C = U (U^T C)
      const size_t cb = (size_t)dl * rp * h->elem();
      rc = ensure_rot_tmp(h, cb);
      if (rc) return rc;
      HIP_TRY(h, hipMemsetAsync(h->rot_tmp, 0, cb, h->stream));
      rc = rot_dict(h, h->C, h->rot_tmp, false);
      if (rc) return rc;
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      Csrc = h->rot_tmp;
    }
    if (h->cfg.storage == PSMF_F64) {
      std::vector<double> buf((size_t)dl * rp);
      HIP_TRY(h, hipMemcpy(buf.data(), Csrc, buf.size() * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < dl; ++i) for (int c = 0; c < r; ++c) C[(size_t)i * r + c] = buf[(size_t)i * rp + c];
    } else {
      std::vector<float> buf((size_t)dl * rp);
      HIP_TRY(h, hipMemcpy(buf.data(), Csrc, buf.size() * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < dl; ++i) for (int c = 0; c < r; ++c) C[(size_t)i * r + c] = (double)buf[(size_t)i * rp + c];
    }
  }
  const size_t rr = (size_t)r * r * sizeof(double);
  if (V) HIP_TRY(h, hipMemcpy(V, h->st->V, rr, hipMemcpyDeviceToHost));
  if (P) HIP_TRY(h, hipMemcpy(P, h->st->P, rr, hipMemcpyDeviceToHost));
  if (Q) HIP_TRY(h, hipMemcpy(Q, h->st->Q, rr, hipMemcpyDeviceToHost));
  if (mu) HIP_TRY(h, hipMemcpy(mu, h->st->mu, r * sizeof(double), hipMemcpyDeviceToHost));
  if (theta && h->cfg.n_theta) HIP_TRY(h, hipMemcpy(theta, h->sp.theta, h->cfg.n_theta * sizeof(double), hipMemcpyDeviceToHost));
  if (gradsum && h->cfg.n_theta) HIP_TRY(h, hipMemcpy(gradsum, h->sp.gradsum, h->cfg.n_theta * sizeof(double), hipMemcpyDeviceToHost));
  if (scalars) {
    DevState* s = h->st;
    double tmp[12];  // rho lam s eta N kappa phi omega ee s_done eta_done N_done
    HIP_TRY(h, hipMemcpy(tmp, &s->rho, sizeof(double) * 12, hipMemcpyDeviceToHost));
    long long k;
    HIP_TRY(h, hipMemcpy(&k, &s->k, sizeof(k), hipMemcpyDeviceToHost));
    scalars[0] = tmp[0]; scalars[1] = tmp[1]; scalars[2] = tmp[9]; scalars[3] = tmp[10];
    scalars[4] = tmp[11]; scalars[5] = tmp[6]; scalars[6] = tmp[7]; scalars[7] = (double)k;
  }
  return PSMF_OK;
}

int psmf_upload_series(psmf_handle h, const void* Y, int dtype, int64_t t0, int64_t nt, int64_t T_total) {
  if (!h || !Y || nt < 0 || t0 < 0) return fail(h, PSMF_ERR_ARG, "psmf_upload_series: bad argument");
  if (dtype != PSMF_F32 && dtype != PSMF_F64) return fail(h, PSMF_ERR_ARG, "psmf_upload_series: dtype");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const size_t dl = h->cfg.d_local, es = h->elem();
  if (T_total < t0 + nt) T_total = t0 + nt;
  if (!h->Y || T_total > h->T_cap) {
    if (h->Y && t0 != 0) return fail(h, PSMF_ERR_STATE, "psmf_upload_series: buffer would grow mid-series; pass T_total on the first block");
    destroy_graph(h);   // graph nodes hold the old buffer addresses
    if (h->Y) HIP_TRY(h, hipFree(h->Y));
    if (h->YP) HIP_TRY(h, hipFree(h->YP));
    if (h->mu_hist) HIP_TRY(h, hipFree(h->mu_hist));
    h->Y = h->YP = nullptr;
    h->mu_hist = nullptr;
    if (h->cfg.masked) {
      if (h->mask) HIP_TRY(h, hipFree(h->mask));
      if (h->sc_hist) HIP_TRY(h, hipFree(h->sc_hist));
      h->mask = nullptr; h->sc_hist = nullptr; h->have_mask = false;
      HIP_TRY(h, hipMalloc((void**)&h->mask, (size_t)T_total * dl));
      HIP_TRY(h, hipMalloc((void**)&h->sc_hist, (size_t)T_total * 2 * sizeof(double)));
      HIP_TRY(h, hipMemset(h->sc_hist, 0, (size_t)T_total * 2 * sizeof(double)));
      h->sp.mask = h->mask;
      h->sp.mg = h->mg;
      h->sp.mg_tr = h->mg + (h->cfg.r * h->cfg.r + 1) + 1;
      h->sp.mg_ntr = (h->cfg.r * h->cfg.r + 1 + 63) / 64;
      h->sp.sc_hist = h->sc_hist;
      h->sp.mask_rows = (int)T_total;
    }
    HIP_TRY(h, hipMalloc(&h->Y, (size_t)T_total * dl * es));
    if (h->cfg.store_y_pred) HIP_TRY(h, hipMalloc(&h->YP, (size_t)T_total * dl * es));
    HIP_TRY(h, hipMalloc((void**)&h->mu_hist, (size_t)(T_total + 1) * h->cfg.r * sizeof(double)));
    h->sp.mu_hist = h->mu_hist;
    h->T_cap = T_total;
    h->sp.Y = h->Y;
    h->sp.YP = h->YP;
    h->sp.store_yp = h->cfg.store_y_pred ? 1 : 0;
    h->sp.series_t0 = 0;
  }
  char* dst = (char*)h->Y + (size_t)t0 * dl * es;
  const size_t n = (size_t)nt * dl;
  if ((dtype == PSMF_F64) == (h->cfg.storage == PSMF_F64)) {
    HIP_TRY(h, hipMemcpy(dst, Y, n * es, hipMemcpyHostToDevice));
  } else {
    const size_t blk = (size_t)1 << 24;
    if (h->cfg.storage == PSMF_F32) {
      std::vector<float> buf(n < blk ? n : blk);
      const double* src = (const double*)Y;
      for (size_t a = 0; a < n; a += blk) {
        const size_t m = n - a < blk ? n - a : blk;
        for (size_t i = 0; i < m; ++i) buf[i] = (float)src[a + i];
        HIP_TRY(h, hipMemcpy(dst + a * 4, buf.data(), m * 4, hipMemcpyHostToDevice));
      }
    } else {
      std::vector<double> buf(n < blk ? n : blk);
      const float* src = (const float*)Y;
      for (size_t a = 0; a < n; a += blk) {
        const size_t m = n - a < blk ? n - a : blk;
        for (size_t i = 0; i < m; ++i) buf[i] = (double)src[a + i];
        HIP_TRY(h, hipMemcpy(dst + a * 8, buf.data(), m * 8, hipMemcpyHostToDevice));
      }
    }
  }
  if (h->rotU && n) {                    // non-diagonal R: the handle keeps the rows y^T U
    rc = ensure_rot_tmp(h, n * es);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->rot_tmp, dst, n * es, hipMemcpyDeviceToDevice, h->stream));
    rc = rot_rows(h, h->rot_tmp, dst, nt, true);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return PSMF_OK;
}

int psmf_run(psmf_handle h, int64_t k_begin, int64_t k_end) {
  if (!h) return PSMF_ERR_ARG;
  if (!h->have_state) return fail(h, PSMF_ERR_STATE, "psmf_run: set_state (C, V, P, mu) first");
  if (!h->Y) return fail(h, PSMF_ERR_STATE, "psmf_run: upload_series first");
  if (k_begin < 0 || k_end < k_begin || k_end > h->T_cap) return fail(h, PSMF_ERR_ARG, "psmf_run: step range outside the uploaded series");
  if (h->cfg.dyn_kind == PSMF_DYN_HOST) return fail(h, PSMF_ERR_STATE, "psmf_run: host-stepped dynamics advance with psmf_step_host");
  if (h->sched && k_end >= h->sched_n) return fail(h, PSMF_ERR_ARG, "psmf_run: step range beyond the R / Q schedules");
  if (h->qmat && k_end >= h->qmat_n) return fail(h, PSMF_ERR_ARG, "psmf_run: step range beyond the Q_k matrix schedule");
  if (h->cfg.nonuniform_R && !h->sp.rho_rows) return fail(h, PSMF_ERR_STATE, "psmf_run: psmf_set_row_noise first (nonuniform_R = 1)");
  if (h->cfg.masked && !h->have_mask) return fail(h, PSMF_ERR_STATE, "psmf_run: psmf_upload_mask first (masked = 1)");
  int rc = set_device(h);
  if (rc) return rc;
  if (h->need_prep || h->k_done != k_begin) {
    rc = prepare(h, k_begin);
    if (rc) return rc;
  }
  int64_t n = k_end - k_begin;
  if (h->engine == 2) {
    const bool pipe_off = !h->sw.block_pipe;
    if (!pipe_off && k_end - k_begin > h->block_steps) {
      rc = enqueue_blocks_pipelined(h, k_begin, k_end);
      if (rc) return rc;
      h->k_done = k_end;
      return PSMF_OK;
    }
    int64_t k = k_begin;
    while (k < k_end) {
      const int nb = (int)((k_end - k) < h->block_steps ? (k_end - k) : h->block_steps);
      rc = enqueue_block(h, k, nb);
      if (rc) return rc;
      k += nb;
    }
    HIP_TRY(h, hipGetLastError());
    h->k_done = k_end;
    return PSMF_OK;
  }
  const int64_t refresh = h->cfg.gram_refresh > 0 && h->sp.track_g ? h->cfg.gram_refresh : 0;
  while (n > 0) {
    int64_t seg = n;
    if (refresh) {
      const int64_t to_next = refresh - (h->k_done % refresh);
      if (to_next < seg) seg = to_next;
    }
    int64_t left = seg;
    if (pstep_usable(h)) {      // the whole segment in ONE launch: C on chip, hand-offs through device flags (psmf_pstep.hip)
      rc = launch_pstep(h, h->k_done, left);
      if (rc) return rc;
      left = 0;
    }
    if (h->cfg.use_graph && !h->host_fn) {   // (a host-mediated all-reduce cannot be captured)
      const int want = 256;
      if (left >= want && h->chunk != want) {
        rc = build_graph(h, want);
        if (rc) return rc;
      }
      while (h->chunk > 0 && left >= h->chunk) {
        HIP_TRY(h, hipGraphLaunch(h->gexec, h->stream));
        left -= h->chunk;
      }
    }
    for (; left > 0; --left) {
      rc = enqueue_step(h);
      if (rc) return rc;
    }
    HIP_TRY(h, hipGetLastError());
    h->k_done += seg;
    n -= seg;
    if (refresh && n > 0 && h->k_done % refresh == 0) {
      rc = enqueue_gram(h);
      if (rc) return rc;
      launch_serial(h, 1);   // recompute eta / N / kappa of the next step with the exact Gram
    }
  }
  return PSMF_OK;
}

int psmf_sync(psmf_handle h) {
  if (!h) return PSMF_ERR_ARG;
  int rc = set_device(h);
  if (rc) return rc;
  const double t_s0 = g_host_timing ? host_now_ms() : 0.0;
  // (a 4-byte device-to-host hipMemcpy[Async] here now and then took 20-70 ms on this stack -- the copy engine waking
  //  up -- which is half a pass of the headline workload; a store from a kernel to mapped host memory does not)
  hipLaunchKernelGGL(psmf::psmf_publish_err_k, dim3(1), dim3(1), 0, h->stream, (const DevState*)h->st, h->err_host_dev);
  const double t_s1 = g_host_timing ? host_now_ms() : 0.0;
  HIP_TRY(h, spin_stream(h->stream));
  for (int i = 0; i < h->evk_pending; ++i) {         // the chained filter launches that finished: their durations
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->evK0[i], h->evK1[i]) == hipSuccess) { h->kernel_ms_sum += ms; ++h->kernel_launches; }
    else (void)hipGetLastError();
  }
  h->evk_pending = 0;
  if (g_host_timing) { const double t = host_now_ms(); if (t_s1 - t_s0 > 5.0) fprintf(stderr, "[psmf host timing] memcpyAsync call %.1f ms\n", t_s1 - t_s0); if (t - t_s1 > 5.0) fprintf(stderr, "[psmf host timing] spin wait %.1f ms\n", t - t_s1); }
  const int err = *h->err_host;
  if (err == -7) {
    long long fl[8] = {0};
    if (h->flags) (void)hipMemcpy(fl, h->flags, sizeof(fl), hipMemcpyDeviceToHost);
    char msg[320];
    snprintf(msg, sizeof(msg), "pipelined blocks: a device-flag hand-off timed out (PSMF_BLOCK_CHAIN=0: one filter launch per block; "
             "PSMF_BLOCK_FLAGS=0: event hand-off); flags: cross-Gram %lld, filter %lld, next block to enqueue %lld",
             fl[0], fl[1], h->seq_next);
    return fail(h, PSMF_ERR_HIP, msg);
  }
  if (err == -8) return fail(h, PSMF_ERR_HIP, "persistent per-step kernel: a hand-off between the hub and the row workgroups timed out "
                                               "(PSMF_STEP_PERSISTENT=0: two launches per timestep)");
  if (err != 0) {
    char msg[128];
    snprintf(msg, sizeof(msg), "singular r x r system (I + kappa Pbar G) at step %d", err);
    return fail(h, PSMF_ERR_NUMERIC, msg);
  }
  return PSMF_OK;
}

int psmf_run_timed(psmf_handle h, int64_t k_begin, int64_t k_end, float* ms) {
  if (!h || !ms) return PSMF_ERR_ARG;
  int rc = set_device(h);
  if (rc) return rc;
  if (h->have_state && h->Y && (h->need_prep || h->k_done != k_begin)) {
    rc = prepare(h, k_begin);   // keep the one-off preparation outside the timed region
    if (rc) return rc;
  }
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  rc = psmf_run(h, k_begin, k_end);
  if (rc) return rc;
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, spin_event(h->ev1));
  HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return psmf_sync(h);
}

int psmf_time_kernel(psmf_handle h, int which, int iters, float* avg_us) {
  if (!h || !avg_us || iters < 1 || which < 0 || which > 2) return PSMF_ERR_ARG;
  if (!h->have_state || !h->Y) return fail(h, PSMF_ERR_STATE, "psmf_time_kernel: needs state and series");
  int rc = set_device(h);
  if (rc) return rc;
  if (h->engine == 2) {
    const int nb = (int)(h->T_cap < h->block_steps ? h->T_cap : h->block_steps);
    psmf::BlockParams b;
    fill_block_params(h, b, h->sp.series_t0, nb);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t cbytes = (size_t)h->cfg.d_local * h->geo.rp * h->elem();
    void* Csave = nullptr; DevState* ssave = nullptr; double* thsave = nullptr;
    const size_t thbytes = 4 * h->th_cap * sizeof(double);     // theta, gradient sums, Adam moments: the filter kernels step them too
    HIP_TRY(h, hipMalloc(&Csave, cbytes));
    HIP_TRY(h, hipMalloc((void**)&ssave, sizeof(DevState)));
    HIP_TRY(h, hipMalloc((void**)&thsave, thbytes));
    HIP_TRY(h, hipMemcpy(Csave, h->C, cbytes, hipMemcpyDeviceToDevice));
    HIP_TRY(h, hipMemcpy(ssave, h->st, sizeof(DevState), hipMemcpyDeviceToDevice));
    HIP_TRY(h, hipMemcpy(thsave, h->thbuf, thbytes, hipMemcpyDeviceToDevice));
    HIP_TRY(h, hipDeviceSynchronize());      // (device-to-device copies on the null stream are not ordered against the handle's non-blocking stream)
    launch_blk_gram(h, b);           // a valid K for the filter / apply measurements
    launch_blk_filter(h, b);
    auto one = [&]() {
      if (which == 0) { HIP_TRY(h, hipMemcpyAsync(h->st, ssave, sizeof(DevState), hipMemcpyDeviceToDevice, h->stream)); launch_blk_filter(h, b); }
      else if (which == 1) {
        // the per-block d-sized contraction: the cross-Gram for the next block (+ reduction) when the series holds
        // two blocks, else the plain block Gram
        if (h->T_cap >= 2 * (int64_t)nb) {
          psmf::BlockParams x = b;
          x.k1 = b.k0 + nb; x.nb1 = nb;
          launch_blk_xgram(h, x, h->XG, h->stream);
        } else {
          launch_blk_gram(h, b);
        }
      }
      else launch_blk_apply(h, b);
      return (int)PSMF_OK;
    };
    for (int i = 0; i < 2; ++i) { rc = one(); if (rc) return rc; }
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < iters; ++i) { rc = one(); if (rc) return rc; }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_us = ms * 1000.f / iters;
    HIP_TRY(h, hipMemcpy(h->C, Csave, cbytes, hipMemcpyDeviceToDevice));
    HIP_TRY(h, hipMemcpy(h->st, ssave, sizeof(DevState), hipMemcpyDeviceToDevice));
    HIP_TRY(h, hipMemcpy(h->thbuf, thsave, thbytes, hipMemcpyDeviceToDevice));
    HIP_TRY(h, hipFree(Csave));
    HIP_TRY(h, hipFree(ssave));
    HIP_TRY(h, hipFree(thsave));
    return PSMF_OK;
  }
  if (h->need_prep) { rc = prepare(h, h->k_done); if (rc) return rc; }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  // save everything the kernels mutate
  const size_t cbytes = (size_t)h->cfg.d_local * h->geo.rp * h->elem();
  void* Csave = nullptr; DevState* ssave = nullptr;
  HIP_TRY(h, hipMalloc(&Csave, cbytes));
  HIP_TRY(h, hipMalloc((void**)&ssave, sizeof(DevState)));
  HIP_TRY(h, hipMemcpy(Csave, h->C, cbytes, hipMemcpyDeviceToDevice));
  HIP_TRY(h, hipMemcpy(ssave, h->st, sizeof(DevState), hipMemcpyDeviceToDevice));
  HIP_TRY(h, hipDeviceSynchronize());
  {  // the sweep reads y_k / writes y_hat_k at the step counter: point it at a valid row of the series
    long long k0 = h->sp.series_t0;
    HIP_TRY(h, hipMemcpy(&h->st->k, &k0, sizeof(k0), hipMemcpyHostToDevice));
  }
  // which = 1 times the full serial stage (first = 0: reduction of the partials left by the last
  // sweep, updates, next-step preparation); the step counter it increments is reset with the state
  for (int i = 0; i < 3; ++i) { if (which == 0) launch_sweep(h); else launch_serial(h, 0); }
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < iters; ++i) { if (which == 0) launch_sweep(h); else launch_serial(h, 0); }
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_us = ms * 1000.f / iters;
  HIP_TRY(h, hipMemcpy(h->C, Csave, cbytes, hipMemcpyDeviceToDevice));
  HIP_TRY(h, hipMemcpy(h->st, ssave, sizeof(DevState), hipMemcpyDeviceToDevice));
  HIP_TRY(h, hipFree(Csave));
  HIP_TRY(h, hipFree(ssave));
  return PSMF_OK;
}

int psmf_geometry(psmf_handle h, int32_t* out7) {
  if (!h || !out7) return PSMF_ERR_ARG;
  out7[0] = h->geo.n_sweep_wg; out7[1] = h->geo.rows_per_wg; out7[2] = h->geo.rp; out7[3] = h->geo.gs;
  out7[4] = h->chunk; out7[5] = h->engine; out7[6] = h->block_steps;
  return PSMF_OK;
}

#ifdef F4_DEBUG
// debug builds only (tools/probe_f4d.py): the scratch area the filter4 kernel writes its per-step residuals to
int psmf_debug_read(psmf_handle h, double* out, int n) {
  if (!h || !out) return PSMF_ERR_ARG;
  int rc = psmf_sync(h);
  if (rc) return rc;
  HIP_TRY(h, hipMemcpy(out, h->st->GR, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return PSMF_OK;
}
#endif

int psmf_filter_kernel(psmf_handle h) {
  if (!h) return PSMF_ERR_ARG;
  return (int)select_filter_kernel(h);
}

int psmf_counters(psmf_handle h, int64_t* out8, int reset) {
  if (!h || !out8) return PSMF_ERR_ARG;
  if (set_device(h) != PSMF_OK) return PSMF_ERR_HIP;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  long long c[8], g[8];
  HIP_TRY(h, hipMemcpy(c, h->st->cnt, sizeof(c), hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(g, h->st->dbg, sizeof(g), hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; ++i) out8[i] = c[i];
  out8[6] = g[5];       // kernel launches of psmf_blk_filter3 (cnt[7] counts blocks; cnt[6] is a raw time stamp)
  if (getenv("PSMF_DBG_BREAKDOWN") && c[7] > 0)
    fprintf(stderr, "[psmf] filter3 per launch: hand-off %.2f us, K %.2f, init %.2f, steps %.2f, end %.2f\n", 0.01 * g[0] / c[7], 0.01 * g[1] / c[7],
            0.01 * g[2] / c[7], 0.01 * g[3] / c[7], 0.01 * g[4] / c[7]);
  if (reset) { HIP_TRY(h, hipMemset(h->st->cnt, 0, sizeof(c))); HIP_TRY(h, hipMemset(h->st->dbg, 0, sizeof(g))); HIP_TRY(h, hipDeviceSynchronize()); }   // (before the next run's kernels on the non-blocking stream count)
  return PSMF_OK;
}

int psmf_filter_kernel_time(psmf_handle h, int64_t* launches, double* total_ms, int reset) {
  if (!h || !launches || !total_ms) return PSMF_ERR_ARG;
  int rc = psmf_sync(h);
  if (rc) return rc;
  *launches = h->kernel_launches;
  *total_ms = h->kernel_ms_sum;
  if (reset) { h->kernel_launches = 0; h->kernel_ms_sum = 0.0; }
  return PSMF_OK;
}

int psmf_download_y_pred(psmf_handle h, void* out, int dtype, int64_t t0, int64_t nt) {
  if (!h || !out) return PSMF_ERR_ARG;
  if (!h->YP) return fail(h, PSMF_ERR_STATE, "psmf_download_y_pred: handle was created with store_y_pred = 0");
  if (t0 < 0 || nt < 0 || t0 + nt > h->T_cap) return fail(h, PSMF_ERR_ARG, "psmf_download_y_pred: range");
  int rc = psmf_sync(h);
  if (rc) return rc;
  const size_t dl = h->cfg.d_local, es = h->elem(), n = (size_t)nt * dl;
  const char* src = (const char*)h->YP + (size_t)t0 * dl * es;
  if (h->rotU && n) {                    // non-diagonal R: y_hat = U (U^T y_hat)
    rc = ensure_rot_tmp(h, n * es);
    if (rc) return rc;
    rc = rot_rows(h, src, h->rot_tmp, nt, false);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    src = (const char*)h->rot_tmp;
  }
  if ((dtype == PSMF_F64) == (h->cfg.storage == PSMF_F64)) {
    HIP_TRY(h, hipMemcpy(out, src, n * es, hipMemcpyDeviceToHost));
  } else if (h->cfg.storage == PSMF_F32) {
    std::vector<float> buf(n);
    HIP_TRY(h, hipMemcpy(buf.data(), src, n * 4, hipMemcpyDeviceToHost));
    double* o = (double*)out;
    for (size_t i = 0; i < n; ++i) o[i] = (double)buf[i];
  } else {
    std::vector<double> buf(n);
    HIP_TRY(h, hipMemcpy(buf.data(), src, n * 8, hipMemcpyDeviceToHost));
    float* o = (float*)out;
    for (size_t i = 0; i < n; ++i) o[i] = (float)buf[i];
  }
  return PSMF_OK;
}

int psmf_download_mu(psmf_handle h, double* out, int64_t k0, int64_t nk) {
  if (!h || !out) return PSMF_ERR_ARG;
  if (!h->mu_hist) return fail(h, PSMF_ERR_STATE, "psmf_download_mu: no series uploaded yet");
  if (k0 < 0 || nk < 0 || k0 + nk > h->T_cap + 1) return fail(h, PSMF_ERR_ARG, "psmf_download_mu: range");
  int rc = psmf_sync(h);
  if (rc) return rc;
  HIP_TRY(h, hipMemcpy(out, h->mu_hist + (size_t)k0 * h->cfg.r, (size_t)nk * h->cfg.r * sizeof(double), hipMemcpyDeviceToHost));
  return PSMF_OK;
}

int psmf_predict(psmf_handle h, int64_t T, int64_t n_pred, double* out) {
  if (!h || !out || n_pred < 0) return PSMF_ERR_ARG;
  if (n_pred == 0) return PSMF_OK;
  int rc = psmf_sync(h);
  if (rc) return rc;
  if (h->cfg.dyn_kind == PSMF_DYN_HOST) return fail(h, PSMF_ERR_STATE, "psmf_predict: host-stepped dynamics roll mu forward on the host; use psmf_project");
  const int r = h->cfg.r;
  std::vector<double> mu(r), theta((size_t)(h->cfg.n_theta > 0 ? h->cfg.n_theta : 1), 0.0), mup((size_t)n_pred * r), nx(r);
  HIP_TRY(h, hipMemcpy(mu.data(), h->st->mu, r * sizeof(double), hipMemcpyDeviceToHost));
  if (h->cfg.n_theta) HIP_TRY(h, hipMemcpy(theta.data(), h->sp.theta, h->cfg.n_theta * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t q = 0; q < n_pred; ++q) {   // psmf.py:183-187, r-sized: done on the host
    dyn_f_host(h->cfg, theta.data(), mu.data(), (double)(T + q + 1), nx.data());
    mu = nx;
    for (int i = 0; i < r; ++i) mup[(size_t)q * r + i] = mu[i];
  }
  return psmf_project(h, mup.data(), n_pred, out);
}

int psmf_project(psmf_handle h, const double* mu, int64_t n_pred, double* out) {
  if (!h || !mu || !out || n_pred < 0) return PSMF_ERR_ARG;
  if (n_pred == 0) return PSMF_OK;
  int rc = psmf_sync(h);
  if (rc) return rc;
  const int r = h->cfg.r, dl = h->cfg.d_local;
  const size_t nmu = (size_t)n_pred * r;
  const size_t mbytes = nmu * sizeof(double), obytes = (size_t)n_pred * dl * sizeof(double);
  rc = ensure_scratch(h, mbytes + obytes);
  if (rc) return rc;
  double* dmu = h->scratch;
  double* dout = h->scratch + nmu;
  HIP_TRY(h, hipMemcpy(dmu, mu, mbytes, hipMemcpyHostToDevice));
  const int grid = (dl + psmf::WG - 1) / psmf::WG;
  const size_t lds = (size_t)64 * r * sizeof(double);
  if (h->cfg.storage == PSMF_F64)
    hipLaunchKernelGGL(psmf::psmf_predict_rows<double>, dim3(grid), dim3(psmf::WG), lds, h->stream,
                       (const double*)h->C, dl, r, h->geo.rp, (const double*)dmu, (int)n_pred, dout);
  else
    hipLaunchKernelGGL(psmf::psmf_predict_rows<float>, dim3(grid), dim3(psmf::WG), lds, h->stream,
                       (const float*)h->C, dl, r, h->geo.rp, (const double*)dmu, (int)n_pred, dout);
  HIP_TRY(h, hipGetLastError());
  if (h->rotU) {                         // non-diagonal R: C here is U^T C -- rotate the projections back
    rc = ensure_rot_tmp(h, obytes);
    if (rc) return rc;
    rc = rot_gemm(h, (const double*)dout, (long long)dl, 1LL, (const double*)h->rotU, 1LL, (long long)dl, (double*)h->rot_tmp, (long long)dl,
                  (long long)n_pred, (long long)dl, (long long)dl);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->rot_tmp, obytes, hipMemcpyDeviceToHost));
    return PSMF_OK;
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out, dout, obytes, hipMemcpyDeviceToHost));
  return PSMF_OK;
}

int psmf_predict_sq_error(psmf_handle h, int64_t T, int64_t n_pred, const double* Y_true, double* out) {
  if (!h || !Y_true || !out || n_pred < 0) return PSMF_ERR_ARG;
  *out = 0.0;
  if (n_pred == 0) return PSMF_OK;
  const size_t dl = h->cfg.d_local, n = (size_t)n_pred * dl;
  std::vector<double> yp(n);
  int rc = psmf_predict(h, T, n_pred, yp.data());      // leaves the roll-out in the scratch buffer: [mu_pred | y_hat]
  if (rc) return rc;
  const size_t off = (size_t)n_pred * h->cfg.r;
  const int grid = 1024;
  rc = ensure_scratch(h, (off + 2 * n + grid) * sizeof(double));      // (may move the buffer: upload the roll-out again)
  if (rc) return rc;
  double* dyp = h->scratch + off;
  double* dyt = dyp + n;
  double* part = dyt + n;
  HIP_TRY(h, hipMemcpy(dyp, yp.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(dyt, Y_true, n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(psmf::psmf_sq_error_k<double>, dim3(grid), dim3(psmf::WG), 0, h->stream, (const double*)dyp, (const double*)dyt, n, part);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<double> hp(grid);
  HIP_TRY(h, hipMemcpy(hp.data(), part, grid * sizeof(double), hipMemcpyDeviceToHost));
  double a = 0.0;
  for (int i = 0; i < grid; ++i) a += hp[i];
  *out = a;
  return PSMF_OK;
}

int psmf_sq_error(psmf_handle h, int64_t t0, int64_t nt, double* out) {
  if (!h || !out) return PSMF_ERR_ARG;
  if (!h->YP || !h->Y) return fail(h, PSMF_ERR_STATE, "psmf_sq_error: needs store_y_pred and an uploaded series");
  if (t0 < 0 || nt < 0 || t0 + nt > h->T_cap) return fail(h, PSMF_ERR_ARG, "psmf_sq_error: range");
  int rc = set_device(h);
  if (rc) return rc;
  const int grid = 1024;
  rc = ensure_scratch(h, grid * sizeof(double));
  if (rc) return rc;
  const size_t dl = h->cfg.d_local, n = (size_t)nt * dl, off = (size_t)t0 * dl;
  if (h->cfg.storage == PSMF_F64)
    hipLaunchKernelGGL(psmf::psmf_sq_error_k<double>, dim3(grid), dim3(psmf::WG), 0, h->stream,
                       (const double*)h->YP + off, (const double*)h->Y + off, n, h->scratch);
  else
    hipLaunchKernelGGL(psmf::psmf_sq_error_k<float>, dim3(grid), dim3(psmf::WG), 0, h->stream,
                       (const float*)h->YP + off, (const float*)h->Y + off, n, h->scratch);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<double> part(grid);
  HIP_TRY(h, hipMemcpy(part.data(), h->scratch, grid * sizeof(double), hipMemcpyDeviceToHost));
  double a = 0.0;
  for (int i = 0; i < grid; ++i) a += part[i];
  *out = a;
  return PSMF_OK;
}

int psmf_comm_unique_id(void* id_out) {
  if (!id_out) return PSMF_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) <= PSMF_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return fail(nullptr, PSMF_ERR_RCCL, "ncclGetUniqueId failed");
  memset(id_out, 0, PSMF_UNIQUE_ID_BYTES);
  memcpy(id_out, &id, sizeof(id));
  return PSMF_OK;
}

int psmf_comm_init(psmf_handle h, int nranks, int rank, const void* unique_id) {
  if (!h || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, PSMF_ERR_ARG, "psmf_comm_init: bad argument");
  if (h->rotU) return fail(h, PSMF_ERR_STATE, "psmf_comm_init: a handle with a noise rotation (non-diagonal R) is one shard by construction");
  int rc = set_device(h);
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  NCCL_TRY(h, ncclCommInitRank(&h->comm, nranks, id, rank));
  h->nranks = nranks;
  h->rank = rank;
  {
    // Connect now: the first collective of each message size sets up its channels (seconds on 8 GPUs), and in the pipelined
    // block engine a filter kernel would be polling for its result meanwhile.  Same sizes, same streams as the engines use.
    const size_t xg_elems = (size_t)(psmf::RB + psmf::XGB) * psmf::XGB;
    double* tmp = nullptr;
    HIP_TRY(h, hipMalloc(&tmp, xg_elems * sizeof(double)));
    HIP_TRY(h, hipMemsetAsync(tmp, 0, xg_elems * sizeof(double), h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t sizes[4] = {(size_t)h->cfg.r + 1, (size_t)h->cfg.r * h->cfg.r, (size_t)psmf::RB * psmf::RB, xg_elems};
    hipStream_t streams[2] = {h->stream, h->bulk};
    for (int si = 0; si < 2; ++si) {
      if (!streams[si]) continue;
      for (int zi = 0; zi < 4; ++zi) {
        const ncclResult_t e = ncclAllReduce(tmp, tmp, sizes[zi], ncclDouble, ncclSum, h->comm, streams[si]);
        if (e != ncclSuccess) { hipFree(tmp); return fail(h, PSMF_ERR_RCCL, std::string("warm-up all-reduce: ") + ncclGetErrorString(e)); }
      }
      const hipError_t he = hipStreamSynchronize(streams[si]);
      if (he != hipSuccess) { hipFree(tmp); return fail(h, PSMF_ERR_HIP, std::string("warm-up all-reduce: ") + hipGetErrorString(he)); }
    }
    hipFree(tmp);
  }
  h->use_coll = nranks > 1 || h->sw.force_collective;
  h->sp.external_reduce = (h->use_coll || h->sp.tail_reduce) ? 1 : 0;
  destroy_graph(h);
  h->need_prep = true;
  return PSMF_OK;
}

int psmf_comm_abort(psmf_handle h) {
  if (!h) return PSMF_ERR_ARG;
  int rc = set_device(h);
  if (rc) return rc;
  if (h->comm) {
    // ncclCommAbort, not ncclCommDestroy: destroy waits for outstanding work and for its peers, and the caller is here because
    // some peer never joined (or stopped answering)
    const ncclResult_t e = ncclCommAbort(h->comm);
    h->comm = nullptr;
    if (e != ncclSuccess) return fail(h, PSMF_ERR_RCCL, std::string("ncclCommAbort: ") + ncclGetErrorString(e));
  }
  h->nranks = 1; h->rank = 0;
  h->use_coll = false;
  h->sp.external_reduce = h->sp.tail_reduce;
  destroy_graph(h);
  h->need_prep = true;
  return PSMF_OK;
}

int psmf_set_row_noise(psmf_handle h, const double* rho_rows, double rho_mean) {
  if (!h || !rho_rows || !(rho_mean > 0.0)) return fail(h, PSMF_ERR_ARG, "psmf_set_row_noise: bad argument");
  if (!h->cfg.nonuniform_R) return fail(h, PSMF_ERR_STATE, "psmf_set_row_noise: the handle was created with nonuniform_R = 0");
  for (int i = 0; i < h->cfg.d_local; ++i)
    if (!(rho_rows[i] >= 0.0)) return fail(h, PSMF_ERR_ARG, "psmf_set_row_noise: diag(R) must be non-negative");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (!h->rho_rows) HIP_TRY(h, hipMalloc((void**)&h->rho_rows, (size_t)h->cfg.d_local * sizeof(double)));
  HIP_TRY(h, hipMemcpy(h->rho_rows, rho_rows, (size_t)h->cfg.d_local * sizeof(double), hipMemcpyHostToDevice));
  h->sp.rho_rows = h->rho_rows;
  h->sp.rho_mean = rho_mean;
  destroy_graph(h);
  h->need_prep = true;
  return PSMF_OK;
}

int psmf_set_noise_rotation(psmf_handle h, const double* U, const double* lam) {
  if (!h || !U || !lam) return fail(h, PSMF_ERR_ARG, "psmf_set_noise_rotation: bad argument");
  if (!h->cfg.nonuniform_R) return fail(h, PSMF_ERR_STATE, "psmf_set_noise_rotation: the handle was created with nonuniform_R = 0");
  if (h->cfg.masked) return fail(h, PSMF_ERR_STATE, "psmf_set_noise_rotation: a masked handle filters per row of the ORIGINAL coordinates; not with a rotation");
  if (h->use_coll || h->cfg.d_local != h->cfg.d)
    return fail(h, PSMF_ERR_STATE, "psmf_set_noise_rotation: a non-diagonal R couples all rows: one shard only (d_local = d, no communicator)");
  if (h->Y || h->have_state) return fail(h, PSMF_ERR_STATE, "psmf_set_noise_rotation: call it before psmf_set_state / psmf_upload_series");
  const size_t d = (size_t)h->cfg.d;
  double tr = 0.0;
  for (size_t i = 0; i < d; ++i) {
    if (!(lam[i] >= 0.0)) return fail(h, PSMF_ERR_ARG, "psmf_set_noise_rotation: eigenvalues of R must be non-negative");
    tr += lam[i];
  }
  if (!(tr > 0.0)) return fail(h, PSMF_ERR_ARG, "psmf_set_noise_rotation: tr(R) must be positive");
  if (d > (size_t)PSMF_ROTATION_DMAX)
    return fail(h, PSMF_ERR_ARG, "psmf_set_noise_rotation: a non-diagonal R is supported up to d = 32768 (the handle keeps the d x d eigenvector matrix "
                                 "resident: 8 d^2 bytes); beyond that use a diagonal R or backend=\"numpy\"");
  // orthonormality over ALL of U in O(d^2): U^T (U z) = z for two probe vectors z (+-1 entries from a fixed generator).  A column
  // pair that is not orthonormal shows in (U^T U - I) z unless z lies in that matrix's null space -- two independent sign
  // patterns do not.  (The full check U^T U = I is O(d^3); a C caller with a wrong U used to get a silently wrong filter.)
  {
    std::vector<double> z(d), t(d), u(d);
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    for (int probe = 0; probe < 2; ++probe) {
      for (size_t i = 0; i < d; ++i) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; z[i] = (lcg >> 63) ? 1.0 : -1.0; }
      for (size_t i = 0; i < d; ++i) { double a = 0.0; const double* row = U + i * d; for (size_t k = 0; k < d; ++k) a += row[k] * z[k]; t[i] = a; }
      for (size_t k = 0; k < d; ++k) u[k] = 0.0;
      for (size_t i = 0; i < d; ++i) { const double ti = t[i]; const double* row = U + i * d; for (size_t k = 0; k < d; ++k) u[k] += row[k] * ti; }
      double worst = 0.0;
      for (size_t k = 0; k < d; ++k) worst = std::fmax(worst, std::fabs(u[k] - z[k]));
      if (!(worst <= 1e-8 * std::sqrt((double)d) + 1e-10))
        return fail(h, PSMF_ERR_ARG, "psmf_set_noise_rotation: the columns of U are not orthonormal (U^T U z != z)");
    }
  }
  int rc = psmf_set_row_noise(h, lam, tr / (double)d);
  if (rc) return rc;
  if (!h->rotU) HIP_TRY(h, hipMalloc((void**)&h->rotU, d * d * sizeof(double)));
  HIP_TRY(h, hipMemcpy(h->rotU, U, d * d * sizeof(double), hipMemcpyHostToDevice));
  return PSMF_OK;
}

int psmf_set_schedules(psmf_handle h, const double* rho_k, const double* q_k, int64_t n) {
  if (!h || n < 0) return PSMF_ERR_ARG;
  if (h->cfg.robust && (rho_k || q_k)) return fail(h, PSMF_ERR_ARG, "psmf_set_schedules: rPSMF runs on its own scaled Q, R (rpsmf.py:123,128,141)");
  if (h->cfg.masked && (rho_k || q_k)) return fail(h, PSMF_ERR_ARG, "psmf_set_schedules: masked handles take a constant R = rho I, Q (the ExperimentImpute filters)");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->sched) { HIP_TRY(h, hipFree(h->sched)); h->sched = nullptr; }
  h->sched_n = 0;
  h->sp.rho_sched = h->sp.q_sched = nullptr;
  if ((rho_k || q_k) && n > 0) {
    // n + 1 entries each, the last one repeated: having finished step k the per-step engine's serial stage prepares step
    // k + 1 and reads entry k + 1 -- one past the schedule on the last step of a run (the value is recomputed by the next prepare)
    std::vector<double> buf((size_t)2 * (n + 1), 1.0);
    if (rho_k) { memcpy(buf.data(), rho_k, (size_t)n * sizeof(double)); buf[n] = rho_k[n - 1]; }
    if (q_k) { memcpy(buf.data() + n + 1, q_k, (size_t)n * sizeof(double)); buf[2 * n + 1] = q_k[n - 1]; }
    HIP_TRY(h, hipMalloc((void**)&h->sched, buf.size() * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->sched, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice));
    h->sched_n = n;
    if (rho_k) h->sp.rho_sched = h->sched;
    if (q_k) h->sp.q_sched = h->sched + n + 1;
  }
  update_solve_dual(h);
  destroy_graph(h);      // the graph's kernel nodes carry StepParams by value
  { const int zero = 0; HIP_TRY(h, hipMemcpy(&h->st->ns_valid, &zero, sizeof(int), hipMemcpyHostToDevice)); }   // another filter kernel may run next: no carried register dump
  h->need_prep = true;
  return PSMF_OK;
}

int psmf_set_q_matrix_schedule(psmf_handle h, const double* Q_k, int64_t n) {
  if (!h || n < 0) return PSMF_ERR_ARG;
  if (Q_k && n > 0) {
    if (h->cfg.robust) return fail(h, PSMF_ERR_ARG, "psmf_set_q_matrix_schedule: rPSMF runs on its own scaled Q (rpsmf.py:123,128)");
    if (h->cfg.masked) return fail(h, PSMF_ERR_ARG, "psmf_set_q_matrix_schedule: masked handles take a constant Q (the ExperimentImpute filters)");
    if (h->engine != 1) return fail(h, PSMF_ERR_ARG, "psmf_set_q_matrix_schedule: a Q_k that is not a multiple of Q_1 needs the per-step engine (engine = 1)");
    if (h->cfg.dyn_kind == PSMF_DYN_HOST) return fail(h, PSMF_ERR_ARG, "psmf_set_q_matrix_schedule: host-stepped dynamics form P_bar (and add Q_k) on the host");
  }
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->qmat) { HIP_TRY(h, hipFree(h->qmat)); h->qmat = nullptr; }
  h->qmat_n = 0;
  h->sp.q_mat = nullptr;
  if (Q_k && n > 0) {
    // n + 1 matrices, the last one repeated (as in psmf_set_schedules: the serial stage prepares one step ahead)
    const size_t rr = (size_t)h->cfg.r * h->cfg.r;
    HIP_TRY(h, hipMalloc((void**)&h->qmat, (size_t)(n + 1) * rr * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->qmat, Q_k, (size_t)n * rr * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->qmat + (size_t)n * rr, Q_k + (size_t)(n - 1) * rr, rr * sizeof(double), hipMemcpyHostToDevice));
    h->qmat_n = n;
    h->sp.q_mat = h->qmat;
  }
  update_solve_dual(h);
  destroy_graph(h);      // the graph's kernel nodes carry StepParams by value
  { const int zero = 0; HIP_TRY(h, hipMemcpy(&h->st->ns_valid, &zero, sizeof(int), hipMemcpyHostToDevice)); }
  h->need_prep = true;
  return PSMF_OK;
}

int psmf_step_host(psmf_handle h, int64_t k, const double* mu_bar, const double* P_bar, double* mu_out, double* gf_out,
                   double* P_out, double* Q_out) {
  if (!h || !mu_bar || !P_bar) return PSMF_ERR_ARG;
  if (h->cfg.dyn_kind != PSMF_DYN_HOST) return fail(h, PSMF_ERR_STATE, "psmf_step_host: the handle was not created with dyn_kind = PSMF_DYN_HOST");
  if (!h->have_state) return fail(h, PSMF_ERR_STATE, "psmf_step_host: set_state (C, V, P, mu) first");
  if (!h->Y) return fail(h, PSMF_ERR_STATE, "psmf_step_host: upload_series first");
  if (k < 0 || k >= h->T_cap) return fail(h, PSMF_ERR_ARG, "psmf_step_host: step outside the uploaded series");
  if (h->sched && k + 1 >= h->sched_n) return fail(h, PSMF_ERR_ARG, "psmf_step_host: step beyond the R / Q schedules");
  int rc = set_device(h);
  if (rc) return rc;
  const int r = h->cfg.r;
  if (h->need_prep || h->k_done != k) {
    hipLaunchKernelGGL(psmf::psmf_prepare_k, dim3(1), dim3(1), 0, h->stream, h->st, (long long)k);
    if (h->mu_hist)
      HIP_TRY(h, hipMemcpyAsync(h->mu_hist + (size_t)(k - h->sp.series_t0) * r, h->st->mu, r * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (h->sp.track_g) { rc = enqueue_gram(h); if (rc) return rc; }
    h->need_prep = false;
    h->k_done = k;
  }
  HIP_TRY(h, hipMemcpyAsync(h->st->mu_bar, mu_bar, r * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->st->Pbar, P_bar, (size_t)r * r * sizeof(double), hipMemcpyHostToDevice, h->stream));
  launch_serial(h, 1);          // w, s, eta, N, kappa of this step from the uploaded mu_bar, P_bar
  rc = enqueue_step(h);         // row sweep (+ r x r solve), all-reduce, serial stage (stops before the next prediction)
  if (rc) return rc;
  HIP_TRY(h, hipGetLastError());
  h->k_done = k + 1;
  rc = psmf_sync(h);
  if (rc) return rc;
  if (mu_out) HIP_TRY(h, hipMemcpy(mu_out, h->st->mu, r * sizeof(double), hipMemcpyDeviceToHost));
  if (gf_out) HIP_TRY(h, hipMemcpy(gf_out, h->st->gf, r * sizeof(double), hipMemcpyDeviceToHost));
  if (P_out) HIP_TRY(h, hipMemcpy(P_out, h->st->P, (size_t)r * r * sizeof(double), hipMemcpyDeviceToHost));
  if (Q_out) HIP_TRY(h, hipMemcpy(Q_out, h->st->Q, (size_t)r * r * sizeof(double), hipMemcpyDeviceToHost));
  return PSMF_OK;
}

int psmf_comm_info(psmf_handle h, int32_t* out4) {
  if (!h || !out4) return PSMF_ERR_ARG;
  out4[0] = h->host_fn ? 2 : (h->comm ? 1 : 0);
  out4[1] = h->nranks; out4[2] = h->rank; out4[3] = h->cfg.device;
  if (h->comm) {          // what RCCL itself says about the communicator the exchanges run on
    int n = -1, rk = -1, dev = -1;
    NCCL_TRY(h, ncclCommCount(h->comm, &n));
    NCCL_TRY(h, ncclCommUserRank(h->comm, &rk));
    NCCL_TRY(h, ncclCommCuDevice(h->comm, &dev));
    out4[1] = n; out4[2] = rk; out4[3] = dev;
  }
  return PSMF_OK;
}

int psmf_device_pci_bus_id(int device, char* buf, int len) {
  if (!buf || len < 16) return fail(nullptr, PSMF_ERR_ARG, "psmf_device_pci_bus_id: buffer of at least 16 bytes");
  if (hipDeviceGetPCIBusId(buf, len, device) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, PSMF_ERR_HIP, "hipDeviceGetPCIBusId failed"); }
  return PSMF_OK;
}

/* ---- masked filter on the large-d handle (psmf_masked.hip) ------------------------------------------------------------- */
int psmf_upload_mask(psmf_handle h, const uint8_t* M, int64_t t0, int64_t nt) {
  if (!h || !M || t0 < 0 || nt < 0) return fail(h, PSMF_ERR_ARG, "psmf_upload_mask: bad argument");
  if (!h->cfg.masked) return fail(h, PSMF_ERR_STATE, "psmf_upload_mask: the handle was created with masked = 0");
  if (!h->mask || t0 + nt > h->T_cap) return fail(h, PSMF_ERR_STATE, "psmf_upload_mask: upload the series first (it sizes the mask buffer)");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(h->mask + (size_t)t0 * h->cfg.d_local, M, (size_t)nt * h->cfg.d_local, hipMemcpyHostToDevice));
  h->have_mask = true;
  return PSMF_OK;
}

int psmf_set_step_size(psmf_handle h, double gam) {
  if (!h || !(gam >= 0.0)) return fail(h, PSMF_ERR_ARG, "psmf_set_step_size: bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(&h->st->sgd_gamma, &gam, sizeof(double), hipMemcpyHostToDevice));
  return PSMF_OK;
}

int psmf_masked_metrics(psmf_handle h, const uint8_t* Mmiss, int64_t t0, int64_t nt, double sig, double* out4) {
  if (!h || !Mmiss || !out4 || t0 < 0 || nt < 1) return fail(h, PSMF_ERR_ARG, "psmf_masked_metrics: bad argument");
  if (!h->cfg.masked || !h->have_mask || !h->YP || t0 + nt > h->T_cap) return fail(h, PSMF_ERR_STATE, "psmf_masked_metrics: needs a masked handle that has run over these steps");
  int rc = psmf_sync(h);
  if (rc) return rc;
  const size_t dl = h->cfg.d_local, nb = (size_t)nt * dl;
  if (h->mmiss_cap < nb) {
    if (h->mmiss) HIP_TRY(h, hipFree(h->mmiss));
    h->mmiss = nullptr; h->mmiss_cap = 0;
    HIP_TRY(h, hipMalloc((void**)&h->mmiss, nb));
    h->mmiss_cap = nb;
  }
  HIP_TRY(h, hipMemcpy(h->mmiss, Mmiss, nb, hipMemcpyHostToDevice));
  const int gx = (int)((dl + psmf::WG - 1) / psmf::WG);
  int gy = (int)((2048 + gx - 1) / gx);                  // ~2 k workgroups in all
  if (gy > nt) gy = (int)nt;
  if (gy < 1) gy = 1;
  const int chunk = (int)((nt + gy - 1) / gy);
  gy = (int)((nt + chunk - 1) / chunk);
  rc = ensure_scratch(h, (size_t)gx * gy * 4 * sizeof(double));
  if (rc) return rc;
  const size_t lds = (size_t)32 * h->cfg.r * sizeof(double);
  if (h->cfg.storage == PSMF_F64)
    hipLaunchKernelGGL(psmf::psmf_masked_metrics_k<double>, dim3(gx, gy), dim3(psmf::WG), lds, h->stream, h->sp, (const uint8_t*)h->mask,
                       (const uint8_t*)h->mmiss, (const double*)h->sc_hist, (long long)t0, (int)nt, chunk, sig, h->cfg.robust, h->scratch);
  else
    hipLaunchKernelGGL(psmf::psmf_masked_metrics_k<float>, dim3(gx, gy), dim3(psmf::WG), lds, h->stream, h->sp, (const uint8_t*)h->mask,
                       (const uint8_t*)h->mmiss, (const double*)h->sc_hist, (long long)t0, (int)nt, chunk, sig, h->cfg.robust, h->scratch);
  HIP_TRY(h, hipGetLastError());
  std::vector<double> part((size_t)gx * gy * 4);
  HIP_TRY(h, hipMemcpyAsync(part.data(), h->scratch, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, spin_stream(h->stream));
  for (int q = 0; q < 4; ++q) out4[q] = 0.0;
  for (size_t b = 0; b < (size_t)gx * gy; ++b)
    for (int q = 0; q < 4; ++q) out4[q] += part[b * 4 + q];       // fixed order
  return PSMF_OK;
}

int psmf_download_step_scalars(psmf_handle h, double* out, int64_t t0, int64_t nt) {
  if (!h || !out || t0 < 0 || nt < 0) return fail(h, PSMF_ERR_ARG, "psmf_download_step_scalars: bad argument");
  if (!h->cfg.masked || !h->sc_hist || t0 + nt > h->T_cap) return fail(h, PSMF_ERR_STATE, "psmf_download_step_scalars: needs a masked handle with an uploaded series");
  int rc = psmf_sync(h);
  if (rc) return rc;
  HIP_TRY(h, hipMemcpy(out, h->sc_hist + 2 * (size_t)t0, (size_t)nt * 2 * sizeof(double), hipMemcpyDeviceToHost));
  return PSMF_OK;
}

int psmf_measure_copy_bandwidth(int device, size_t bytes, int iters, double* gbps) {
  if (!gbps || iters < 1 || bytes < (size_t)1 << 20) return fail(nullptr, PSMF_ERR_ARG, "psmf_measure_copy_bandwidth: bad argument");
  psmf_handle h = nullptr;       // errors go to the create-error slot
  HIP_TRY(h, hipSetDevice(device));
  const size_t n16 = bytes / 16;
  void *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t s = nullptr;
  HIP_TRY(h, hipMalloc(&src, n16 * 16));
  HIP_TRY(h, hipMalloc(&dst, n16 * 16));
  HIP_TRY(h, hipMemset(src, 1, n16 * 16));
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  HIP_TRY(h, hipEventCreate(&e0));
  HIP_TRY(h, hipEventCreate(&e1));
  // four 16-byte vectors per thread (measured on MI355X, 1 GiB: 4.1 TB/s with 4 workgroups per CU, 5.0 with one workgroup
  // per 16 KiB; 4 GiB: 5.5 TB/s)
  size_t gsz = n16 / ((size_t)psmf::WG * 4);
  if (gsz < 1024) gsz = 1024;
  if (gsz > ((size_t)1 << 20)) gsz = (size_t)1 << 20;
  const int grid = getenv("PSMF_COPY_GRID") ? atoi(getenv("PSMF_COPY_GRID")) : (int)gsz;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(psmf::psmf_copy_k, dim3(grid), dim3(psmf::WG), 0, s, (const float4*)src, (float4*)dst, n16);
  HIP_TRY(h, hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(psmf::psmf_copy_k, dim3(grid), dim3(psmf::WG), 0, s, (const float4*)src, (float4*)dst, n16);
  HIP_TRY(h, hipEventRecord(e1, s));
  HIP_TRY(h, hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
  *gbps = 2.0 * (double)(n16 * 16) * iters / (ms * 1e-3) / 1e9;     // read + write
  hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s); hipFree(src); hipFree(dst);
  return PSMF_OK;
}

int psmf_comm_init_host(psmf_handle h, int nranks, int rank, psmf_allreduce_fn fn, void* ctx) {
  if (!h || !fn || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, PSMF_ERR_ARG, "psmf_comm_init_host: bad argument");
  if (h->rotU) return fail(h, PSMF_ERR_STATE, "psmf_comm_init_host: a handle with a noise rotation (non-diagonal R) is one shard by construction");
  if (h->comm) return fail(h, PSMF_ERR_STATE, "psmf_comm_init_host: the handle already has an RCCL communicator");
  int rc = set_device(h);
  if (rc) return rc;
  if (!h->host_buf) HIP_TRY(h, hipHostMalloc((void**)&h->host_buf, psmf_filter::kHostBufElems * sizeof(double), hipHostMallocDefault));
  h->host_fn = fn;
  h->host_ctx = ctx;
  h->nranks = nranks;
  h->rank = rank;
  h->use_coll = true;               // also with one rank: the exchange path is what this communicator exists to exercise
  h->sp.external_reduce = 1;
  destroy_graph(h);
  h->need_prep = true;
  return PSMF_OK;
}

}  // extern "C"

#include "psmf_impute.hip"

// psmf_impute_run beyond one workgroup's LDS (d > 512 or r > 16): the replicas one after the other on the masked per-step engine
// of the large-d handle (psmf_masked.hip), float64 storage.  ExperimentImpute/PSMF.py:59-95, rPSMF.py:75-148: the prior mean of
// column 0 is X[:, n - 1] (the initial X on pass 0, the last posterior afterwards -- i.e. the running mean), C, V, P carry over the
// passes, rPSMF restarts Q, rho, lambda at every pass; X[:, t] is the posterior mean of column t (the mean history).
int impute_run_large(const psmf_impute_config* cfg, const double* YorgInt, const uint8_t* M, const uint8_t* Mmiss, double* C, double* X,
                     const double* V, const double* P, const double* Q, double rho, double* Epred, double* Efull, double* inside,
                     double* Yrec, double* YrecL, double* YrecH, int32_t* status, float* elapsed_ms) {
  auto failc = [&](int code, const std::string& msg) { g_create_error = "psmf_impute_run: " + msg; return code; };
  (void)failc;
  const int d = cfg->d, n = cfg->n, r = cfg->r, B = cfg->batch, robust = cfg->method == 1, meth = cfg->method;
  psmf_config pc;
  memset(&pc, 0, sizeof(pc));
  pc.abi_version = PSMF_ABI_VERSION; pc.d = d; pc.r = r; pc.row0 = 0; pc.d_local = d; pc.robust = robust;
  pc.coef_update = 1; pc.eta_full = 1; pc.pbar_predict = 1; pc.dyn_kind = PSMF_DYN_RANDOM_WALK; pc.n_theta = 0;
  pc.storage = PSMF_F64; pc.store_y_pred = 1; pc.update_every = 1; pc.device = cfg->device; pc.use_graph = 1; pc.engine = 1; pc.masked = meth >= 2 ? meth : 1;
  pc.alpha = pc.beta = 1.0; pc.adam_lr = 1e-3; pc.adam_b1 = 0.9; pc.adam_b2 = 0.999;
  psmf_handle h = nullptr;
  int rc = psmf_create(&h, &pc);
  if (rc) return rc;               // (g_create_error holds the message)
  auto bail = [&](int code) { g_create_error = std::string("psmf_impute_run: ") + psmf_last_error(h); psmf_destroy(h); return code; };
  rc = psmf_upload_series(h, YorgInt, PSMF_F64, 0, n, n);
  if (rc) return bail(rc);
  const size_t nd = (size_t)n * d;
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  std::vector<double> yp, sc;
  double total_ms = 0.0;
  for (int b = 0; b < B; ++b) {
    double* Cb = C + (size_t)b * d * r;
    double* Xb = X + (size_t)b * n * r;
    rc = psmf_upload_mask(h, M + (size_t)b * nd, 0, n);
    if (rc) return bail(rc);
    if (meth == 3) {       // TMF (TMF.py:47,60): Pbar = I / nu at every step, nu = 2 -- as Q with P = 0 (the serial stage keeps P at 0); V unused
      std::vector<double> Inu((size_t)r * r, 0.0), Zr((size_t)r * r, 0.0), Iv((size_t)r * r, 0.0);
      for (int i = 0; i < r; ++i) { Inu[(size_t)i * r + i] = 0.5; Iv[(size_t)i * r + i] = 1.0; }
      rc = psmf_set_state(h, Cb, Iv.data(), Zr.data(), Inu.data(), Xb + (size_t)(n - 1) * r, 1.0, 0.0, nullptr);
    } else {
      rc = psmf_set_state(h, Cb, V, P, Q, Xb + (size_t)(n - 1) * r, rho, robust ? cfg->lambda0 : 0.0, nullptr);
    }
    if (rc) return bail(rc);
    bool bad = false;
    double m4[4] = {0, 0, 0, 0};
    const auto t_start = std::chrono::steady_clock::now();
    for (int it = 0; it < cfg->n_iter && !bad; ++it) {
      if (it > 0 && robust) {        // rPSMF.py:77-79: Q, R, lambda restart; V, P, C and the mean carry over
        rc = psmf_set_state(h, nullptr, nullptr, nullptr, Q, nullptr, rho, cfg->lambda0, nullptr);
        if (rc) return bail(rc);
      }
      if (meth >= 2) {       // gam = 1e-6 / (pass + 1)^0.7  (MLESMF.py:59-60, TMF.py:46-48)
        rc = psmf_set_step_size(h, 1e-6 / std::pow((double)(it + 1), 0.7));
        if (rc) return bail(rc);
      }
      rc = psmf_run(h, 0, n);
      if (rc) return bail(rc);
      rc = psmf_masked_metrics(h, Mmiss + (size_t)b * nd, 0, n, meth == 3 ? 0.0 : cfg->sig, m4);
      if (rc == PSMF_ERR_NUMERIC) { bad = true; break; }
      if (rc) return bail(rc);
      Epred[(size_t)b * cfg->n_iter + it] = std::sqrt(m4[0] / m4[3]);
      Efull[(size_t)b * cfg->n_iter + it] = std::sqrt(m4[1] / m4[3]);
      if (!std::isfinite(m4[0]) || !std::isfinite(m4[1])) bad = true;
    }
    total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    if (!bad) {
      inside[b] = m4[2] / m4[3];
      rc = psmf_get_state(h, Cb, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      if (!rc) rc = psmf_download_mu(h, Xb, 1, n);
      if (!rc && cfg->want_bands) {
        yp.resize(nd); sc.resize((size_t)2 * n);
        rc = psmf_download_y_pred(h, yp.data(), PSMF_F64, 0, n);
        if (!rc) rc = psmf_download_step_scalars(h, sc.data(), 0, n);
        if (!rc) {
          const uint8_t* Mb = M + (size_t)b * nd;
          for (int t = 0; t < n; ++t)
            for (int i = 0; i < d; ++i) {
              const size_t at = (size_t)b * nd + (size_t)t * d + i;
              const double yh = yp[(size_t)t * d + i];
              const double band = meth == 3 ? 0.0 : cfg->sig * std::sqrt(robust ? (Mb[(size_t)t * d + i] ? sc[2 * t] : 0.0) + sc[2 * t + 1] : sc[2 * t] + sc[2 * t + 1]);
              Yrec[at] = yh; YrecL[at] = yh - band; YrecH[at] = yh + band;
            }
        }
      }
      if (rc == PSMF_ERR_NUMERIC) bad = true;
      else if (rc) return bail(rc);
    }
    if (status) status[b] = bad ? PSMF_ERR_NUMERIC : PSMF_OK;
    if (bad) {
      if (!status) return bail(PSMF_ERR_NUMERIC);
      for (int it = 0; it < cfg->n_iter; ++it) Epred[(size_t)b * cfg->n_iter + it] = Efull[(size_t)b * cfg->n_iter + it] = qnan;
      inside[b] = qnan;
    }
  }
  if (elapsed_ms) *elapsed_ms = (float)total_ms;
  psmf_destroy(h);
  return PSMF_OK;
}
