// Persistent per-step engine (psmf_pstep.hip): interface between its translation unit and the C-ABI host code.
#pragma once
#include "psmf_device.h"

namespace psmf {

constexpr int PSTEP_NPMAX = 12;      // row passes a row workgroup can keep in registers (4 float64 per lane and pass; y_k beside them)
constexpr int PSTEP_PKT_MAX = 4 * RM + 1;      // granules of the hub -> rows packet
// Replicas of words that MANY workgroups poll at once, 4 KB apart; workgroup i polls replica i % R, the writers store all replicas.
// Measured (d = 1e5, r = 32): eight replicas of the hub's packet COST 0.9 us per timestep (24 granule stores per lane of the publishing
// wave, on the critical path) and bought nothing -- the packet has one; eight replicas of the masked handles' Gram-partial flags (every
// row workgroup polls every flag; eight 4-byte stores per workgroup) are kept.
constexpr int PSTEP_PKT_REP = 1;
constexpr int PSTEP_GF_REP = 8;
constexpr int PSTEP_REP_STRIDE = 4096;         // bytes between replicas
constexpr int PSTEP_FANIN_ROWS = 17;           // partial rows one thread of the hub's fan-in sums (all its loads in flight at once)
constexpr int PSTEP_FANIN_ROWS_BIG = 20;       // ... at 33 <= r <= 48 (the workers of that hub hold nothing in registers): d_local = 1e5 at r = 48 is 196 rows of 10 segments

struct PstepParams {
  StepParams sp;               // the handle's parameter block
  long long k_begin;           // series index of the first step of the launch (== st->k)
  int n_steps;
  int n_row_wg;                // row workgroups; grid = n_row_wg + 1 (block 0 = hub)
  int rows_per_wg;
  int np;                      // row passes per workgroup (<= PSTEP_NPMAX)
  int ncol2;                   // doubles per partial row (even, >= r + 1)
  unsigned* flags;             // rows -> hub: one epoch word per row workgroup        } one block, zeroed before every launch
  unsigned long long* pkt;     // hub -> rows: PSTEP_PKT_MAX {tag, value} granules      }
  double* part;                // rows -> hub: n_row_wg x ncol2 partial sums (write-through stores)
  // masked handles (cfg.masked = 1, PSMF / rPSMF): the masked Gram of the NEXT step travels rows -> rows -> hub every timestep
  int masked;                  // 1: sp.mask is the observation mask; G_m = sum_i m_i c_i c_i^T is formed per step from the on-chip C
  int nge;                     // elements of a Gram partial: upper 16 x 16 tiles in the MFMA layout (256 each) + the observed count
  int slice_len;               // elements of the reduced Gram one row workgroup sums over all partials (ceil(nge / n_row_wg))
  unsigned* gflags;            // rows -> rows: partial Gram of the epoch published       } in the zeroed block
  unsigned* sflags;            // (spare)                                                 }
  double* gpart;               // n_row_wg x nge
  double* gslice;              // rows -> hub: the reduced Gram as 2 nge {tag, half} granules (element e at granules 2 e, 2 e + 1); in the zeroed block
  double* mg_out;              // the handle's reduced-Gram buffer (r*r + 1 | trace shares): what a launch leaves for the next one
  int mg_ntr;
  long long* prof;             // diagnostic builds (-DPSTEP_PROF): per-phase shader-clock sums, [0..15] hub workers, [16..23] solve wave, [24..31] row workgroup 1
};

struct PstepPlan {
  int n_row_wg, rows_per_wg, np, ncol2;
  size_t zero_bytes;           // flags + packet: the block a launch zeroes (starts the allocation, multiple of 16 bytes)
  size_t off_pkt, off_part, total_bytes;
  int nge, slice_len;          // masked handles (0 otherwise)
  size_t off_gflags, off_sflags, off_gpart, off_gslice;
};

// geometry of a launch for d_local rows at rank r on a device with n_cu compute units; false: the shape does not fit the kernel
bool pstep_plan(int d_local, int r, int n_cu, bool storage_f64, bool masked, PstepPlan* out);
// one launch = n_steps timesteps (the communication block must have been zeroed on the same stream)
hipError_t pstep_launch(const PstepParams& q, bool storage_f64, hipStream_t stream);
// one-off per process: dynamic-LDS attribute of every instance
hipError_t pstep_init();

}  // namespace psmf
