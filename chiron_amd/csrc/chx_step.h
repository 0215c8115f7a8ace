// Launch-side view of the fused step kernels: the argument block the host loop (chx_launch_window.hip) fills and the
// kernel translation unit (chx_lj.hip) consumes.  Internal, not exported.
#pragma once
#include "chx_common.h"
#include "chx_internal.h"

struct LJParams {
  float T;      // r2 <= T  <=>  d < cutoff
  float sig2;   // sigma^2
  float eps4;   // 4 eps
  float eps24;  // 24 eps
};

LJParams chx_make_lj(float sigma, float epsilon, float cutoff);

struct StepArgs {
  int n;
  const float4* x_cur;
  float* x_next;  // (n,4)
  float* v;       // (n,3)
  float4* ref4;   // written only at a reference rebuild event of the deferred scheme (snap4 set)
  const float* mass;
  const uint32_t* perm;
  const uint32_t* fl;
  const int32_t* deg;
  int chunks;
  Box3 box;
  LJParams lj;
  int wrap;
  float T_half_skin;
  float hdt, a, b, kT;
  uint32_t k0, k1;
  const float* noise;
  long long noise_step0;
  int32_t* ctrl;
  // dynamically pruned inner list (NULL: always run on the full force list)
  uint32_t* fli;
  int32_t* degi;
  int32_t* host_ctrl;   // host-mapped copy of the control block (chx_langevin_t::host_ctrl), or NULL
  float T_inner;        // r2 <= T_inner: kept by the pruning pass  ((cutoff + inner skin)^2, padded up)
  float half_inner;     // a particle whose path since the pass is longer than this invalidates the inner list
  // deferred rebuilds (chx_langevin_t::snap4; nullptr = the halting scheme)
  const float4* snap4;  // positions the force list was built from
  int32_t* ev_log;
  int ev_cap;
  // the run's last (force-only) launch also writes the state in user order (chx_langevin_t::final_x3 ...); NULL = off
  float* x3_out;
  float* v3_out;
  float* ref3_out;
};

// One fused [pair force + BAOAB + wrap + skin check] launch on the force list with g lanes per particle (k_lj_step);
// prune / epoch: the launch's place in the refresh cadence of the pruned inner list (chx_launch_lj_steps works them out).
int chx_launch_step(const StepArgs& a, int g, long long step, int n_kicks, int do_aoa, bool prune, int epoch, hipStream_t st);
// The same without a neighbor list: all-pairs force + BAOAB (k_ap_step).
void chx_launch_ap_step(const StepArgs& a, long long step, int n_kicks, int do_aoa, int use_cutoff, hipStream_t st);

// ---- several replicas in one launch (k_lj_step_batch; include/chx.h: chx_langevin_batch_run) ----
#define CHX_BATCH_MAX 8   // items per packet; more replicas than that go out as several packets side by side
struct BatchItem {
  StepArgs a;
  long long step;
  const uint32_t* sfl;   // the list this replica's launch most probably walks (lj_step_body's speculative loads)
  const int32_t* sdeg;
  unsigned nblocks;
  int epoch;
  int n_kicks;
  int do_aoa;
  int prune;             // this replica's launch refreshes its inner list (k_lj_step's PRUNE instantiation)
  int pad;
};
struct BatchArgs {
  BatchItem item[CHX_BATCH_MAX];
};
// One packet: `count` items of ONE class (all steady-state launches or none, the same lanes per particle g).
// `beside_previous`: the packet may run beside the one dispatched just before it (the other classes of the same step); the
// first packet of a step waits for everything before it.
int chx_launch_step_batch(const BatchItem* items, int count, int g, bool hot, bool beside_previous, hipStream_t st);
