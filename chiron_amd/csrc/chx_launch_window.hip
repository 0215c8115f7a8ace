// libchx: the host side of the fused Langevin loop -- the bounded launch queue ("launch window") that keeps a few
// step launches queued ahead of the device, polls the host-mapped copy of the control block and stops at the halt word
// (include/chx.h: chx_langevin_fused_run).  No kernels here: they are in chx_lj.hip, reached through chx_step.h.
//
// Side effect on the calling thread (documented in include/chx.h): with the sleeping poll policy the thread's timer
// slack is lowered to 1 us once (prctl(PR_SET_TIMERSLACK)), so that a 5 us nap is not rounded up to the kernel's
// default 50 us.  CHX_POLL_TIMERSLACK=0 leaves the slack alone.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>

#include "chx_step.h"

// ---------------------------------------------------------------------------
// How the launch window waits for the device when its queue is full.
//   sleep (default): nanosleep between polls, with the thread's timer slack lowered once so that a 5 us nap is
//                    not rounded up to the kernel's default 50 us; the queue refills in bursts (see the loop).
//   spin  (CHX_POLL=spin): busy-wait with the architecture's relax hint -- one core at 100 % per rank.
// Eight ranks on one host share its CPU quota with eight HIP helper threads.
// ---------------------------------------------------------------------------
struct PollPolicy {
  long sleep_ns;  // 0: spin
};
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  sched_yield();
#endif
}
// (run-time switchable: chx_set_poll_sleep_ns -- the launch-regime probe of chiron_amd/_device.py spins for its
// measurement, where a voluntary context switch of the launching thread can only be a collision with the runtime)
// diagnostics of the sleeping window (chx_poll_stats): waits, naps, wake-ups that found the queue (nearly) empty, ns asleep
static int64_t g_poll_stats[4] = {0, 0, 0, 0};
extern "C" int chx_poll_stats(int64_t* out4_h, int32_t reset) {
  if (out4_h)
    for (int i = 0; i < 4; ++i) out4_h[i] = __atomic_load_n(&g_poll_stats[i], __ATOMIC_RELAXED);
  if (reset)
    for (int i = 0; i < 4; ++i) __atomic_store_n(&g_poll_stats[i], 0, __ATOMIC_RELAXED);
  return CHX_OK;
}
static long g_poll_sleep_ns = [] {
  const char* e = getenv("CHX_POLL");
  if (e && (e[0] == 's' && e[1] == 'p')) return 0L;  // "spin"
  const char* n = getenv("CHX_POLL_SLEEP_NS");
  long v = n ? atol(n) : 5000L;
  return v < 0 ? 0L : v;
}();
extern "C" int64_t chx_set_poll_sleep_ns(int64_t ns) {
  const long old = __atomic_exchange_n(&g_poll_sleep_ns, (long)(ns < 0 ? 0 : ns), __ATOMIC_RELAXED);
  return (int64_t)old;
}
static PollPolicy poll_policy() {
  const long ns = __atomic_load_n(&g_poll_sleep_ns, __ATOMIC_RELAXED);
  if (ns > 0) {
    static thread_local bool slack_set = false;
    if (!slack_set) {
      static const bool leave = [] { const char* e = getenv("CHX_POLL_TIMERSLACK"); return e && e[0] == '0'; }();
      if (!leave) (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);  // this thread only; 1 us
      slack_set = true;
    }
  }
  return PollPolicy{ns};
}

// the launch-invariant part of a replica's argument block (x_cur / x_next are set per launch)
static StepArgs make_step_args(const chx_langevin_t* p, int do_aoa, bool& use_inner) {
  StepArgs a;
  a.n = p->n;
  a.v = p->v;
  a.ref4 = reinterpret_cast<float4*>(p->ref4);
  a.mass = p->mass;
  a.perm = p->perm;
  a.fl = p->fl;
  a.deg = p->deg;
  a.chunks = p->fl_chunks;
  a.box = make_box3(p->box);
  a.lj = chx_make_lj(p->sigma, p->epsilon, p->cutoff);
  a.wrap = p->wrap;
  a.T_half_skin = chx_cutoff_threshold_h(p->skin * 0.5f);
  a.hdt = p->half_dt;
  a.a = p->a;
  a.b = p->b;
  a.kT = p->kT;
  a.k0 = p->k0;
  a.k1 = p->k1;
  a.noise = p->noise;
  a.noise_step0 = p->noise_step0;
  a.ctrl = p->ctrl;
  a.host_ctrl = p->fl != nullptr ? p->host_ctrl : nullptr;
  use_inner = p->fl_inner && p->deg_inner && p->prune_every > 0 && p->inner_skin > 0.f;  // (ref_inner4: unused)
  a.fli = use_inner ? p->fl_inner : nullptr;
  a.degi = use_inner ? p->deg_inner : nullptr;
  {
    // keep |r| < cutoff + inner_skin (padded up), distrust the table once a particle has moved
    // inner_skin / 2 (padded down): the paddings dwarf fp32 rounding
    const float rk = (p->cutoff + p->inner_skin) * 1.00001f;
    const float hm = 0.5f * p->inner_skin * 0.9999f;
    a.T_inner = rk * rk;
    a.half_inner = hm;
  }
  // deferred rebuilds need both position sets and the force-list path
  const bool deferred = p->snap4 != nullptr && p->fl != nullptr && p->ref4 != nullptr;
  a.snap4 = deferred ? reinterpret_cast<const float4*>(p->snap4) : nullptr;
  a.ev_log = deferred ? p->ev_log : nullptr;
  a.ev_cap = deferred ? p->ev_log_cap : 0;
  // (only the force-only launch that ends a run writes the user-order copies)
  a.x3_out = do_aoa ? nullptr : p->final_x3;
  a.v3_out = do_aoa ? nullptr : p->final_v3;
  a.ref3_out = do_aoa ? nullptr : p->final_ref3;
  return a;
}

// launch i of the chunk reads (i even ? x_even : x_odd) and writes the other buffer
// window > 0 (needs p->host_ctrl, readable by the host at the same address): launch step s only once fewer than
// `window` earlier launches are still waiting to begin, and stop at the first step the halt word rules out;
// *n_enqueued = launches made.  What the host polls is the copy of the control block every launch stores into
// host_ctrl when it begins (word 0 = its step index, word 3 = the halt word).
int chx_launch_lj_steps(const chx_langevin_t* p, float* x_even, float* x_odd, int64_t step0,
                        int n_steps, int kicks_first, int do_aoa, hipStream_t st, int window, int32_t* n_enqueued) {
  if (n_enqueued) *n_enqueued = 0;
  bool use_inner = false;
  StepArgs a = make_step_args(p, do_aoa, use_inner);
  const bool throttle = window > 0 && a.host_ctrl != nullptr;
  const PollPolicy poll = poll_policy();
  for (int i = 0; i < n_steps; ++i) {
    if (throttle) {
      const long long s = step0 + i;
      const volatile int32_t* hc = a.host_ctrl;
      if ((long long)hc[3] <= s) return CHX_OK;  // the skin check has fired: launches from s on would be no-ops
      if (s - hc[0] > window) {
        // The queue is full.  Spinning here keeps one core at 100 % per rank for nothing: the device needs
        // ~5 us per queued launch.  The sleeping policy therefore lets the queue drain to `low` launches before
        // it refills it in one burst (the host enqueues a launch in ~2.5 us, so the burst outruns the device),
        // sleeping in between; the spinning policy (CHX_POLL=spin) refills launch by launch.
        const int low = poll.sleep_ns > 0 ? window - window / 2 : window;
        int last = hc[0];
        bool timing = false;
        timespec t0{};
        if (poll.sleep_ns > 0) __atomic_fetch_add(&g_poll_stats[0], 1, __ATOMIC_RELAXED);
        for (unsigned spins = 0;; ++spins) {
          if ((long long)hc[3] <= s) return CHX_OK;
          const int begun = hc[0];  // newest launch that has begun; begun+1 .. s-1 are still waiting
          if (s - begun <= low) {
            if (poll.sleep_ns > 0 && s - begun <= 2) __atomic_fetch_add(&g_poll_stats[2], 1, __ATOMIC_RELAXED);  // overslept: the queue is all but dry
            break;
          }
          if (poll.sleep_ns > 0) {
            const timespec nap = {0, poll.sleep_ns};
            timespec n0, n1;
            clock_gettime(CLOCK_MONOTONIC, &n0);
            nanosleep(&nap, nullptr);
            clock_gettime(CLOCK_MONOTONIC, &n1);
            __atomic_fetch_add(&g_poll_stats[1], 1, __ATOMIC_RELAXED);  // (several launching threads may share the counters)
            __atomic_fetch_add(&g_poll_stats[3], (n1.tv_sec - n0.tv_sec) * 1000000000LL + (n1.tv_nsec - n0.tv_nsec), __ATOMIC_RELAXED);
          } else {
            cpu_relax();
          }
          if (poll.sleep_ns > 0 || (spins & 0xFFFF) == 0xFFFF) {  // a stream that makes no progress for a minute is not coming back
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (begun != last || !timing) last = begun, t0 = t1, timing = true;
            else if (t1.tv_sec - t0.tv_sec > 60) {
              chx_set_error("chx_langevin_fused_run: no step launch has begun for 60 s (step %lld, newest begun %d)", s, begun);
              return CHX_ERR_LAUNCH;
            }
          }
        }
      }
    }
    if (n_enqueued) *n_enqueued = i + 1;
    a.x_cur = reinterpret_cast<const float4*>((i & 1) ? x_odd : x_even);
    a.x_next = (i & 1) ? x_even : x_odd;
    const int nk = i == 0 ? kicks_first : 2;
    if (p->fl == nullptr) {  // no neighbor list: all-pairs force + BAOAB in one launch
      const int use_cutoff = p->cutoff >= 0.0f ? 1 : 0;
      if (!use_cutoff) a.lj = chx_make_lj(p->sigma, p->epsilon, 1.0f);
      a.ref4 = nullptr;  // no skin check on these paths
      a.snap4 = nullptr;
      chx_launch_ap_step(a, (long long)(step0 + i), nk, do_aoa, use_cutoff, st);
      continue;
    }
    // refreshes of the inner list sit at fixed step indices (counted from prune_phase), not at chunk
    // boundaries: a trajectory does not depend on how its steps are batched into calls
    // (the force-only launch that ends a run is included: it may be the first one after a rebuild)
    const bool since = use_inner && (step0 + i >= p->prune_phase);
    const bool prune = since && ((step0 + i - p->prune_phase) % p->prune_every == 0);
    const int epoch = since ? (int)((step0 + i - p->prune_phase) / p->prune_every) : -1;
    const int rc = chx_launch_step(a, p->g, step0 + i, nk, do_aoa, prune, epoch, st);
    if (rc != CHX_OK) return rc;
  }
  return CHX_OK;
}



// ---------------------------------------------------------------------------
// Several replicas, one launch per step (include/chx.h: chx_langevin_batch_run / chx_langevin_batch_kick).
// Launch i of the call advances every replica r with i < n_steps[r] by one step -- replica r's step step0[r] + i, from
// (i even ? x_even[r] : x_odd[r]) -- in ONE dispatch (k_lj_step_batch; one packet per class of launch when the replicas'
// launches differ in kind, side by side).  The window is counted on the first participating replica's control block (all
// items of a launch begin together); the halt words of ALL participants are watched, and the call returns at the first
// one seen: that replica's list has to be rebuilt behind the no-ops it has queued, the others simply continue with the
// next call.
// ---------------------------------------------------------------------------
int chx_launch_lj_steps_batch(chx_batch_item_t* items, int n_items, int do_aoa, int window, hipStream_t st) {
  constexpr int MAXR = 64;
  if (n_items < 1 || n_items > MAXR) {
    chx_set_error("chx_langevin_batch: %d replicas (1 .. %d)", n_items, MAXR);
    return CHX_ERR_INVALID;
  }
  StepArgs args[MAXR];
  bool inner[MAXR];
  int max_steps = 0;
  for (int r = 0; r < n_items; ++r) {
    args[r] = make_step_args(items[r].p, do_aoa, inner[r]);
    items[r].n_enqueued = 0;
    max_steps = items[r].n_steps > max_steps ? items[r].n_steps : max_steps;
  }
  const bool throttle = do_aoa && window > 0;
  const PollPolicy poll = poll_policy();
  auto active = [&](int r, int i) { return i < items[r].n_steps; };  // does item r take part in launch i?
  auto halted = [&](int i) {  // has some participant of launch i been ruled out by its halt word?
    for (int r = 0; r < n_items; ++r) {
      if (!active(r, i)) continue;
      const volatile int32_t* hc = args[r].host_ctrl;
      if ((long long)hc[3] <= items[r].step0 + i) return true;
    }
    return false;
  };
  for (int i = 0; i < max_steps; ++i) {
    if (throttle) {
      if (halted(i)) return CHX_OK;
      int lead = -1;
      for (int r = 0; r < n_items && lead < 0; ++r)
        if (active(r, i)) lead = r;
      const volatile int32_t* hc = args[lead].host_ctrl;
      const long long s = items[lead].step0 + i;
      if (s - hc[0] > window) {
        const int low = poll.sleep_ns > 0 ? window - window / 2 : window;
        int last = hc[0];
        bool timing = false;
        timespec t0{};
        if (poll.sleep_ns > 0) __atomic_fetch_add(&g_poll_stats[0], 1, __ATOMIC_RELAXED);
        for (unsigned spins = 0;; ++spins) {
          if (halted(i)) return CHX_OK;
          const int begun = hc[0];
          if (s - begun <= low) break;
          if (poll.sleep_ns > 0) {
            const timespec nap = {0, poll.sleep_ns};
            nanosleep(&nap, nullptr);
            __atomic_fetch_add(&g_poll_stats[1], 1, __ATOMIC_RELAXED);
          } else {
            cpu_relax();
          }
          if (poll.sleep_ns > 0 || (spins & 0xFFFF) == 0xFFFF) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (begun != last || !timing) last = begun, t0 = t1, timing = true;
            else if (t1.tv_sec - t0.tv_sec > 60) {
              chx_set_error("chx_langevin_batch_run: no step launch has begun for 60 s (step %lld, newest begun %d)", s, begun);
              return CHX_ERR_LAUNCH;
            }
          }
        }
      }
    }
    // this launch's items, sorted into classes (lanes per particle, steady-state launch or not): one packet per class (and
    // per CHX_BATCH_MAX items), the smallest first.  Whether an item refreshes its inner list is a flag of the item, not a
    // class: as packets of their own the refreshes (one replica in eight every fourth step) cost 2 us per step, and
    // neither order nor a missing barrier bit made consecutive packets of the queue overlap (measured, round 6).
    struct Class {
      int g, count;
      bool hot;
      int member[MAXR];
    };
    static thread_local Class cls[MAXR];
    int ncls = 0;
    int epoch_of[MAXR], nk_of[MAXR];
    bool prune_of[MAXR];
    for (int r = 0; r < n_items; ++r) {
      if (!active(r, i)) continue;
      const chx_langevin_t* p = items[r].p;
      const long long step = items[r].step0 + i;
      const int nk = i == 0 ? items[r].kicks_first : 2;
      const bool since = inner[r] && (step >= p->prune_phase);
      const bool prune = since && ((step - p->prune_phase) % p->prune_every == 0);
      epoch_of[r] = since ? (int)((step - p->prune_phase) / p->prune_every) : -1;
      nk_of[r] = nk;
      prune_of[r] = prune;
      const StepArgs& a = args[r];
      const bool hot = nk == 2 && do_aoa && a.wrap && a.box.periodic && a.ref4 && a.fli && a.perm && !a.noise;
      int c = 0;
      while (c < ncls && !(cls[c].g == p->g && cls[c].hot == hot && cls[c].count < CHX_BATCH_MAX)) ++c;
      if (c == ncls) cls[ncls++] = Class{p->g, 0, hot, {}};
      cls[c].member[cls[c].count++] = r;
    }
    int order[MAXR];
    for (int c = 0; c < ncls; ++c) order[c] = c;
    for (int x = 1; x < ncls; ++x)  // (insertion sort by item count, ascending; a handful of classes at most)
      for (int y = x; y > 0 && cls[order[y]].count < cls[order[y - 1]].count; --y) {
        const int t = order[y];
        order[y] = order[y - 1], order[y - 1] = t;
      }
    for (int oc = 0; oc < ncls; ++oc) {
      const Class& C = cls[order[oc]];
      BatchItem pack[CHX_BATCH_MAX];
      for (int m = 0; m < C.count; ++m) {
        const int r = C.member[m];
        const chx_langevin_t* p = items[r].p;
        StepArgs& a = args[r];
        a.x_cur = reinterpret_cast<const float4*>((i & 1) ? items[r].x_odd : items[r].x_even);
        a.x_next = (i & 1) ? items[r].x_even : items[r].x_odd;
        BatchItem& it = pack[m];
        it.a = a;
        it.step = items[r].step0 + i;
        const bool spec_inner = a.fli != nullptr && !prune_of[r] && epoch_of[r] >= 0;  // (as launch_step_g chooses)
        it.sfl = spec_inner ? a.fli : a.fl;
        it.sdeg = spec_inner ? a.degi : a.deg;
        it.nblocks = (unsigned)(((int64_t)a.n * p->g + 255) / 256);
        it.epoch = epoch_of[r];
        it.n_kicks = nk_of[r];
        it.do_aoa = do_aoa;
        it.prune = prune_of[r] ? 1 : 0;
        it.pad = 0;
      }
      const int rc = chx_launch_step_batch(pack, C.count, C.g, C.hot, oc > 0, st);
      if (rc != CHX_OK) return rc;
    }
    for (int r = 0; r < n_items; ++r)
      if (active(r, i)) items[r].n_enqueued = i + 1;
  }
  return CHX_OK;
}
