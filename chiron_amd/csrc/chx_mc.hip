// libchx: device-resident MonteCarloDisplacementMove loop on the force list.
//
// The reference (chiron/mcmc.py:243-306 -> :357-463 -> :680-787) runs one trial as
//   key = split; x' = wrap(x + sigma N(0,1)); [rebuild list if skin check]; u' = beta U(x');
//   key = split; accept iff -(u'-u) >= 0 or uniform(key) < exp(-(u'-u))
// with a host round trip per trial.  Here a whole chunk of trials is enqueued at once:
// two launches per trial (decision of the previous trial + proposal + skin check; list-kernel energy), the
// accepted configuration changes hands by flipping a buffer index in device memory, the
// PRNG chain key lives on the device and advances by the same two splits per trial, and
// the first trial whose proposal violates the skin halts the chunk (every later launch is
// a no-op) so the host can run exactly that trial through the rebuild path.
#include "chx_common.h"
#include "chx_internal.h"

#include <limits.h>
#include <stdlib.h>

struct McArgs {
  int n;
  float4* xa;
  float4* xb;
  const float4* ref4;
  const uint32_t* perm;
  const float* subset;
  Box3 box;
  float T_half_skin;
  float sigma_d;
  double beta;
  int32_t* state;
  double* energy;
  int32_t* accept_log;
  float* energy_log;
  const double* ws;
  int nblocks;  // list-kernel partials in ws
};

__device__ __forceinline__ void mc_split(uint32_t k0, uint32_t k1, uint32_t& c0, uint32_t& c1,
                                         uint32_t& s0, uint32_t& s1) {
  const uint4 o = philox4x32(make_uint4(0u, 0u, 0u, CHX_STREAM_SPLIT), k0, k1);
  c0 = o.x;
  c1 = o.y;
  s0 = o.z;
  s1 = o.w;
}

// One launch of the trial loop.  DECIDE: take the decision of trial `trial - 1` from the partials its energy launch
// left in ws -- EVERY workgroup does, redundantly and identically (fixed summation order, one thread's arithmetic),
// because the proposal below needs the outcome and a launch boundary costs more than this; workgroup 0 records it.
// PROPOSE: propose trial `trial`.  The chain state (current buffer, counters, key; energies) lives in two slots that
// take turns by trial parity: the state BEFORE trial t's proposal sits in slot t & 1, so a launch reads one slot
// and (workgroup 0) writes the other, and no workgroup reads what another one of the same launch writes.
template <bool DECIDE, bool PROPOSE>
__global__ __launch_bounds__(256) void k_mc_step(McArgs a, long long trial, int log_slot) {
  if ((long long)a.state[CHX_MC_HALT] <= trial - (DECIDE ? 1 : 0)) return;  // (the halt word is shared: slot 0)
  const int p = blockIdx.x * 256 + threadIdx.x;
  // both candidates for "the current configuration" are requested before the decision is known
  float4 xa_p = make_float4(0.f, 0.f, 0.f, 0.f), xb_p = xa_p;
  if (PROPOSE && p < a.n) {
    xa_p = a.xa[p];
    xb_p = a.xb[p];
  }
  const int32_t* in = a.state + CHX_MC_SLOT_WORDS * (int)((DECIDE ? trial - 1 : trial) & 1);
  int cur = in[CHX_MC_CUR];
  uint32_t key0 = (uint32_t)in[CHX_MC_KEY0], key1 = (uint32_t)in[CHX_MC_KEY1];
  if (DECIDE) {
    __shared__ int32_t sh[3];
    double s = 0.0;
    for (int i = threadIdx.x; i < a.nblocks; i += 256) s += a.ws[i];
    s = block_sum<256>(s);
    if (threadIdx.x == 0) {
      const double* e_in = a.energy + 2 * (int)((trial - 1) & 1);
      const float e_prop = (float)s;  // the potential energy is an fp32 number (k_sum_ws)
      uint32_t c0, c1, p0, p1, d0, d1, u0, u1;
      mc_split(key0, key1, c0, c1, p0, p1);  // proposal key
      mc_split(c0, c1, d0, d1, u0, u1);      // acceptance key
      const double U = (double)u32_to_uniform(philox4x32(make_uint4(0u, 0u, 0u, CHX_STREAM_UNIFORM), u0, u1).x);
      const double up = a.beta * (double)e_prop;
      const double uc = a.beta * e_in[0];
      const double lp = -up + uc;  // log_proposal_ratio  :770
      int acc = 0;
      if (!isnan(up)) acc = (-lp <= 0.0 || U < exp(lp)) ? 1 : 0;  // :428-429, :544-545
      const int new_cur = cur ^ acc;
      sh[0] = new_cur;
      sh[1] = (int32_t)d0;
      sh[2] = (int32_t)d1;
      if (blockIdx.x == 0) {
        int32_t* out = a.state + CHX_MC_SLOT_WORDS * (int)(trial & 1);
        double* e_out = a.energy + 2 * (int)(trial & 1);
        const double e_cur = acc ? (double)e_prop : e_in[0];
        out[CHX_MC_CUR] = new_cur;
        out[CHX_MC_ACCEPTED] = in[CHX_MC_ACCEPTED] + acc;
        out[CHX_MC_PROPOSED] = in[CHX_MC_PROPOSED] + 1;
        out[CHX_MC_KEY0] = (int32_t)d0;
        out[CHX_MC_KEY1] = (int32_t)d1;
        e_out[0] = e_cur;
        e_out[1] = (double)e_prop;
        if (a.accept_log) a.accept_log[log_slot] = acc;
        if (a.energy_log) a.energy_log[log_slot] = (float)e_cur;
      }
    }
    __syncthreads();
    cur = sh[0];
    key0 = (uint32_t)sh[1];
    key1 = (uint32_t)sh[2];
  }
  if (!PROPOSE || p >= a.n) return;
  const bool cur_b = (cur & 1) != 0;
  const float4 xi = cur_b ? xb_p : xa_p;
  float4* xo = cur_b ? a.xa : a.xb;
  uint32_t c0, c1, k0, k1;
  mc_split(key0, key1, c0, c1, k0, k1);
  const uint32_t o = a.perm ? a.perm[p] : (uint32_t)p;
  float z0, z1, z2;
  philox_normals3(o, 0u, CHX_STREAM_MC_DISPLACE, k0, k1, z0, z1, z2);
  // normal * sigma [* mask] then x + d   (chiron/mcmc.py:733-748)
  float dx = fmul(z0, a.sigma_d), dy = fmul(z1, a.sigma_d), dz = fmul(z2, a.sigma_d);
  if (a.subset) {
    const float mk = a.subset[o];
    dx = fmul(dx, mk);
    dy = fmul(dy, mk);
    dz = fmul(dz, mk);
  }
  float xn = fadd(xi.x, dx), yn = fadd(xi.y, dy), zn = fadd(xi.z, dz);
  if (a.box.periodic) {  // :751-755
    xn = wrap1(xn, a.box.lx);
    yn = wrap1(yn, a.box.ly);
    zn = wrap1(zn, a.box.lz);
  }
  xo[p] = make_float4(xn, yn, zn, 0.0f);
  const float4 r0 = a.ref4[p];
  float rx, ry, rz;
  displacement(a.box, xn, yn, zn, r0.x, r0.y, r0.z, rx, ry, rz);
  if (norm2(rx, ry, rz) > a.T_half_skin) atomicMin(&a.state[CHX_MC_HALT], (int)trial);  // neighbors.py:868
}

extern "C" int chx_mc_displace_chunk(const chx_mc_t* p, int64_t trial0, int32_t n_trials,
                                     chx_stream_t stream) {
  CHX_REQUIRE(p != nullptr, "null parameter block");
  CHX_REQUIRE(p->n > 0, "n <= 0");
  CHX_REQUIRE(n_trials >= 0, "n_trials < 0");
  CHX_REQUIRE(trial0 >= 0 && trial0 + n_trials < (int64_t)INT_MAX, "trial index out of range");
  CHX_REQUIRE(p->x_a && p->x_b && p->ref4 && p->fl && p->deg && p->state && p->energy && p->ws,
              "null pointer");
  CHX_REQUIRE(p->fl_chunks >= 1, "fl_chunks < 1");
  CHX_REQUIRE(!p->box.periodic || (p->box.lx > 0 && p->box.ly > 0 && p->box.lz > 0),
              "box lengths must be > 0");
  hipStream_t st = chx_stream(stream);
  McArgs a;
  a.n = p->n;
  a.xa = reinterpret_cast<float4*>(p->x_a);
  a.xb = reinterpret_cast<float4*>(p->x_b);
  a.ref4 = reinterpret_cast<const float4*>(p->ref4);
  a.perm = p->perm;
  a.subset = p->subset_mask;
  a.box = make_box3(p->box);
  a.T_half_skin = chx_cutoff_threshold_h(p->skin * 0.5f);
  a.sigma_d = p->sigma_d;
  a.beta = p->beta;
  a.state = p->state;
  a.energy = p->energy;
  a.accept_log = p->accept_log;
  a.energy_log = p->energy_log;
  a.ws = p->ws;
  a.nblocks = chx_lj_fl_blocks(p->n, p->g);
  chx_fl_select_t sel;
  sel.halt = p->state + CHX_MC_HALT;
  sel.x4_alt = p->x_b;
  sel.flip = 1;  // the proposed configuration sits in the buffer that is NOT current
  const unsigned pb = (unsigned)((p->n + 255) / 256);
  // CHX_MC_DECIDE_LAUNCH=1: every decision a one-block launch of its own (three launches per trial; A/B, tests)
  static const bool decide_launch = [] { const char* e = getenv("CHX_MC_DECIDE_LAUNCH"); return e && e[0] == '1'; }();
  for (int i = 0; i < n_trials; ++i) {
    const long long t = trial0 + i;
    if (i == 0 || decide_launch) {
      hipLaunchKernelGGL((k_mc_step<false, true>), dim3(pb), dim3(256), 0, st, a, t, 0);
    } else {
      hipLaunchKernelGGL((k_mc_step<true, true>), dim3(pb), dim3(256), 0, st, a, t, i - 1);  // decides trial t - 1
    }
    sel.state = p->state + CHX_MC_SLOT_WORDS * (int)(t & 1);
    sel.trial = t;
    int rc = chx_launch_lj_fl(p->x_a, p->fl, p->deg, p->n, p->g, p->fl_chunks, p->box, p->sigma,
                              p->epsilon, p->cutoff, nullptr, nullptr, p->ws, nullptr, st, &sel);
    if (rc != CHX_OK) return rc;
    if (i == n_trials - 1 || decide_launch) {
      hipLaunchKernelGGL((k_mc_step<true, false>), dim3(1), dim3(256), 0, st, a, t + 1, i);  // decides trial t
    }
  }
  return chx_check_launch("chx_mc_displace_chunk");
}
