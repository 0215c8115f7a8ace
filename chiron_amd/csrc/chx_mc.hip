// libchx: device-resident MonteCarloDisplacementMove loop on the force list -- ONE launch per trial.
//
// The reference (chiron/mcmc.py:243-306 -> :357-463 -> :680-787) runs one trial as
//   key = split; x' = wrap(x + sigma N(0,1)); [rebuild list if skin check]; u' = beta U(x');
//   key = split; accept iff -(u'-u) >= 0 or uniform(key) < exp(-(u'-u))
// with a host round trip per trial.  Here a whole chunk of trials is enqueued at once and trial t is one launch
// (k_mc_trial; rounds 1-3 used two: a proposal launch and an energy launch):
//   head   the decision of trial t-1, taken by every workgroup from the per-block energy partials launch t-1 left
//          (redundantly and identically: a launch of its own would cost more than the arithmetic);
//   body   the energy of the proposal x'_t over the force list (the pair-force kernel's row loop, chx_lj_row.h);
//   tail   the NEXT proposal.  The chain key advances by two splits per trial whatever the decisions are, so the
//          displacement d_{t+1} of every particle is known now; what is not known is which configuration it will be
//          added to -- x_t (trial t rejected) or x'_t (accepted).  The launch writes BOTH candidates,
//          wrap(x_t + d_{t+1}) and wrap(x'_t + d_{t+1}), next to the pair (x_t, x'_t) itself, and launch t+1 picks by
//          the decision it takes at its head: the energy loop gathers its neighbours from ONE array, like the
//          pair-force kernel, and no launch reads what it writes (the buffers alternate with the trial's parity).
// Because the key chain does not depend on decisions, the HOST works it out for the whole chunk (a few hundred
// microseconds for 4096 trials, chx_mc_key_chain_h) and uploads one row per trial -- its proposal key and the uniform
// of its decision: five Philox blocks per wavefront and launch on the device cost 1.7 us of a 9 us launch (v_mul_hi_u32
// is a quarter-rate instruction; measured, round 4), one scalar load costs nothing.
// The first trial whose proposal violates the list's skin (chiron/neighbors.py:868) halts the chunk -- every later
// launch is a no-op -- so the host can run exactly that trial through the reference's rebuild branch.
#include "chx_common.h"
#include "chx_internal.h"
#include "chx_lj_row.h"
#include "chx_launch.h"

#include <limits.h>
#include <stdlib.h>

struct McArgs {
  int n;
  float4* cur4;   // [parity][copy][n]: the configuration before the parity's trial (which copy: the decision)
  float4* prop4;  // [parity][copy][n]: the two candidates for the trial's proposal
  const float4* ref4;
  const uint32_t* perm;
  const float* subset;
  const uint32_t* fl;
  const int32_t* deg;
  int chunks;
  Box3 box;
  LJParams lj;
  float T_half_skin;
  float sigma_d;
  double beta;
  int32_t* state;
  double* energy;
  int32_t* accept_log;
  float* energy_log;
  double* ws;   // two halves of ws_half doubles: launch t writes half t & 1 and reads the other (the partials of t - 1)
  int ws_half;
  int nblocks;  // per-block energy partials per half
  const uint4* keyrow;  // key table row of the launch's own trial: {proposal key (2 words), uniform of its decision and its logarithm (fp32 bits)}
  // the inner list (see "Inner list" below); fli == NULL: off
  uint32_t* fli;    // same geometry as fl
  int32_t* cnti;    // (n, G): entries of a lane's even words | odd words << 16
  float4* refi4;    // (n,4): the proposal the inner list was taken at
  float T_inner;       // r2 <= T_inner: listed (cutoff + inner skin)
  float T_inner_half;  // a candidate further than this from refi4 asks for a new inner list
};

// ---------------------------------------------------------------------------------------------------------------
// Inner list.  Two thirds of the force list's entries lie between the cutoff and cutoff + skin and contribute exact
// zeros to every trial's energy.  A launch that is told to (state[CHX_MC_REFRESH] == its trial: the first of a chunk,
// then on demand) walks the force list as before and, on the way, copies the entries within cutoff + inner skin of its
// proposal into an inner list that the following launches walk instead, until a particle has strayed inner skin / 2
// from where it stood then: every launch checks BOTH candidates it writes for the next trial against that position and
// books a refresh for the next trial if one fails -- one trial ahead, so the launch that would misuse the list is the
// one that renews it.
// The refresh word follows the rule of the state slots: launch t READS state[CHX_MC_REFRESH] of slot t & 1 and BOOKS
// into the word of the other slot, so no workgroup tests a word that a workgroup of the same launch writes (a grid
// that is not co-resident -- N = 262 144 with four lanes per particle, liquid density with eight -- runs its last
// workgroups after its first have finished: with ONE word they would see the booking for t + 1, skip the refresh
// they are part of and walk rows this launch has not written.  Round-4 review).
// The energy has to come out in the BITS the force list gives (the host loop of MCMove and compute_energy use it, and a
// decision is a comparison of those bits).  A lane adds its entries up in two accumulators, the even and the odd words
// of its quads in chunk order, and an entry beyond the cutoff adds an exact zero: so the inner list is compacted PER
// LANE AND PER WORD PARITY -- every kept entry stays with its lane and its accumulator, in its order -- and the sums
// are the same additions without the zeros.  cnti holds the two lengths of a lane; unused words of the chunks a lane
// walks hold the row's own index (gathered, never counted).
// ---------------------------------------------------------------------------------------------------------------
struct InnerOut {
  uint32_t* lane;  // this lane's word 0 of chunk 0
  size_t stride;   // words between consecutive chunks
  int ne, no;
};
__device__ __forceinline__ void inner_put(InnerOut& o, bool odd, uint32_t ent) {
  const int s = odd ? o.no++ : o.ne++;
  o.lane[(size_t)(s >> 1) * o.stride + (size_t)(2 * (s & 1) + (odd ? 1 : 0))] = ent;
}

// the force-list walk of lj_row_loop (same calls, same order: same energy bits) that also fills the lane's inner list
template <int G, int MODE>
__device__ __forceinline__ void mc_refresh_loop(const float4* __restrict__ x4, const uint4* __restrict__ col, size_t stride,
                                                int d, int nc, int q, const float4& xi, uint4 e4, float4 (&xj)[4],
                                                const Box3& box, const LJParams& lj, RowAcc& A, float T_inner,
                                                InnerOut& o) {
  constexpr int W = 4 * G;
  int c = 0;
  if (G > 1 || nc > 0) do {
    const int k0 = c * W + q;
    uint4 nx = make_uint4(0u, 0u, 0u, 0u);
    if (c + 1 < nc) nx = col[(size_t)(c + 1) * stride];
    if (k0 < d) {
      lj_pair2<MODE, true>(xi, (f2){xj[0].x, xj[0].y}, (f2){xj[0].z, xj[0].w}, (f2){xj[1].x, xj[1].y},
                           (f2){xj[1].z, xj[1].w}, e4.x, e4.y, k0 + G < d, box, lj, T_inner, A);
      if (A.keep0) inner_put(o, false, e4.x);
      if (A.keep1) inner_put(o, true, e4.y);
    }
    if (k0 + 2 * G < d) {
      lj_pair2<MODE, true>(xi, (f2){xj[2].x, xj[2].y}, (f2){xj[2].z, xj[2].w}, (f2){xj[3].x, xj[3].y},
                           (f2){xj[3].z, xj[3].w}, e4.z, e4.w, k0 + 3 * G < d, box, lj, T_inner, A);
      if (A.keep0) inner_put(o, false, e4.z);
      if (A.keep1) inner_put(o, true, e4.w);
    }
    if (c + 1 < nc) {
      e4 = nx;
      xj[0] = x4[e4.x & ~CHX_FL_HI];
      xj[1] = x4[e4.y & ~CHX_FL_HI];
      xj[2] = x4[e4.z & ~CHX_FL_HI];
      xj[3] = x4[e4.w & ~CHX_FL_HI];
    }
  } while (++c < nc);
}

// the walk of a lane's inner list: ne even-word entries, no odd-word entries, nc = chunks that hold them
template <int G, int MODE>
__device__ __forceinline__ void mc_inner_loop(const float4* __restrict__ x4, const uint4* __restrict__ col, size_t stride,
                                              int ne, int no, int nc, const float4& xi, uint4 e4, float4 (&xj)[4],
                                              const Box3& box, const LJParams& lj, RowAcc& A) {
  int c = 0;
  do {  // (at least one pass, like lj_row_loop: chunk 0 is in flight already)
    uint4 nx = make_uint4(0u, 0u, 0u, 0u);
    if (c + 1 < nc) nx = col[(size_t)(c + 1) * stride];
    const bool a0 = 2 * c < ne, a1 = 2 * c < no, a2 = 2 * c + 1 < ne, a3 = 2 * c + 1 < no;
    if (a0 | a1)
      lj_pair2<MODE, false, true>(xi, (f2){xj[0].x, xj[0].y}, (f2){xj[0].z, xj[0].w}, (f2){xj[1].x, xj[1].y},
                                  (f2){xj[1].z, xj[1].w}, e4.x, e4.y, a1, box, lj, 0.f, A, a0);
    if (a2 | a3)
      lj_pair2<MODE, false, true>(xi, (f2){xj[2].x, xj[2].y}, (f2){xj[2].z, xj[2].w}, (f2){xj[3].x, xj[3].y},
                                  (f2){xj[3].z, xj[3].w}, e4.z, e4.w, a3, box, lj, 0.f, A, a2);
    if (c + 1 < nc) {
      e4 = nx;
      xj[0] = x4[e4.x & ~CHX_FL_HI];
      xj[1] = x4[e4.y & ~CHX_FL_HI];
      xj[2] = x4[e4.z & ~CHX_FL_HI];
      xj[3] = x4[e4.w & ~CHX_FL_HI];
    }
  } while (++c < nc);
}

// general path of one lane (lj_row_finish's: one entry at a time through the general minimum image) over `nc` chunks of
// `row`; entry (c, u) counts when act(c, u)
template <typename Act>
__device__ __forceinline__ void mc_general_lane(const float4* __restrict__ x4, const uint32_t* __restrict__ row, size_t stride_words,
                                                int nc, const float4& xi, const Box3& box, const LJParams& lj, RowAcc& A, Act act) {
  A.ax = A.ay = A.az = A.ae = f2_splat(0.f);
  for (int c = 0; c < nc; ++c) {
    for (int u = 0; u < 4; ++u) {
      if (!act(c, u)) continue;
      const uint32_t en_t = row[(size_t)c * stride_words + u];
      const float4 xo = x4[en_t & ~CHX_FL_HI];
      const bool hi = (en_t & CHX_FL_HI) != 0u;
      float gx, gy, gz;
      displacement_oriented(box, hi, xi.x, xi.y, xi.z, xo.x, xo.y, xo.z, gx, gy, gz);
      const float g2 = norm2(gx, gy, gz);
      if (g2 <= lj.T) {
        float e, fs;
        lj_pair(g2, lj.sig2, lj.eps4, lj.eps24, e, fs);
        A.ae[u & 1] += e;  // same accumulator parity as the fast path
      }
    }
  }
}

// energy share of one lane, lj_row_finish's bits.  REFRESH: over the force list (L: its loads), filling the inner list;
// else over the inner list (L: the inner list's loads, d_raw = the lane's packed lengths).
template <int G, bool REFRESH>
__device__ __forceinline__ float mc_row_energy(RowLoads& L, const float4* __restrict__ x4, const uint32_t* __restrict__ list, int n,
                                               int q, bool valid, const Box3& box, bool periodic, const LJParams& lj,
                                               const McArgs& a, float4& xi_out) {
  constexpr int W = 4 * G;
  const int pc = L.pc;
  const size_t stride = (size_t)n * G;  // uint4 units between consecutive chunks
  const float4 xi = L.xi;
  xi_out = xi;
  RowAcc A;
  A.ax = A.ay = A.az = A.ae = f2_splat(0.f);
  A.bad = false;
  A.keep0 = A.keep1 = false;
  const uint32_t* row = list + ((size_t)pc * G + q) * 4;
  if (REFRESH) {
    const int d = valid ? L.d_raw : 0;
    const int nc = (d + W - 1) / W;
    InnerOut o = {a.fli + ((size_t)pc * G + q) * 4, stride * 4, 0, 0};
    if (!periodic) mc_refresh_loop<G, LJ_OPEN>(x4, L.col, stride, d, nc, q, xi, L.e4, L.xj, box, lj, A, a.T_inner, o);
    else mc_refresh_loop<G, LJ_SHIFT>(x4, L.col, stride, d, nc, q, xi, L.e4, L.xj, box, lj, A, a.T_inner, o);
    if (valid) {
      // whole chunks (at least chunk 0, which every launch gathers before it knows the lengths) must hold valid indices
      int nci = ((o.ne > o.no ? o.ne : o.no) + 1) >> 1;
      nci = nci < 1 ? 1 : nci;
      const int ne = o.ne, no = o.no;
      while (o.ne < 2 * nci) inner_put(o, false, (uint32_t)pc);
      while (o.no < 2 * nci) inner_put(o, true, (uint32_t)pc);
      a.cnti[(size_t)pc * G + q] = ne | (no << 16);
    }
    if (__builtin_expect(A.bad, 0))
      mc_general_lane(x4, row, stride * 4, nc, xi, box, lj, A, [&](int c, int u) { return c * W + u * G + q < d; });
  } else {
    const int ne = valid ? (L.d_raw & 0xFFFF) : 0, no = valid ? ((L.d_raw >> 16) & 0xFFFF) : 0;
    const int nc = ((ne > no ? ne : no) + 1) >> 1;
    if (!periodic) mc_inner_loop<G, LJ_OPEN>(x4, L.col, stride, ne, no, nc, xi, L.e4, L.xj, box, lj, A);
    else mc_inner_loop<G, LJ_SHIFT>(x4, L.col, stride, ne, no, nc, xi, L.e4, L.xj, box, lj, A);
    if (__builtin_expect(A.bad, 0))
      mc_general_lane(x4, row, stride * 4, nc, xi, box, lj, A, [&](int c, int u) { return 2 * c + (u >> 1) < ((u & 1) ? no : ne); });
  }
  return 0.f + (A.ae.x + A.ae.y);
}

// sigma N(0,1) [* mask] of one particle (chiron/mcmc.py:733-748; k: the proposal key of the trial, o: original particle id)
__device__ __forceinline__ void mc_displacement(const McArgs& a, uint32_t o, uint32_t k0, uint32_t k1, float& dx, float& dy,
                                                float& dz) {
  float z0, z1, z2;
  philox_normals3(o, 0u, CHX_STREAM_MC_DISPLACE, k0, k1, z0, z1, z2);
  dx = fmul(z0, a.sigma_d), dy = fmul(z1, a.sigma_d), dz = fmul(z2, a.sigma_d);
  if (a.subset) {
    const float mk = a.subset[o];
    dx = fmul(dx, mk);
    dy = fmul(dy, mk);
    dz = fmul(dz, mk);
  }
}
// x + d, wrapped (:748-755)
__device__ __forceinline__ float4 mc_displaced(const McArgs& a, float4 x, float dx, float dy, float dz) {
  float xn = fadd(x.x, dx), yn = fadd(x.y, dy), zn = fadd(x.z, dz);
  if (a.box.periodic) {
    xn = wrap1(xn, a.box.lx);
    yn = wrap1(yn, a.box.ly);
    zn = wrap1(zn, a.box.lz);
  }
  return make_float4(xn, yn, zn, 0.0f);
}

// this thread's share of the energy partials launch `trial - 1` left
__device__ __forceinline__ double mc_partials(const McArgs& a, long long trial) {
  const double* ws_in = a.ws + (size_t)a.ws_half * (int)((trial - 1) & 1);
  double part = 0.0;
  for (int i = threadIdx.x; i < a.nblocks; i += 256) part += ws_in[i];
  return part;
}

// The decision of trial `trial - 1` (chiron/mcmc.py:428-436, :531-548) from those partials and the trial's uniform: one
// thread's arithmetic in a fixed summation order, so every workgroup that calls this arrives at the same bits;
// workgroup 0 records it in the slot of `trial`.
// (log_u: fp32 log of the uniform, made on the host with the key table: most decisions are then a comparison -- the
// fp64 exponential sits on the launch's dependent chain, one thread computing it while 255 wait -- and the exponential is
// only evaluated when log_prob and log(uniform) are closer than the fp32 rounding of the latter could explain, so every
// decision is the one `uniform < exp(log_prob)` gives)
__device__ __forceinline__ int mc_decide(const McArgs& a, long long trial, int log_slot, double part, float uniform, float log_u) {
  __shared__ int32_t sh_acc;
  const int b = (int)(trial & 1);
  const int32_t* in = a.state + CHX_MC_SLOT_WORDS * (b ^ 1);
  const double s = block_sum<256>(part);
  if (threadIdx.x == 0) {
    const double* e_in = a.energy + 2 * (b ^ 1);
    const float e_prop = (float)s;  // the potential energy is an fp32 number (k_sum_ws)
    const double U = (double)uniform;
    const double up = a.beta * (double)e_prop;
    const double uc = a.beta * e_in[0];
    const double lp = -up + uc;  // log_proposal_ratio  :770
    int acc = 0;
    if (!isnan(up)) {  // :428-429, :544-545
      const double lu = (double)log_u;
      const double tol = 4.0e-7 * (fabs(lu) + 1.0);
      if (-lp <= 0.0 || lp > lu + tol) acc = 1;
      else if (lp < lu - tol) acc = 0;
      else acc = U < exp(lp) ? 1 : 0;
    }
    sh_acc = acc;
    if (blockIdx.x == 0) {
      int32_t* out = a.state + CHX_MC_SLOT_WORDS * b;
      double* e_out = a.energy + 2 * b;
      const double e_cur = acc ? (double)e_prop : e_in[0];
      out[CHX_MC_CUR] = acc;  // which copy of cur4[trial & 1] holds the configuration before trial `trial`
      out[CHX_MC_ACCEPTED] = in[CHX_MC_ACCEPTED] + acc;
      out[CHX_MC_PROPOSED] = in[CHX_MC_PROPOSED] + 1;
      e_out[0] = e_cur;
      e_out[1] = (double)e_prop;
      if (a.accept_log) a.accept_log[log_slot] = acc;
      if (a.energy_log) a.energy_log[log_slot] = (float)e_cur;
    }
  }
  __syncthreads();
  return sh_acc;
}

// the chunk's last launch: the decision of its last trial, one workgroup (a.keyrow: the row of `trial`)
__global__ __launch_bounds__(256) void k_mc_decide(McArgs a, long long trial, int log_slot) {
  const double part = mc_partials(a, trial);
  const float uniform = __uint_as_float(a.keyrow[-1].z), log_u = __uint_as_float(a.keyrow[-1].w);
  if ((long long)a.state[CHX_MC_HALT] <= trial - 1) return;
  (void)mc_decide(a, trial, log_slot, part, uniform, log_u);
}

// the chunk's first launch: the configuration before trial `trial` sits in copy state.CUR of cur4[trial & 1]; make
// both copies hold it and both candidates hold its proposal, so that the first trial launch needs no decision
__global__ __launch_bounds__(256) void k_mc_prologue(McArgs a, long long trial) {
  if ((long long)a.state[CHX_MC_HALT] <= trial) return;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  const int b = (int)(trial & 1);
  const int sel = a.state[CHX_MC_SLOT_WORDS * b + CHX_MC_CUR] & 1;
  float4* cur = a.cur4 + (size_t)2 * b * a.n;
  float4* prop = a.prop4 + (size_t)2 * b * a.n;
  const float4 xc = cur[(size_t)sel * a.n + p];
  const uint4 row = a.keyrow[0];
  const uint32_t o = a.perm ? a.perm[p] : (uint32_t)p;
  float dx, dy, dz;
  mc_displacement(a, o, row.x, row.y, dx, dy, dz);
  const float4 xp = mc_displaced(a, xc, dx, dy, dz);
  cur[(size_t)(1 - sel) * a.n + p] = xc;
  prop[p] = xp;
  prop[(size_t)a.n + p] = xp;
  if (p == 0 && a.fli != nullptr) {
    a.state[CHX_MC_SLOT_WORDS * b + CHX_MC_REFRESH] = (int)trial;  // the chunk's first trial takes the inner list
    a.state[CHX_MC_SLOT_WORDS * (b ^ 1) + CHX_MC_REFRESH] = -1;    // (nothing booked for the trial after it yet)
  }
}

// Trial `trial` in one launch (see the head of this file).  Reads the buffers of parity trial & 1, writes the other.
// DECIDE = false: the chunk's first trial (the prologue made both copies equal: nothing to decide).
// The launch is a chain of dependent round trips, not a throughput problem (like k_lj_step), so everything that does
// not depend on the decision is requested before it: the row's index quad and length, both copies of the particle's
// own x_t / x'_t, its reference position, the energy partials, the key rows.  Behind the decision only the neighbour
// gathers, the particle's next displacement (one Philox block, computed while the gathers are in flight), the pair
// arithmetic and the stores remain.
template <int G, bool DECIDE, bool INNER>
__global__ __launch_bounds__(256) void k_mc_trial(McArgs a, long long trial, int log_slot, unsigned nblocks) {
  const int b = (int)(trial & 1);
  const unsigned blk = xcd_block(blockIdx.x, nblocks);
  const int tid = blk * 256 + threadIdx.x;
  const int p = tid / G;
  const int q = tid % G;
  const bool valid = p < a.n;
  const int pc = valid ? p : 0;
  // ---- loads that do not wait for anything (relaxed wavefront-scope atomics: not sunk below the branches) ----
  const uint4* col = reinterpret_cast<const uint4*>(a.fl) + (size_t)pc * G + q;
  const uint4 e4 = pinned_load16(col);
  const int d_raw = pinned_load4(a.deg + pc);
  const float4* cur_b = a.cur4 + (size_t)2 * b * a.n;
  const float4* prop_b = a.prop4 + (size_t)2 * b * a.n;
  const uint4 xp0 = pinned_load16(prop_b + pc), xp1 = DECIDE ? pinned_load16(prop_b + a.n + pc) : xp0;
  const uint4 xc0 = pinned_load16(cur_b + pc), xc1 = DECIDE ? pinned_load16(cur_b + a.n + pc) : xc0;
  const uint4 r0b = pinned_load16(a.ref4 + pc);
  const uint32_t o = a.perm ? (uint32_t)pinned_load4(reinterpret_cast<const int32_t*>(a.perm) + pc) : (uint32_t)pc;
  const double part = DECIDE ? mc_partials(a, trial) : 0.0;
  const float uniform = DECIDE ? __uint_as_float(a.keyrow[-1].z) : 0.f;  // (the uniform of trial - 1's decision ...
  const float log_u = DECIDE ? __uint_as_float(a.keyrow[-1].w) : 0.f;    // ... and its logarithm)
  const uint4 next = a.keyrow[1];                                         // (.xy: the proposal key of trial + 1)
  // (inner list: its first quad, the lane's lengths and the position it was taken at -- requested whether this launch
  // turns out to walk it or to renew it)
  const uint4* coli = INNER ? reinterpret_cast<const uint4*>(a.fli) + (size_t)pc * G + q : col;
  const uint4 e4i = INNER ? pinned_load16(coli) : e4;
  const int cnt_raw = INNER ? pinned_load4(a.cnti + (size_t)pc * G + q) : 0;
  const uint4 rib = INNER ? pinned_load16(a.refi4 + pc) : r0b;
  // (launch-uniform: the word of slot b is written by the prologue and by launch trial - 1 only; this launch books into
  // the other slot's)
  const bool refresh = INNER && (long long)a.state[CHX_MC_SLOT_WORDS * b + CHX_MC_REFRESH] == trial;
  if ((long long)a.state[CHX_MC_HALT] <= trial - (DECIDE ? 1 : 0)) return;  // (the halt word is shared: slot 0)
  // ---- the decision of trial - 1: every workgroup, same bits ----
  const int sel = DECIDE ? mc_decide(a, trial, log_slot, part, uniform, log_u) : 0;
  // ---- energy of x'_t over the force list: the pair-force kernel's row loop and its reduction, bit for bit ----
  const float4* prop = prop_b + (size_t)sel * a.n;  // x'_t
  const uint4 xpb = sel ? xp1 : xp0, xcb = sel ? xc1 : xc0;
  // (the neighbour gathers are in flight from here.  Gathering the first chunk from BOTH candidates ahead of the
  // decision -- so that the two chains run side by side -- was measured in round 4: 116 k -> 110-113 k trials/s, the
  // four extra scattered loads per lane cost more than the decision's latency)
  const bool walk_inner = INNER && !refresh;
  RowLoads rl = lj_row_adopt<G>(prop, walk_inner ? coli : col, pc, xpb, walk_inner ? cnt_raw : d_raw, walk_inner ? e4i : e4);
  // this particle's next displacement sigma N(0,1) [* mask] (chiron/mcmc.py:733-748), under the gathers' latency
  float dx, dy, dz;
  mc_displacement(a, o, next.x, next.y, dx, dy, dz);
  float4 xp;
  float en = 0.f;
  if (!INNER) {
    float fx = 0.f, fy = 0.f, fz = 0.f;
    lj_row_finish<G>(rl, prop, a.fl, a.n, q, valid, a.box, a.box.periodic != 0, a.lj, xp, fx, fy, fz, en);
  } else if (refresh) {
    en = mc_row_energy<G, true>(rl, prop, a.fl, a.n, q, valid, a.box, a.box.periodic != 0, a.lj, a, xp);
  } else {
    en = mc_row_energy<G, false>(rl, prop, a.fli, a.n, q, valid, a.box, a.box.periodic != 0, a.lj, a, xp);
  }
  en = group_sum<G>(en);
  double mine = (valid && q == 0) ? 0.5 * (double)en : 0.0;  // every pair is visited from both ends
  mine = block_sum<256>(mine);
  if (threadIdx.x == 0) a.ws[(size_t)a.ws_half * b + blk] = mine;  // (never the half the heads of this launch are reading)
  if (INNER && refresh && blockIdx.x == 0 && threadIdx.x == 0) a.state[CHX_MC_REFRESHES] += 1;  // (statistics)
  if (!valid) return;
  // ---- the tail: four jobs per particle, dealt to its G lanes (role = q, q + G, ...: one each with four lanes or more)
  // so that a wavefront issues the tail's instructions once, not four times over for lane 0 --
  //   role 0  skin check of this trial's proposal (chiron/neighbors.py:868: the first violating trial halts the chunk)
  //           and the configuration should the trial be rejected;
  //   role 1, 2  the next trial's candidates wrap(x_t + d), wrap(x'_t + d) (chiron/mcmc.py:733-755 on either outcome),
  //           each checked against the inner list's guarantee one trial ahead: whichever becomes the next proposal must
  //           lie within inner skin / 2 of where this particle stood when the list was taken (this launch's own
  //           proposal if it renews it);
  //   role 3  the configuration should the trial be accepted.
  // One instruction stream for all roles: a base position (+ displacement, wrapped), one minimum-image distance against
  // the role's reference and threshold, one 16-byte store to the role's slot. ----
  const float4 xc = make_float4(__uint_as_float(xcb.x), __uint_as_float(xcb.y), __uint_as_float(xcb.z), 0.0f);
  const float4 xpc = make_float4(xp.x, xp.y, xp.z, 0.0f);
  float4* cur_n = a.cur4 + (size_t)2 * (b ^ 1) * a.n;
  float4* prop_n = a.prop4 + (size_t)2 * (b ^ 1) * a.n;
  float rfx = __uint_as_float(rib.x), rfy = __uint_as_float(rib.y), rfz = __uint_as_float(rib.z);
  if (INNER && refresh) rfx = xp.x, rfy = xp.y, rfz = xp.z;
#pragma unroll
  for (int role = q; role < 4; role += G) {
    const bool cand = role == 1 || role == 2;
    const float4 base = (role == 0 || role == 1) ? xc : xpc;
    const float4 out = cand ? mc_displaced(a, base, dx, dy, dz) : base;
    float4* dst = (cand ? prop_n : cur_n) + (size_t)((role == 2 || role == 3) ? a.n : 0) + p;
    *dst = out;
    if (role == 3 || (cand && !INNER)) continue;
    // (role 0 measures the PROPOSAL against the list's reference positions; a candidate against the inner list's)
    const float px = role == 0 ? xp.x : out.x, py = role == 0 ? xp.y : out.y, pz = role == 0 ? xp.z : out.z;
    const float qx = role == 0 ? __uint_as_float(r0b.x) : rfx, qy = role == 0 ? __uint_as_float(r0b.y) : rfy,
                qz = role == 0 ? __uint_as_float(r0b.z) : rfz;
    float rx, ry, rz;
    displacement(a.box, px, py, pz, qx, qy, qz, rx, ry, rz);
    const float r2 = norm2(rx, ry, rz);
    if (role == 0) {
      if (r2 > a.T_half_skin) atomicMin(&a.state[CHX_MC_HALT], (int)trial);
    } else if (r2 > a.T_inner_half) {
      // (every booking lane stores the same value: a relaxed store, not an atomic -- with sigma_d of a few hundredths of
      // a nanometre most particles book, and read-modify-writes of one address serialise chip-wide)
      __hip_atomic_store(&a.state[CHX_MC_SLOT_WORDS * (b ^ 1) + CHX_MC_REFRESH], (int)trial + 1, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (INNER && refresh && q == 0) a.refi4[p] = xpc;
}

template <int G>
static void launch_trial(const McArgs& a, long long t, bool decide, int log_slot, hipStream_t st) {
  const unsigned blocks = (unsigned)(((int64_t)a.n * G + 255) / 256);
  // (CHX_LAUNCH: an AQL packet on the step queue, which chx_mc_displace_chunk has opened -- chx_launch.h)
  if (a.fli != nullptr) {
    if (decide) CHX_LAUNCH_HOT((k_mc_trial<G, true, true>), dim3(blocks), dim3(256), st, a, t, log_slot, blocks);
    else CHX_LAUNCH_HOT((k_mc_trial<G, false, true>), dim3(blocks), dim3(256), st, a, t, log_slot, blocks);
  } else {
    if (decide) CHX_LAUNCH_HOT((k_mc_trial<G, true, false>), dim3(blocks), dim3(256), st, a, t, log_slot, blocks);
    else CHX_LAUNCH_HOT((k_mc_trial<G, false, false>), dim3(blocks), dim3(256), st, a, t, log_slot, blocks);
  }
}

extern "C" int chx_mc_key_chain_h(uint32_t k0, uint32_t k1, int32_t n_trials, uint32_t* keytab_h, uint32_t* chain_out_h) {
  CHX_REQUIRE(n_trials >= 0, "n_trials < 0");
  uint32_t o[4];
  for (int i = 0; i <= n_trials; ++i) {
    if (chain_out_h) chain_out_h[2 * i] = k0, chain_out_h[2 * i + 1] = k1;
    chx_philox4x32_h(0u, 0u, 0u, CHX_STREAM_SPLIT, k0, k1, o);  // key, subkey = split(key): the proposal key (mcmc.py:724)
    const uint32_t c0 = o[0], c1 = o[1], p0 = o[2], p1 = o[3];
    chx_philox4x32_h(0u, 0u, 0u, CHX_STREAM_SPLIT, c0, c1, o);  // key, subkey = split(key): the acceptance key (:531)
    const uint32_t u0 = o[2], u1 = o[3];
    k0 = o[0], k1 = o[1];
    if (keytab_h) {
      chx_philox4x32_h(0u, 0u, 0u, CHX_STREAM_UNIFORM, u0, u1, o);
      const float U = ((float)(o[0] >> 9) + 0.5f) * (1.0f / 8388608.0f);  // u32_to_uniform (chx_common.h), exact in fp32
      uint32_t ub;
      __builtin_memcpy(&ub, &U, 4);
      const float LU = (float)log((double)U);  // (what mc_decide compares against before it falls back on exp)
      uint32_t lb;
      __builtin_memcpy(&lb, &LU, 4);
      keytab_h[4 * i] = p0, keytab_h[4 * i + 1] = p1, keytab_h[4 * i + 2] = ub, keytab_h[4 * i + 3] = lb;
    }
  }
  return CHX_OK;
}

extern "C" int chx_mc_displace_chunk(const chx_mc_t* p, int64_t trial0, int32_t n_trials,
                                     chx_stream_t stream) {
  CHX_REQUIRE(p != nullptr, "null parameter block");
  CHX_REQUIRE(p->n > 0, "n <= 0");
  CHX_REQUIRE(n_trials >= 0, "n_trials < 0");
  CHX_REQUIRE(trial0 >= 0 && trial0 + n_trials < (int64_t)INT_MAX, "trial index out of range");
  CHX_REQUIRE(p->cur4 && p->prop4 && p->ref4 && p->fl && p->deg && p->state && p->energy && p->ws,
              "null pointer");
  CHX_REQUIRE(p->keytab && p->keytab_h, "null key table");
  CHX_REQUIRE(p->fl_chunks >= 1, "fl_chunks < 1");
  CHX_REQUIRE(p->g == 1 || p->g == 2 || p->g == 4 || p->g == 8,
              "lanes per particle must be 1, 2, 4 or 8");
  CHX_REQUIRE(!p->box.periodic || (p->box.lx > 0 && p->box.ly > 0 && p->box.lz > 0),
              "box lengths must be > 0");
  if (n_trials == 0) return CHX_OK;
  hipStream_t st = chx_stream(stream);
  // the key chain of the chunk: rows 0 .. n_trials (the last one feeds the candidates the last launch writes)
  int rc = chx_mc_key_chain_h(p->k0, p->k1, n_trials, p->keytab_h, p->chain_out_h);
  if (rc != CHX_OK) return rc;
  if (hipMemcpyAsync(p->keytab, p->keytab_h, sizeof(uint32_t) * 4 * (size_t)(n_trials + 1), hipMemcpyHostToDevice, st) != hipSuccess) {
    chx_set_error("chx_mc_displace_chunk: key table upload failed");
    return CHX_ERR_LAUNCH;
  }
  McArgs a;
  a.n = p->n;
  a.cur4 = reinterpret_cast<float4*>(p->cur4);
  a.prop4 = reinterpret_cast<float4*>(p->prop4);
  a.ref4 = reinterpret_cast<const float4*>(p->ref4);
  a.perm = p->perm;
  a.subset = p->subset_mask;
  a.fl = p->fl;
  a.deg = p->deg;
  a.chunks = p->fl_chunks;
  a.box = make_box3(p->box);
  a.lj = chx_make_lj(p->sigma, p->epsilon, p->cutoff);
  a.T_half_skin = chx_cutoff_threshold_h(p->skin * 0.5f);
  a.sigma_d = p->sigma_d;
  a.beta = p->beta;
  a.state = p->state;
  a.energy = p->energy;
  a.accept_log = p->accept_log;
  a.energy_log = p->energy_log;
  a.ws = p->ws;
  a.ws_half = (int)chx_lj_ws_count(p->n);
  a.nblocks = chx_lj_fl_blocks(p->n, p->g);
  a.fli = nullptr;
  a.cnti = nullptr;
  a.refi4 = nullptr;
  a.T_inner = a.T_inner_half = 0.f;
  if (p->fli != nullptr && p->inner_skin > 0.f) {
    CHX_REQUIRE(p->cnti && p->refi4, "inner list: null pointer");
    CHX_REQUIRE((int64_t)p->fl_chunks * 2 < 32768, "inner list: too many chunks for the packed lane lengths");
    a.fli = p->fli;
    a.cnti = p->cnti;
    a.refi4 = reinterpret_cast<float4*>(p->refi4);
    a.T_inner = chx_cutoff_threshold_h(p->cutoff + p->inner_skin);
    // (1e-4 nm inside the half skin: the triangle inequality behind the guarantee is exact, the fp32 displacements
    // it is checked with are not)
    const float half = p->inner_skin * 0.5f - 1.0e-4f;
    a.T_inner_half = half > 0.f ? chx_cutoff_threshold_h(half) : 0.f;
  }
  const uint4* tab = reinterpret_cast<const uint4*>(p->keytab);
  a.keyrow = tab;
  // The chunk's launches ride on the step queue (include/chx.h: chx_steps_open): the gate goes up behind the key-table
  // upload above, the host side of a trial is a packet instead of a HIP launch, and the packets carry no acquire fence.
  // The CALLER joins (chx_steps_join) before it reads the chunk's results with anything but a libchx call.
  {
    const int rc_open = chx_aql_open(st);
    if (rc_open != CHX_OK) return rc_open;
  }
  CHX_LAUNCH(k_mc_prologue, dim3((unsigned)((p->n + 255) / 256)), dim3(256), st, a, (long long)trial0);
  for (int i = 0; i < n_trials; ++i) {
    const long long t = trial0 + i;
    const bool decide = i > 0;  // (decides trial t - 1, logged at i - 1)
    a.keyrow = tab + i;
    switch (p->g) {
      case 1: launch_trial<1>(a, t, decide, i - 1, st); break;
      case 2: launch_trial<2>(a, t, decide, i - 1, st); break;
      case 4: launch_trial<4>(a, t, decide, i - 1, st); break;
      case 8: launch_trial<8>(a, t, decide, i - 1, st); break;
      default: launch_trial<8>(a, t, decide, i - 1, st); break;
    }
  }
  a.keyrow = tab + n_trials;
  CHX_LAUNCH(k_mc_decide, dim3(1), dim3(256), st, a, (long long)(trial0 + n_trials), n_trials - 1);
  return chx_check_launch("chx_mc_displace_chunk");
}
