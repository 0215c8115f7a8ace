// libchx: Lennard-Jones energy/force kernels.
//
//  * force list (full neighbor table derived from the reference half list)
//  * list kernel: G lanes cooperate on one particle, coalesced index reads,
//    float4 gathers of neighbor positions, DPP/shuffle reduction, no atomics
//  * all-pairs kernel: LDS-staged particle tiles, 16 lanes per particle
//
// Pair geometry always follows the reference: r_ij = x_lo - x_hi (lo/hi by ORIGINAL
// particle id, chiron/neighbors.py:773-779), min-imaged as in :79, and the cutoff
// mask is the exact `d < cutoff` of :782, so the set of interacting pairs is
// bit-identical to the reference's; only the summation order differs.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>

#include "chx_common.h"
#include "chx_internal.h"
#include "chx_step.h"
#include "chx_lj_row.h"
#include "chx_aql.h"

// ---------------------------------------------------------------------------
// force list construction
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_fl_count(const uint32_t* __restrict__ nl,
                                                  const int32_t* __restrict__ n_nbr, int n, int m,
                                                  const uint32_t* __restrict__ inv,
                                                  int32_t* __restrict__ deg) {
  const int lane = lane_id();
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  int cnt = n_nbr[i];
  cnt = cnt < m ? cnt : m;
  for (int k = lane; k < cnt; k += CHX_WAVE) {
    uint32_t j = nl[(size_t)i * m + k];
    atomicAdd(&deg[inv ? inv[j] : j], 1);
  }
  if (lane == 0 && cnt > 0) atomicAdd(&deg[inv ? inv[i] : (uint32_t)i], cnt);
}

__global__ void k_imax(const int32_t* __restrict__ v, int n, int32_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int x = i < n ? v[i] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int y = __shfl_xor(x, o, CHX_WAVE);
    x = y > x ? y : x;
  }
  if (lane_id() == 0 && x > 0) atomicMax(out, x);
}

extern "C" int chx_fl_count(const uint32_t* nl, const int32_t* n_nbr, int32_t n, int32_t m,
                            const uint32_t* inv, int32_t* deg_out, int32_t* maxdeg_out,
                            chx_stream_t stream) {
  CHX_REQUIRE(n >= 0 && m >= 1, "need n >= 0, m >= 1");
  CHX_REQUIRE(maxdeg_out, "null pointer");
  hipStream_t st = chx_stream(stream);
  if (hipMemsetAsync(maxdeg_out, 0, sizeof(int32_t), st) != hipSuccess) {
    chx_set_error("chx_fl_count: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(nl && n_nbr && deg_out, "null pointer");
  if (hipMemsetAsync(deg_out, 0, sizeof(int32_t) * (size_t)n, st) != hipSuccess) {
    chx_set_error("chx_fl_count: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  const unsigned wb = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);
  hipLaunchKernelGGL(k_fl_count, dim3(wb), dim3(256), 0, st, nl, n_nbr, n, m, inv, deg_out);
  hipLaunchKernelGGL(k_imax, dim3((n + 255) / 256), dim3(256), 0, st, deg_out, n, maxdeg_out);
  return chx_check_launch("chx_fl_count");
}

__global__ __launch_bounds__(256) void k_fl_fill(const uint32_t* __restrict__ nl,
                                                 const int32_t* __restrict__ n_nbr, int n, int m,
                                                 const uint32_t* __restrict__ inv, int g,
                                                 int32_t* __restrict__ cursor,
                                                 uint32_t* __restrict__ fl) {
  const int lane = lane_id();
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  int cnt = n_nbr[i];
  cnt = cnt < m ? cnt : m;
  if (cnt == 0) return;
  const uint32_t si = inv ? inv[i] : (uint32_t)i;
  int base = 0;
  if (lane == 0) base = atomicAdd(&cursor[si], cnt);
  base = __shfl(base, 0, CHX_WAVE);
  for (int k = lane; k < cnt; k += CHX_WAVE) {
    const uint32_t j = nl[(size_t)i * m + k];  // j > i in original ids
    const uint32_t sj = inv ? inv[j] : j;
    fl[chx_fl_addr((int)si, base + k, n, g)] = sj | CHX_FL_HI;
    const int slot = atomicAdd(&cursor[sj], 1);
    fl[chx_fl_addr((int)sj, slot, n, g)] = si;
  }
}

// slots between a row's end and the end of its last touched chunk (at least chunk 0): the row's
// own index, a valid gather target whose zero-length pair keeps lj_row on its fast path
__global__ __launch_bounds__(256) void k_fl_pad(const int32_t* __restrict__ deg, int n, int g,
                                                uint32_t* __restrict__ fl) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int w = CHX_FL_QUAD * g;
  const int d = deg[p];
  int end = (d + w - 1) / w * w;
  end = end < w ? w : end;
  for (int sl = d; sl < end; ++sl) fl[chx_fl_addr(p, sl, n, g)] = (uint32_t)p;
}

extern "C" int chx_fl_fill(const uint32_t* nl, const int32_t* n_nbr, int32_t n, int32_t m,
                           const uint32_t* inv, int32_t g, int32_t* cursor_ws, uint32_t* fl_out,
                           chx_stream_t stream) {
  CHX_REQUIRE(n >= 0 && m >= 1, "need n >= 0, m >= 1");
  CHX_REQUIRE(g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64,
              "g must be a power of two <= 64");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(nl && n_nbr && cursor_ws && fl_out, "null pointer");
  hipStream_t st = chx_stream(stream);
  if (hipMemsetAsync(cursor_ws, 0, sizeof(int32_t) * (size_t)n, st) != hipSuccess) {
    chx_set_error("chx_fl_fill: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  const unsigned wb = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);
  hipLaunchKernelGGL(k_fl_fill, dim3(wb), dim3(256), 0, st, nl, n_nbr, n, m, inv, g, cursor_ws,
                     fl_out);
  // the cursors now hold the row lengths
  hipLaunchKernelGGL(k_fl_pad, dim3((n + 255) / 256), dim3(256), 0, st, cursor_ws, n, g, fl_out);
  return chx_check_launch("chx_fl_fill");
}

// 64 particles per block, each thread insertion-sorts its own column of an LDS tile
__global__ __launch_bounds__(64) void k_fl_sort(uint32_t* __restrict__ fl,
                                                const int32_t* __restrict__ deg, int n, int g) {
  extern __shared__ uint32_t tile[];  // [maxdeg][64]
  const int t = threadIdx.x;
  const int p = blockIdx.x * 64 + t;
  if (p >= n) return;
  const int d = deg[p];
  for (int k = 0; k < d; ++k) tile[k * 64 + t] = fl[chx_fl_addr(p, k, n, g)];
  for (int a = 1; a < d; ++a) {
    const uint32_t v = tile[a * 64 + t];
    const uint32_t key = v & ~CHX_FL_HI;
    int b = a - 1;
    while (b >= 0 && (tile[b * 64 + t] & ~CHX_FL_HI) > key) {
      tile[(b + 1) * 64 + t] = tile[b * 64 + t];
      --b;
    }
    tile[(b + 1) * 64 + t] = v;
  }
  for (int k = 0; k < d; ++k) fl[chx_fl_addr(p, k, n, g)] = tile[k * 64 + t];
}

extern "C" int chx_fl_sort(uint32_t* fl, const int32_t* deg, int32_t n, int32_t g, int32_t maxdeg,
                           chx_stream_t stream) {
  CHX_REQUIRE(n >= 0 && maxdeg >= 0, "need n >= 0, maxdeg >= 0");
  CHX_REQUIRE(g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64,
              "g must be a power of two <= 64");
  if (n == 0 || maxdeg <= 1) return CHX_OK;
  CHX_REQUIRE(fl && deg, "null pointer");
  const size_t lds = (size_t)maxdeg * 64 * sizeof(uint32_t);
  if (lds > 64 * 1024) {
    chx_set_error("chx_fl_sort: maxdeg %d exceeds the on-chip tile (rows stay unsorted)", maxdeg);
    return CHX_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(k_fl_sort, dim3((n + 63) / 64), dim3(64), lds, chx_stream(stream), fl, deg, n, g);
  return chx_check_launch("chx_fl_sort");
}

// ---------------------------------------------------------------------------
// energy reduction: per-block fp64 partials -> one fp32 scalar (fixed order)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sum_ws(const double* __restrict__ ws, int count,
                                                float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) s += ws[i];
  s = block_sum<256>(s);
  if (threadIdx.x == 0) out[0] = (float)s;
}

extern "C" int64_t chx_lj_ws_count(int32_t n) {
  // enough for the list kernel at g = 64 (one particle per wave) and the all-pairs kernel
  return n > 0 ? ((int64_t)n * 64 + 255) / 256 + 1 : 1;
}

// ---------------------------------------------------------------------------
// list kernel
// ---------------------------------------------------------------------------
template <int G, bool WANT_F, bool WANT_E>
__global__ __launch_bounds__(256) void k_lj_fl(const float4* __restrict__ x4,
                                               const uint32_t* __restrict__ fl,
                                               const int32_t* __restrict__ deg, int n, int chunks,
                                               Box3 box, LJParams lj, float* __restrict__ f3,
                                               double* __restrict__ ws, int32_t* ctrl) {
  if (ctrl != nullptr && ctrl[0] != 0) {  // skin check fired in this step's integrate kernel
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl[1] = 1;
    return;
  }
  const unsigned blk = xcd_block(blockIdx.x, gridDim.x);
  const int tid = blk * 256 + threadIdx.x;
  const int p = tid / G;
  const int q = tid % G;
  const bool valid = p < n;
  float4 xi;
  float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
  lj_row<G>(x4, fl, deg, n, chunks, p, q, valid, box, lj, xi, fx, fy, fz, en);
  if (WANT_F) fx = group_sum<G>(fx), fy = group_sum<G>(fy), fz = group_sum<G>(fz);
  if (WANT_E) en = group_sum<G>(en);
  if (WANT_F && valid && q == 0) store_wt3(f3 + 3 * (size_t)p, fx, fy, fz);  // write-through: see step_epilogue
  if (WANT_E) {
    // every pair is visited from both ends
    double mine = (valid && q == 0) ? 0.5 * (double)en : 0.0;
    mine = block_sum<256>(mine);
    if (threadIdx.x == 0) ws[blk] = mine;
  }
}

// ---------------------------------------------------------------------------
// fused MD step: pair force on x_cur, then (lanes 0..2 of the particle's lane
// group, one Cartesian component each) the BAOAB update that consumes it:
//   v += (dt/2) F/m   [x n_kicks: trailing kick of the previous step + leading kick]
//   x += (dt/2) v ; v = a v + b sigma_v xi ; x += (dt/2) v ; wrap ; skin check
// (chiron/integrators.py:181-195, :239-241).  Positions ping-pong between two
// buffers so no launch reads what it writes.  ctrl[3] = first step that must NOT
// run because the skin check fired (INT_MAX when clear): every later launch of the
// chunk is a no-op and the host rebuilds the list at exactly the reference's step.
// ---------------------------------------------------------------------------
// operands of one particle's integrate work
struct StepParticle {
  bool live;
  int pp;
  uint32_t o;  // original particle id (RNG counter / noise row)
  float m;
  float bs;  // b sigma_v = b sqrt(kT / m), chiron/integrators.py:131,186
  float v[3];
  float4 x, ref, snap;
  float path0;  // path length travelled since the last pruning pass, before this step (x.w; 0 in a pruning launch)
  float z[3];
};

// One Cartesian component of the BAOAB update (chiron/integrators.py:181-195) with the wrap (:239),
// the squared displacement from the list's reference positions (skin check, :241) and from the
// positions of the last pruning pass.  FAST assumes 0 <= x_new < L and |x_new - ref| < L and
// returns true when the assumption fails (the caller then takes the general instantiation):
//   * x - floor(x / L) L == x bit for bit when +0 <= x < L (x / L rounds below 1: the spacing of
//     floats under L is at least 2^-24 L), tested on the bit patterns so that -0.0 takes the
//     general path (it wraps to +0.0);
//   * min_image_fast == min_image for |d| < L (tests/test_host_logic.py, exhaustive on the CPU).
template <bool FAST, bool HOT, bool DEFER>
__device__ __forceinline__ bool integrate_component(const StepArgs& a, int n_kicks, int do_aoa, float F, float m,
                                                    float bs, float z, float v, float xo, float rf, float sf, float L,
                                                    float hL, float& vn, float& xn, float& sq, float& sqi, float& sqs) {
  // HOT: the steady-state launch (both kicks, A-O-A, periodic wrap, skin check, inner list), with
  // the launch-uniform switches folded at compile time
  const bool periodic = HOT || a.box.periodic;
  const bool has_ref = HOT || a.ref4 != nullptr;
  const bool has_inner = HOT || a.fli != nullptr;
  const bool do_wrap = HOT || (a.wrap && a.box.periodic);
  if (HOT) n_kicks = 2, do_aoa = 1;
  bool rare = false;
  const float kick = fdiv(fmul(a.hdt, F), m);
  float vc = v;
  for (int k = 0; k < n_kicks; ++k) vc = fadd(vc, kick);  // B  :181 / :195
  xn = 0.f, sq = 0.f, sqi = 0.f, sqs = 0.f;
  if (do_aoa) {
    float xc = fadd(xo, fmul(a.hdt, vc));      // A  :183
    vc = fadd(fmul(a.a, vc), fmul(bs, z));     // O  :186
    xc = fadd(xc, fmul(a.hdt, vc));            // A  :188
    // speed-only bookkeeping for the pruned inner list: this step's (unwrapped) displacement, squared
    if (has_inner) sqi = (xc - xo) * (xc - xo);
    if (do_wrap) {                             // :239
      if (FAST) {
        rare |= !(__float_as_uint(xc) < __float_as_uint(L));
      } else {
        xc = wrap1(xc, L);
      }
    }
    xn = xc;
    if (has_ref) {
      float r = fsub(xc, rf);
      if (periodic) {
        if (FAST) {
          rare |= !(fabsf(r) < L);
          r = min_image_fast(r, L, hL);
        } else {
          r = min_image(r, L, hL);
        }
      }
      sq = fmul(r, r);
    }
    if (DEFER) {  // deferred rebuilds: distance from the positions the force list was built from
      float r = fsub(xc, sf);
      if (periodic) {
        if (FAST) {
          rare |= !(fabsf(r) < L);
          r = min_image_fast(r, L, hL);
        } else {
          r = min_image(r, L, hL);
        }
      }
      sqs = fmul(r, r);
    }
  }
  vn = vc;
  return rare;
}

// The loads of one particle's integrate work (no arithmetic: see finish_step_particle).
template <bool HOT, bool PRUNE, bool DEFER = false>
__device__ __forceinline__ StepParticle load_step_operands(const StepArgs& a, long long step, int do_aoa, int p0, int pl,
                                                           int ppb) {
  StepParticle t;
  t.pp = p0 + pl;
  t.live = pl < ppb && t.pp < a.n;
  t.m = 1.f;
  t.bs = 0.f;
  t.o = 0u;
  t.v[0] = t.v[1] = t.v[2] = 0.f;
  t.z[0] = t.z[1] = t.z[2] = 0.f;
  t.x = t.ref = t.snap = make_float4(0.f, 0.f, 0.f, 0.f);
  t.path0 = 0.f;
  if (t.live) {
    // (with a Verlet list the mass rides in ref4.w -- chx_pack_langevin puts it there -- and needs no load of its own)
    if (!(HOT || a.ref4)) t.m = a.mass[t.pp];
#pragma unroll
    for (int c = 0; c < 3; ++c) t.v[c] = a.v[3 * (size_t)t.pp + c];
    if (HOT || a.ref4) {
      t.ref = a.ref4[t.pp];
      t.m = t.ref.w;
    }
    if (DEFER && do_aoa) t.snap = a.snap4[t.pp];
    // (a reference rebuild event of the decoupled scheme copies the input positions into ref4 even in the force-only
    // launch that ends a run; so does that launch's user-order copy of the final state)
    if (do_aoa || DEFER || (!HOT && a.x3_out != nullptr)) t.x = a.x_cur[t.pp];
    if (do_aoa) {
      // the path travelled since the last pruning pass rides in x4.w; a pruning launch starts it over
      t.path0 = PRUNE ? 0.f : t.x.w;
      t.o = (HOT || a.perm) ? a.perm[t.pp] : (uint32_t)t.pp;
      if (!HOT && a.noise) {
        const float* row = a.noise + ((size_t)(step - a.noise_step0) * (size_t)a.n + t.o) * 3;
        t.z[0] = row[0], t.z[1] = row[1], t.z[2] = row[2];
      }
    }
  }
  return t;
}

// b sigma_v and the O-step noise of the particle: pure arithmetic on the operands above, placed by the
// caller where it overlaps memory latency (k_lj_step: between the issue of the first neighbor gathers and
// their use)
template <bool HOT>
__device__ __forceinline__ void finish_step_particle(const StepArgs& a, long long step, int do_aoa, StepParticle& t) {
  if (t.live && do_aoa) {
    t.bs = fmul(a.b, fsqrt(fdiv(a.kT, t.m)));
    if (HOT || !a.noise) {
      // one Philox block serves the three components of the particle
      philox_normals3(t.o, (uint32_t)step, CHX_STREAM_LANGEVIN, a.k0, a.k1, t.z[0], t.z[1], t.z[2]);
    }
  }
}

template <bool HOT, bool PRUNE>
__device__ __forceinline__ StepParticle load_step_particle(const StepArgs& a, long long step, int do_aoa, int p0,
                                                           int pl, int ppb) {
  StepParticle t = load_step_operands<HOT, PRUNE>(a, step, do_aoa, p0, pl, ppb);
  finish_step_particle<HOT>(a, step, do_aoa, t);
  return t;
}

// The integrate half of a fused step launch: one particle per thread, packed into the first PPB
// threads of the block (tk), forces from the block's LDS table sF.
//      These few wavefronts are the tail of the launch, so the common case (coordinates inside
//      the box, |x - ref| < L) runs branch-free; a particle that falls outside it is redone
//      through the general formulas, which give the same bits wherever both apply.
// Returns true when this thread's particle has moved too far for the pruned inner list.
template <bool HOT, bool DEFER = false>
__device__ __forceinline__ bool step_epilogue(const StepArgs& a, const StepParticle& tk, const float* sF, long long step,
                                              int n_kicks, int do_aoa, bool is_event = false) {
  bool stale = false;
  if (tk.live) {
    // (three explicit calls on scalars: local arrays of box lengths here ended up as a per-thread array in LDS)
    float vn0, vn1, vn2, xn0, xn1, xn2, sq0, sq1, sq2, si0, si1, si2, ss0, ss1, ss2;
    const float F0 = sF[3 * threadIdx.x], F1 = sF[3 * threadIdx.x + 1], F2 = sF[3 * threadIdx.x + 2];
    // deferred rebuilds: a launch that starts with a reference rebuild event measures the skin check of its new
    // positions from its INPUT positions -- the reference positions the rebuild has just taken
    // (chiron/integrators.py:241-242: build(x) stores ref_positions = x, chiron/neighbors.py:671)
    const bool ev = DEFER && is_event;
    const float r0 = ev ? tk.x.x : tk.ref.x, r1 = ev ? tk.x.y : tk.ref.y, r2 = ev ? tk.x.z : tk.ref.z;
    bool rare = integrate_component<true, HOT, DEFER>(a, n_kicks, do_aoa, F0, tk.m, tk.bs, tk.z[0], tk.v[0], tk.x.x, r0, tk.snap.x,
                                               a.box.lx, a.box.hx, vn0, xn0, sq0, si0, ss0);
    rare |= integrate_component<true, HOT, DEFER>(a, n_kicks, do_aoa, F1, tk.m, tk.bs, tk.z[1], tk.v[1], tk.x.y, r1, tk.snap.y,
                                           a.box.ly, a.box.hy, vn1, xn1, sq1, si1, ss1);
    rare |= integrate_component<true, HOT, DEFER>(a, n_kicks, do_aoa, F2, tk.m, tk.bs, tk.z[2], tk.v[2], tk.x.z, r2, tk.snap.z,
                                           a.box.lz, a.box.hz, vn2, xn2, sq2, si2, ss2);
    if (__builtin_expect(rare, 0)) {
      (void)integrate_component<false, HOT, DEFER>(a, n_kicks, do_aoa, F0, tk.m, tk.bs, tk.z[0], tk.v[0], tk.x.x, r0, tk.snap.x,
                                            a.box.lx, a.box.hx, vn0, xn0, sq0, si0, ss0);
      (void)integrate_component<false, HOT, DEFER>(a, n_kicks, do_aoa, F1, tk.m, tk.bs, tk.z[1], tk.v[1], tk.x.y, r1, tk.snap.y,
                                            a.box.ly, a.box.hy, vn1, xn1, sq1, si1, ss1);
      (void)integrate_component<false, HOT, DEFER>(a, n_kicks, do_aoa, F2, tk.m, tk.bs, tk.z[2], tk.v[2], tk.x.z, r2, tk.snap.z,
                                            a.box.lz, a.box.hz, vn2, xn2, sq2, si2, ss2);
    }
    const float vn[3] = {vn0, vn1, vn2}, xn[3] = {xn0, xn1, xn2}, sq[3] = {sq0, sq1, sq2}, sqi[3] = {si0, si1, si2};
    // The step's results leave as WRITE-THROUGH stores (sc1): nothing the next launch needs stays dirty in
    // this XCD's L2, so the write-back at the kernel boundary has nothing to do (the boundary costs about
    // B / 6 TB/s for B dirty bytes: 7.3 MB at N = 262144), and the next launch finds L2 cold either way.
    store_wt3(a.v + 3 * (size_t)tk.pp, vn[0], vn[1], vn[2]);
    // the reference's rebuild (its side effect on the state): reference positions := this launch's input positions
    if (ev) store_wt4(a.ref4 + tk.pp, make_float4(tk.x.x, tk.x.y, tk.x.z, tk.m));
    if (!HOT && !do_aoa && (a.x3_out != nullptr || a.v3_out != nullptr || a.ref3_out != nullptr)) {
      // the run's last launch: the state it leaves, in USER order (what chx_unpack_langevin would copy out behind it)
      const size_t o = 3 * (size_t)(a.perm ? a.perm[tk.pp] : (uint32_t)tk.pp);
      if (a.v3_out != nullptr) a.v3_out[o] = vn[0], a.v3_out[o + 1] = vn[1], a.v3_out[o + 2] = vn[2];
      if (a.x3_out != nullptr) a.x3_out[o] = tk.x.x, a.x3_out[o + 1] = tk.x.y, a.x3_out[o + 2] = tk.x.z;
      if (a.ref3_out != nullptr) {
        a.ref3_out[o] = ev ? tk.x.x : tk.ref.x;
        a.ref3_out[o + 1] = ev ? tk.x.y : tk.ref.y;
        a.ref3_out[o + 2] = ev ? tk.x.z : tk.ref.z;
      }
    }
    if (do_aoa) {
      // x4.w: path length since the last pruning pass (a sum of per-step displacements bounds the displacement)
      const float path = tk.path0 + __builtin_sqrtf((sqi[0] + sqi[1]) + sqi[2]);
      store_wt4(reinterpret_cast<float4*>(a.x_next) + tk.pp, make_float4(xn[0], xn[1], xn[2], path));
      // skin check :241: (rx^2 + ry^2) + rz^2 in the reference's order
      const bool moved = (HOT || a.ref4) && fadd(fadd(sq[0], sq[1]), sq[2]) > a.T_half_skin;
      if (!DEFER) {
        if (moved) {
          atomicMin(&a.ctrl[3], (int)(step + 1));
          // (tell the host at once: the copy this launch published when it started cannot know yet)
          if (a.host_ctrl) __hip_atomic_store(&a.host_ctrl[3], (int)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      } else {
        // deferred rebuilds: the reference's check only books its rebuild for the next launch (the word of the
        // OTHER step parity: no launch reads a word its own blocks write; every writer stores the same value) ...
        if (moved) a.ctrl[10 + (int)((step + 1) & 1)] = (int)(step + 1);
        // ... and what halts the stream is the force list running out of its guarantee (distance from ITS positions)
        const float d2s = (ss0 + ss1) + ss2;
        if (d2s > a.T_half_skin) {
          atomicMin(&a.ctrl[3], (int)(step + 1));
          if (a.host_ctrl) __hip_atomic_store(&a.host_ctrl[3], (int)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if ((HOT || a.fli) && path > a.half_inner) stale = true;
    }
  }
  return stale;
}

// PRUNE: the launch that refreshes the dynamically pruned inner list (see below): it walks the FULL
// force list and, next to the forces, copies the entries with |r| < cutoff + inner_skin into the
// inner list, in order, and restarts every particle's path length (x4.w) -- the pruning costs no pass of its own.
// `epoch` counts the refreshes since the state was loaded (-1 before the first): the two
// "inner list is stale" words hold the epoch in which they were raised, so a refresh invalidates
// them without anybody having to clear a word another block may be reading.
template <int G, bool HOT, bool PRUNE, bool DEFER>
__device__ __forceinline__ void lj_step_body(int32_t* ctrl_first, long long step, int epoch, int n_lead, unsigned nblocks,
                                             const float4* x_lead, const uint32_t* sfl, const int32_t* sdeg,
                                             const StepArgs& a, int n_kicks, int do_aoa) {
  // (Leading scalar arguments are the ones the dispatcher preloads into SGPRs: the control block, the step
  // index and everything the FIRST vector loads need -- particle count, grid size (gridDim would be a fetch
  // from the dispatch packet), the position buffer and the list this launch most probably walks (sfl, sdeg:
  // the pruned inner list in the steady state) -- so those loads can leave before any scalar fetch returns.
  // n_lead == a.n and x_lead == a.x_cur; the copies inside StepArgs arrive with the rest of the block.)
  // `a` is NEVER written (see the note at the launch-uniform switches below): ctrl_first == a.ctrl,
  // n_lead == a.n and x_lead == a.x_cur; code that runs before the argument block has arrived uses the leading copies.
  constexpr int PPB = 256 / G;  // particles per block (<= 256)
  const unsigned blk = xcd_block(blockIdx.x, nblocks);
  const int tid = blk * 256 + threadIdx.x;
  const int p = tid / G;
  const int q = tid % G;
  const bool valid = p < n_lead;
  // ---- speculative first loads of this lane's row, ahead of the control words (pinned_load*) ----
  const int pc_s = valid ? p : 0;
  const uint4* scol = reinterpret_cast<const uint4*>(sfl) + (size_t)pc_s * G + q;
  const uint4 s_e4 = pinned_load16(scol);
  const uint4 s_xi = pinned_load16(x_lead + pc_s);
  const int s_d = pinned_load4(sdeg + pc_s);
  if (HOT) n_kicks = 2, do_aoa = 1;  // what the launcher checked (launch_step_g), restated for the optimiser
  // The other launch-uniform switches are folded where they are read (`HOT || ...`).  Do NOT write them into the
  // by-value argument struct (or a copy of its box): the struct then lives in a private copy, the vectoriser reads
  // overlapping pairs of the box half-lengths from it, SROA gives up, and the backend parks the leftover array in
  // LDS indexed by the flat work-item id -- which needs the workgroup size from the DISPATCH PACKET, a scalar load
  // from host memory in every wavefront: the launch then takes 24-30 us instead of 5 (measured, round 2).
  const bool periodic = HOT || a.box.periodic != 0;
  __shared__ float sF[3 * PPB];
  // the control block as this launch finds it goes to the host-mapped copy (one wavefront, system-scope stores;
  // no-op launches publish too: that is how the host learns about a halt)
  // word 0 of the copy is this launch's step index: "launch `step` has begun", i.e. every earlier launch of the
  // stream has finished -- what chx_langevin_fused_run's launch window counts with
  if (a.host_ctrl != nullptr && blockIdx.x == 0 && threadIdx.x < CHX_CTRL_WORDS)
    __hip_atomic_store(&a.host_ctrl[threadIdx.x], threadIdx.x == 0 ? (int)step : ctrl_first[threadIdx.x], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // (naming the arguments the first vector loads need in an asm statement ahead of this branch makes
  // their fetch leave together with the halt word's -- and measured 3 % slower: the stale-word read
  // turns into a vector load and a second argument fetch still follows the branch)
  const int halt_step = ctrl_first[3];
  if ((long long)halt_step <= step) return;  // block-uniform
  // Which table this launch walks: the pruned inner list, unless some particle has moved too far
  // since the last refresh (word raised by an EARLIER launch; the two words alternate with the
  // step parity so that no launch reads a word its own blocks may be writing).
  // (Requesting both control words together ahead of the branch -- two s_load_dwordx4 in one asm
  // block, since the compiler narrows and sinks plain reads -- measured 3 % SLOWER at N = 32768.)
  // (read unconditionally and combined without a short circuit: a branch on `epoch` in front of this
  // read puts a second exposed scalar round trip ahead of every other load of the launch)
  const int stale_word = ctrl_first[4 + (int)(step & 1)];
  const bool inner = !PRUNE && (HOT || a.fli != nullptr) && ((epoch >= 0) & (stale_word != epoch));
  const uint32_t* fl_use = inner ? a.fli : a.fl;
  const int32_t* deg_use = inner ? a.degi : a.deg;
  // deferred rebuilds: did the previous launch book a reference rebuild for this step? (launch-uniform; the word of
  // this step's parity was last written by an EARLIER launch, and a stale value never equals a later step)
  const bool is_event = DEFER && ctrl_first[10 + (int)(step & 1)] == (int)step;
  const int p0 = blk * PPB;
  // ---- operands of this thread's integrate work (threads 0..PPB-1 own one particle each: the
  //      first wavefronts of the block), requested before the force loop so their latency and the
  //      Philox arithmetic hide under it ----
  StepParticle tk = load_step_operands<HOT, PRUNE, DEFER>(a, step, do_aoa, p0, threadIdx.x, PPB);

  float4 xi;
  float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
  // the row's own loads and first gathers go out together with the integrate operands; the Philox draw and
  // b sigma_v (~500 instructions in the block's first wavefront) run while the gathers are in flight
  // (with one lane per particle every thread integrates and the launch is throughput-bound: there the
  // arithmetic stays in front of the row loads, measured 2-3 % faster)
  if (G == 1) finish_step_particle<HOT>(a, step, do_aoa, tk);
  // the speculated list is the one to walk (launch-uniform; the steady state): adopt what is in flight.
  // Otherwise (inner list gone stale since the host chose) the row is loaded again from the right table.
  RowLoads rl = (fl_use == sfl) ? lj_row_adopt<G>(a.x_cur, scol, pc_s, s_xi, s_d, s_e4)
                                : lj_row_begin<G>(a.x_cur, fl_use, deg_use, a.n, p, q, valid);
  if (G > 1) finish_step_particle<HOT>(a, step, do_aoa, tk);
  if (PRUNE) {
    constexpr int W = 4 * G;
    int kept = 0;
    lj_row_finish<G, true>(rl, a.x_cur, fl_use, a.n, q, valid, a.box, periodic, a.lj, xi, fx, fy, fz, en,
                           PruneOut{a.fli, a.T_inner}, &kept);
    if (valid) {
      // the force kernel gathers whole chunks (chunk 0 speculatively): pad the last touched chunk
      int end = (kept + W - 1) / W * W;
      end = end < W ? W : end;
      for (int sl = kept + q; sl < end; sl += G) a.fli[chx_fl_addr(p, sl, a.n, G)] = (uint32_t)p;
      if (q == 0) a.degi[p] = kept;
    }
  } else {
    lj_row_finish<G>(rl, a.x_cur, fl_use, a.n, q, valid, a.box, periodic, a.lj, xi, fx, fy, fz, en);
  }
  fx = group_sum<G>(fx), fy = group_sum<G>(fy), fz = group_sum<G>(fz);
  if (q == 0) {
    const int pl = threadIdx.x / G;
    sF[3 * pl] = fx;
    sF[3 * pl + 1] = fy;
    sF[3 * pl + 2] = fz;
  }
  __syncthreads();
  // ---- integrate: one particle per thread, packed into the first PPB threads of the block ----
  const bool stale = step_epilogue<HOT, DEFER>(a, tk, sF, step, n_kicks, do_aoa, is_event);
  // the inner list stays off until the next refresh (a new epoch) outdates the word
  // (agent-scope stores = write-through: a steady-state launch dispatched through the step queue ends without a
  // release fence, chx_aql.h, so nothing another XCD must see may stay behind as a dirty line of this XCD's L2)
  if (do_aoa && (HOT || a.fli) && epoch >= 0 && (stale || (!inner && !PRUNE && threadIdx.x == 0)))
    __hip_atomic_store(&ctrl_first[4 + (int)((step + 1) & 1)], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (is_event) {  // the reference's rebuild count and the step it happened at (chiron/integrators.py:242)
      const int c = ctrl_first[8] + 1;
      ctrl_first[8] = c;
      ctrl_first[9] = (int)step;
      if (a.ev_log != nullptr && a.ev_cap > 0) a.ev_log[(c - 1) % a.ev_cap] = (int)step;
      if (a.host_ctrl != nullptr) {
        __hip_atomic_store(&a.host_ctrl[8], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.host_ctrl[9], (int)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // (the counters go to the host copy as they change: the copy made at launch start cannot hold this launch's own)
    const int w = PRUNE ? 7 : 6;
    if (PRUNE || ((HOT || a.fli) && !inner)) {  // refreshes of the inner list / launches on the full list
      const int c = ctrl_first[w] + 1;
      __hip_atomic_store(&ctrl_first[w], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.host_ctrl != nullptr) __hip_atomic_store(&a.host_ctrl[w], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// One replica, one launch per step: the arguments the first loads need lead the block (see lj_step_body).
template <int G, bool HOT, bool PRUNE, bool DEFER>
__global__ __launch_bounds__(256) void k_lj_step(int32_t* ctrl_first, long long step, int epoch, int n_lead,
                                                 unsigned nblocks, const float4* x_lead, const uint32_t* sfl,
                                                 const int32_t* sdeg, StepArgs a, int n_kicks, int do_aoa) {
  lj_step_body<G, HOT, PRUNE, DEFER>(ctrl_first, step, epoch, n_lead, nblocks, x_lead, sfl, sdeg, a, n_kicks, do_aoa);
}

// SEVERAL replicas, one launch per step (include/chx.h: chx_langevin_batch_run; chiron/multistate.py:497-510 is a serial
// loop over replicas).  A step of an N = 32 768 replica is a latency chain that fills a quarter of the card; eight of
// them in one grid are a throughput problem the size of N = 262 144.  blockIdx.y picks the replica's item from the
// argument block (a scalar load with a uniform offset: the block is never copied), blockIdx.x is the replica's block;
// everything else is lj_step_body -- the same instructions in the same order as the one-replica launch, so a replica's
// trajectory does not depend on whether it is stepped alone or in a batch.  The items of one packet share the HOT
// switch (steady-state launch or not); whether an item refreshes its inner list is a per-item flag.  A step whose replicas
// fall into several classes -- a first step here, steady state there -- goes out as one packet per class, the later ones
// without the barrier bit.
// Each replica has its own control block: a replica whose skin check has fired turns into a no-op (its halt word) while
// the others carry on.
template <int G, bool HOT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HOT ? 7 : 4))) void k_lj_step_batch(BatchArgs b) {
  const BatchItem& it = b.item[blockIdx.y];
  if (blockIdx.x >= it.nblocks) return;  // (replicas of different sizes share the grid's x extent)
  // (block-uniform: a replica that refreshes its inner list in this step takes the pruning instantiation, the others the
  // plain one -- in the same grid, so that the refresh of one replica in eight does not cost a dispatch of its own)
  if (it.prune)
    lj_step_body<G, HOT, true, false>(it.a.ctrl, it.step, it.epoch, it.a.n, it.nblocks, it.a.x_cur, it.sfl, it.sdeg, it.a, it.n_kicks,
                                      it.do_aoa);
  else
    lj_step_body<G, HOT, false, false>(it.a.ctrl, it.step, it.epoch, it.a.n, it.nblocks, it.a.x_cur, it.sfl, it.sdeg, it.a, it.n_kicks,
                                       it.do_aoa);
}

// ---------------------------------------------------------------------------
// Dynamic pruning: the reference's list keeps every pair within cutoff + skin of the last build
// (skin = 0.5 nm in its examples: 3.3x the volume of the cutoff sphere), but a pair can only
// interact while |r| < cutoff.  Every few steps a PRUNE launch of k_lj_step copies, in order, the
// entries with |r_now| < cutoff + inner_skin into a second table of the same layout and restarts the
// path length every particle travels from then on (x4.w: a running sum of per-step displacements, which
// bounds the displacement); the following launches walk that table for as long as no particle's path has
// grown beyond inner_skin / 2 (checked next to the reference's own skin check) and fall back to the
// full table otherwise, until the next refresh.  A dropped pair was farther than cutoff + inner_skin
// and each partner has moved less than inner_skin / 2, so it is still outside the cutoff: the set of
// interacting pairs -- the reference's set -- is unchanged; only the summation order differs.
// ---------------------------------------------------------------------------
// all-pairs kernels: 16 lanes per particle, j-tiles of 256 particles staged through LDS
#define AP_SLICES 16
#define AP_TILE 256

// ---------------------------------------------------------------------------
// Fused MD step without a neighbor list: the all-pairs force of k_lj_allpairs (nbr_list=None: no
// PBC, chiron/potential.py:235-258; PairListNsqrd: PBC, chiron/neighbors.py:1106-1216) followed, in
// the same launch, by the BAOAB update that consumes it (step_epilogue) -- one launch per step
// instead of an integrate launch plus a force launch.  16 lanes per particle, 16 particles per
// block, j-tiles of 256 particles staged through LDS straight from the float4 position buffer.
// These paths have no skin check (PairListNsqrd.check only looks at N), so nothing ever halts.
// ---------------------------------------------------------------------------
// SLICES lanes share one particle: 64 (a whole wavefront) for the small systems this path is meant
// for -- at N = 512 a lane then evaluates 8 pairs instead of 32, and the launch is a dependent-issue
// chain, not a throughput problem -- and 16 beyond.
// DIRECT (the small-system instantiation, SLICES = 64): no LDS tiles.  A lane's candidates j = sl, sl + 64, ... are
// read straight from the position buffer, eight coalesced loads in flight per lane -- at N = 512 that is the whole row
// behind ONE exposed round trip, where the tiled form pays one per tile plus two barriers each; same pairs in the same
// order in every lane, so the same bits.
template <int SLICES, bool DIRECT>
__global__ __launch_bounds__(256) void k_ap_step(StepArgs a, long long step, int n_kicks, int do_aoa, int use_cutoff) {
  constexpr int PPB = 256 / SLICES;
  __shared__ float4 tile[DIRECT ? 1 : AP_TILE];
  __shared__ float sF[3 * PPB];
  const int p0 = blockIdx.x * PPB;
  const StepParticle tk = load_step_particle<false, false>(a, step, do_aoa, p0, threadIdx.x, PPB);
  const int il = threadIdx.x / SLICES;
  const int sl = threadIdx.x % SLICES;
  const int i = p0 + il;
  const bool valid = i < a.n;
  const float4 xi = a.x_cur[valid ? i : 0];
  float fx = 0.f, fy = 0.f, fz = 0.f;
  auto pair = [&](int j, const float4& xj) {
    const bool hi = j > i;
    float rx, ry, rz;
    bool bad;  // branch-free minimum image, general formula for |d| >= L (same bits where both apply)
    displacement_oriented_fast(a.box, hi, xi.x, xi.y, xi.z, xj.x, xj.y, xj.z, rx, ry, rz, bad);
    if (__builtin_expect(bad, 0)) displacement_oriented(a.box, hi, xi.x, xi.y, xi.z, xj.x, xj.y, xj.z, rx, ry, rz);
    const float r2 = norm2(rx, ry, rz);
    if (!use_cutoff || r2 <= a.lj.T) {
      float e, fs;
      lj_pair(r2, a.lj.sig2, a.lj.eps4, a.lj.eps24, e, fs);
      if (!hi) fs = -fs;
      fx += fs * rx;
      fy += fs * ry;
      fz += fs * rz;
    }
  };
  if (DIRECT) {
    constexpr int B = 8;  // loads in flight per lane
    for (int j0 = sl; j0 < a.n; j0 += B * SLICES) {
      float4 xj[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const int j = j0 + k * SLICES;
        xj[k] = a.x_cur[j < a.n ? j : 0];
      }
      if (valid) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
          const int j = j0 + k * SLICES;
          if (j < a.n && j != i) pair(j, xj[k]);
        }
      }
    }
  } else {
    for (int j0 = 0; j0 < a.n; j0 += AP_TILE) {
      __syncthreads();
      const int jj = j0 + threadIdx.x;
      if (jj < a.n) tile[threadIdx.x] = a.x_cur[jj];
      __syncthreads();
      const int cnt = (a.n - j0) < AP_TILE ? (a.n - j0) : AP_TILE;
      if (valid) {
        for (int u = sl; u < cnt; u += SLICES) {
          const int j = j0 + u;
          if (j == i) continue;
          pair(j, tile[u]);
        }
      }
    }
  }
  // (DPP swizzles inside a row of 16 lanes, shuffles across rows: the force-list kernels' reduction)
  fx = group_sum<SLICES>(fx), fy = group_sum<SLICES>(fy), fz = group_sum<SLICES>(fz);
  if (sl == 0) {
    sF[3 * il] = fx;
    sF[3 * il + 1] = fy;
    sF[3 * il + 2] = fz;
  }
  __syncthreads();
  (void)step_epilogue<false>(a, tk, sF, step, n_kicks, do_aoa);
}

template <int G>
static int launch_lj_fl_g(const float4* x4, const uint32_t* fl, const int32_t* deg, int n, int chunks,
                          Box3 box, LJParams lj, float* f3, float* energy, double* ws, int32_t* ctrl,
                          hipStream_t st) {
  const unsigned blocks = (unsigned)(((int64_t)n * G + 255) / 256);
  if (f3 && energy) {
    hipLaunchKernelGGL((k_lj_fl<G, true, true>), dim3(blocks), dim3(256), 0, st, x4, fl, deg, n, chunks,
                       box, lj, f3, ws, ctrl);
  } else if (f3) {
    hipLaunchKernelGGL((k_lj_fl<G, true, false>), dim3(blocks), dim3(256), 0, st, x4, fl, deg, n, chunks,
                       box, lj, f3, ws, ctrl);
  } else {
    hipLaunchKernelGGL((k_lj_fl<G, false, true>), dim3(blocks), dim3(256), 0, st, x4, fl, deg, n, chunks,
                       box, lj, f3, ws, ctrl);
  }
  if (energy) {
    hipLaunchKernelGGL(k_sum_ws, dim3(1), dim3(256), 0, st, ws, (int)blocks, energy);
  }
  return CHX_OK;
}

LJParams chx_make_lj(float sigma, float epsilon, float cutoff) {
  LJParams lj;
  lj.T = chx_cutoff_threshold_h(cutoff);
  lj.sig2 = sigma * sigma;
  lj.eps4 = 4.0f * epsilon;
  lj.eps24 = 24.0f * epsilon;
  return lj;
}

int chx_launch_lj_fl(const float* x4, const uint32_t* fl, const int32_t* deg, int n, int g, int chunks,
                     chx_box_t box, float sigma, float epsilon, float cutoff, float* f3,
                     float* energy, double* ws, int32_t* ctrl, hipStream_t st) {
  const float4* xx = reinterpret_cast<const float4*>(x4);
  const Box3 b3 = make_box3(box);
  const LJParams lj = chx_make_lj(sigma, epsilon, cutoff);
  switch (g) {
    case 1: return launch_lj_fl_g<1>(xx, fl, deg, n, chunks, b3, lj, f3, energy, ws, ctrl, st);
    case 2: return launch_lj_fl_g<2>(xx, fl, deg, n, chunks, b3, lj, f3, energy, ws, ctrl, st);
    case 4: return launch_lj_fl_g<4>(xx, fl, deg, n, chunks, b3, lj, f3, energy, ws, ctrl, st);
    case 8: return launch_lj_fl_g<8>(xx, fl, deg, n, chunks, b3, lj, f3, energy, ws, ctrl, st);
    case 16: return launch_lj_fl_g<16>(xx, fl, deg, n, chunks, b3, lj, f3, energy, ws, ctrl, st);
    case 32: return launch_lj_fl_g<32>(xx, fl, deg, n, chunks, b3, lj, f3, energy, ws, ctrl, st);
    case 64: return launch_lj_fl_g<64>(xx, fl, deg, n, chunks, b3, lj, f3, energy, ws, ctrl, st);
  }
  chx_set_error("chx_launch_lj_fl: unsupported g = %d", g);
  return CHX_ERR_INVALID;
}

// the step kernel's kernarg segment as the compiler lays it out (checked against the code object's metadata when the
// step queue looks the kernel up: chx_aql_dispatch refuses a block shorter than the kernel's)
struct StepKernarg {
  int32_t* ctrl_first;
  long long step;
  int epoch;
  int n_lead;
  unsigned nblocks;
  const float4* x_lead;
  const uint32_t* sfl;
  const int32_t* sdeg;
  StepArgs a;
  int n_kicks;
  int do_aoa;
};
struct ApKernarg {
  StepArgs a;
  long long step;
  int n_kicks;
  int do_aoa;
  int use_cutoff;
};

template <int G>
static int launch_step_g(const StepArgs& a, long long step, int n_kicks, int do_aoa, bool prune, int epoch,
                         hipStream_t st) {
  const unsigned blocks = (unsigned)(((int64_t)a.n * G + 255) / 256);
  static const bool dbg = getenv("CHX_DEBUG_SYNC") != nullptr;  // serialise + narrate (fault hunting)
  if (dbg) fprintf(stderr, "[chx] k_lj_step<%d> step %lld kicks %d aoa %d prune %d epoch %d\n", G, step, n_kicks, do_aoa, (int)prune, epoch);
  // steady-state launches take the instantiation with the launch-uniform switches folded
  const bool hot = n_kicks == 2 && do_aoa && a.wrap && a.box.periodic && a.ref4 && a.fli && a.perm && !a.noise;
  // the list the launch will most probably walk (the kernel checks: see its speculative loads)
  const bool spec_inner = a.fli != nullptr && !prune && epoch >= 0;
  const uint32_t* sfl = spec_inner ? a.fli : a.fl;
  const int32_t* sdeg = spec_inner ? a.degi : a.deg;
  // DEFER (deferred rebuilds, chx_langevin_t::snap4) is a template parameter, not a run-time switch: compiled into the
  // one kernel, its extra registers and branches cost the halting scheme's steady-state launch 2.5 % (measured, round 3)
  const bool defer = a.snap4 != nullptr;
  if (chx_aql_on()) {
    // the step queue (chx_aql.h): the same kernel, dispatched as an AQL packet.  A steady-state launch leaves nothing
    // but write-through stores behind (step_epilogue; the control words are stored at agent scope) and carries no
    // release fence; a launch that refreshes the inner list or takes the general paths ends with an agent-scope release.
    // (looked up once per instantiation; several threads may step replicas of their own: one atomic word per entry, the
    // look-up itself is idempotent)
    static const AqlKernel absent{};
    static std::atomic<const AqlKernel*> table[2][2][2];
    const int ih = hot ? 1 : 0, ip = prune ? 1 : 0, id = defer ? 1 : 0;
    const AqlKernel* k = table[ih][ip][id].load(std::memory_order_acquire);
    if (k == nullptr) {
      char name[96];
      snprintf(name, sizeof(name), "void k_lj_step<%d, %s, %s, %s>", G, hot ? "true" : "false", prune ? "true" : "false",
               defer ? "true" : "false");
      k = chx_aql_kernel(name);
      if (k == nullptr) k = &absent;
      table[ih][ip][id].store(k, std::memory_order_release);
    }
    if (k == &absent) k = nullptr;
    if (k == nullptr) {  // (a code object from other sources than this library: refuse, do not mix launch paths)
      chx_set_error("chx_launch_step: libchx_step.hsaco holds no k_lj_step<%d, %d, %d, %d>", G, ih, ip, id);
      return CHX_ERR_LAUNCH;
    }
    {
      StepKernarg ka;
      std::memset(&ka, 0, sizeof(ka));
      ka.ctrl_first = a.ctrl, ka.step = step, ka.epoch = epoch, ka.n_lead = a.n, ka.nblocks = blocks;
      ka.x_lead = a.x_cur, ka.sfl = sfl, ka.sdeg = sdeg, ka.a = a, ka.n_kicks = n_kicks, ka.do_aoa = do_aoa;
      return chx_aql_dispatch(st, k, blocks, &ka, sizeof(ka), !(hot && !prune && !defer), a.x3_out != nullptr);
    }
  }
#define CHX_LAUNCH_STEP(HOT_, PRUNE_, DEFER_)                                                                             \
  hipLaunchKernelGGL((k_lj_step<G, HOT_, PRUNE_, DEFER_>), dim3(blocks), dim3(256), 0, st, a.ctrl, step, epoch, a.n, blocks, \
                     a.x_cur, sfl, sdeg, a, n_kicks, do_aoa)
  if (defer) {
    if (hot && prune) CHX_LAUNCH_STEP(true, true, true);
    else if (hot) CHX_LAUNCH_STEP(true, false, true);
    else if (prune) CHX_LAUNCH_STEP(false, true, true);
    else CHX_LAUNCH_STEP(false, false, true);
  } else {
    if (hot && prune) CHX_LAUNCH_STEP(true, true, false);
    else if (hot) CHX_LAUNCH_STEP(true, false, false);
    else if (prune) CHX_LAUNCH_STEP(false, true, false);
    else CHX_LAUNCH_STEP(false, false, false);
  }
#undef CHX_LAUNCH_STEP
  if (dbg) (void)hipStreamSynchronize(st);
  return CHX_OK;
}

// one fused step launch on the force list (the host loop lives in chx_launch_window.hip)
int chx_launch_step(const StepArgs& a, int g, long long step, int n_kicks, int do_aoa, bool prune, int epoch, hipStream_t st) {
  switch (g) {
    case 1: return launch_step_g<1>(a, step, n_kicks, do_aoa, prune, epoch, st);
    case 2: return launch_step_g<2>(a, step, n_kicks, do_aoa, prune, epoch, st);
    case 4: return launch_step_g<4>(a, step, n_kicks, do_aoa, prune, epoch, st);
    case 8: return launch_step_g<8>(a, step, n_kicks, do_aoa, prune, epoch, st);
    default:
      // (round 6: the fused step is instantiated for the lane counts a list builder ever chooses -- the pair-force kernel
      // alone takes up to 64)
      chx_set_error("chx_launch_step: unsupported g = %d (the fused step takes 1, 2, 4 or 8 lanes per particle)", g);
      return CHX_ERR_UNSUPPORTED;
  }
}

// one batched packet (chx_step.h)
template <int G>
static int launch_step_batch_g(const BatchItem* items, int count, bool hot, bool beside, hipStream_t st) {
  alignas(16) static thread_local BatchArgs ba;
  unsigned maxb = 0;
  bool any_prune = false, any_final = false;
  for (int i = 0; i < count; ++i) {
    ba.item[i] = items[i];
    maxb = items[i].nblocks > maxb ? items[i].nblocks : maxb;
    any_prune |= items[i].prune != 0;
    any_final |= items[i].a.x3_out != nullptr || items[i].a.v3_out != nullptr;
  }
  for (int i = count; i < CHX_BATCH_MAX; ++i) ba.item[i].nblocks = 0;  // (never addressed: the grid's y extent is `count`)
  static const bool dbg = getenv("CHX_DEBUG_SYNC") != nullptr;
  if (dbg) fprintf(stderr, "[chx] k_lj_step_batch<%d, %d> x %d items, first step %lld\n", G, (int)hot, count, items[0].step);
  if (chx_aql_on()) {
    static const AqlKernel absent{};
    static std::atomic<const AqlKernel*> table[2];
    const int ih = hot ? 1 : 0;
    const AqlKernel* k = table[ih].load(std::memory_order_acquire);
    if (k == nullptr) {
      char name[96];
      snprintf(name, sizeof(name), "void k_lj_step_batch<%d, %s>", G, hot ? "true" : "false");
      k = chx_aql_kernel(name);
      if (k == nullptr) k = &absent;
      table[ih].store(k, std::memory_order_release);
    }
    if (k == &absent) {
      chx_set_error("chx_launch_step_batch: libchx_step.hsaco holds no k_lj_step_batch<%d, %d>", G, ih);
      return CHX_ERR_LAUNCH;
    }
    // (fences as for the one-replica launch: a packet of steady-state items that refresh nothing leaves write-through
    // stores only)
    return chx_aql_dispatch_grid(st, k, dim3(maxb, (unsigned)count), &ba, sizeof(ba), !(hot && !any_prune), beside, any_final);
  }
  if (hot) hipLaunchKernelGGL((k_lj_step_batch<G, true>), dim3(maxb, (unsigned)count), dim3(256), 0, st, ba);
  else hipLaunchKernelGGL((k_lj_step_batch<G, false>), dim3(maxb, (unsigned)count), dim3(256), 0, st, ba);
  if (dbg) (void)hipStreamSynchronize(st);
  return CHX_OK;
}

int chx_launch_step_batch(const BatchItem* items, int count, int g, bool hot, bool beside_previous, hipStream_t st) {
  if (count < 1 || count > CHX_BATCH_MAX) {
    chx_set_error("chx_launch_step_batch: %d items (1 .. %d)", count, CHX_BATCH_MAX);
    return CHX_ERR_INVALID;
  }
  switch (g) {
    case 1: return launch_step_batch_g<1>(items, count, hot, beside_previous, st);
    case 2: return launch_step_batch_g<2>(items, count, hot, beside_previous, st);
    case 4: return launch_step_batch_g<4>(items, count, hot, beside_previous, st);
    case 8: return launch_step_batch_g<8>(items, count, hot, beside_previous, st);
    default:
      chx_set_error("chx_launch_step_batch: unsupported g = %d (1, 2, 4, 8)", g);
      return CHX_ERR_UNSUPPORTED;
  }
}

// one fused step launch without a neighbor list: all-pairs force + BAOAB (k_ap_step)
void chx_launch_ap_step(const StepArgs& a, long long step, int n_kicks, int do_aoa, int use_cutoff, hipStream_t st) {
  if (chx_aql_on()) {
    static const AqlKernel* small = chx_aql_kernel("void k_ap_step<64, true>");
    static const AqlKernel* large = chx_aql_kernel("void k_ap_step<16, false>");
    const AqlKernel* k = a.n <= 2048 ? small : large;
    if (k == nullptr) chx_aql_join(st);  // (cannot happen with a code object built from this tree; keep the order at least)
    if (k != nullptr) {
      ApKernarg ka;
      std::memset(&ka, 0, sizeof(ka));
      ka.a = a, ka.step = step, ka.n_kicks = n_kicks, ka.do_aoa = do_aoa, ka.use_cutoff = use_cutoff;
      const unsigned blocks = a.n <= 2048 ? (unsigned)((a.n + 3) / 4) : (unsigned)((a.n + 15) / 16);
      // (an A-O-A launch of this kernel leaves write-through stores only -- positions and velocities, step_epilogue; no
      // control words on the paths without a Verlet list -- so it needs no release fence; the force-only launch that ends a
      // run also writes the state in user order with plain stores)
      // (a failed dispatch is reported by the entry point's chx_check_launch; never fall back to a HIP launch beside
      // packets that are still outstanding)
      chx_note_launch_failure(chx_aql_dispatch(st, k, blocks, &ka, sizeof(ka), do_aoa == 0));
      return;
    }
  }
  if (a.n <= 2048) {
    hipLaunchKernelGGL((k_ap_step<64, true>), dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, st, a, step, n_kicks, do_aoa, use_cutoff);
  } else {
    hipLaunchKernelGGL((k_ap_step<16, false>), dim3((unsigned)((a.n + 15) / 16)), dim3(256), 0, st, a, step, n_kicks, do_aoa, use_cutoff);
  }
}

extern "C" int chx_lj_fl_energy_force(const float* x4, const uint32_t* fl, const int32_t* deg,
                                      int32_t n, int32_t g, int32_t chunks, chx_box_t box, float sigma,
                                      float epsilon, float cutoff, float* force3_out,
                                      float* energy_out, double* ws, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  CHX_REQUIRE(force3_out || energy_out, "nothing to compute");
  CHX_REQUIRE(!energy_out || ws, "energy needs a workspace");
  if (n == 0) {
    if (energy_out) (void)hipMemsetAsync(energy_out, 0, sizeof(float), chx_stream(stream));
    return CHX_OK;
  }
  CHX_REQUIRE(x4 && fl && deg, "null pointer");
  CHX_REQUIRE(chunks >= 1, "chunks < 1");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  int rc = chx_launch_lj_fl(x4, fl, deg, n, g, chunks, box, sigma, epsilon, cutoff, force3_out,
                            energy_out, ws, nullptr, chx_stream(stream));
  if (rc != CHX_OK) return rc;
  return chx_check_launch("chx_lj_fl_energy_force");
}

// ---------------------------------------------------------------------------
// Direct evaluation on the PUBLIC half list (chiron/potential.py:260-279 on the arrays of
// chiron/neighbors.py:617-626), for callers that hold only (neighbor_list, n_neighbors):
// one wavefront per row, the row's own force reduced in registers, the reaction scattered
// with float atomics (summation order of the forces is therefore not reproducible; the
// energy is).  The hot loops never use this: they run on the force list above.
// ---------------------------------------------------------------------------
template <bool WANT_F, bool WANT_E>
__global__ __launch_bounds__(256) void k_lj_nl(const float* __restrict__ x, const uint32_t* __restrict__ nl,
                                               const int32_t* __restrict__ n_nbr, int n, int m, Box3 box,
                                               LJParams lj, float* __restrict__ f3, double* __restrict__ ws) {
  const int lane = lane_id();
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 6;
  float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
  if (i < n) {
    int cnt = n_nbr[i];
    cnt = cnt < m ? cnt : m;
    const float xi = x[3 * (size_t)i], yi = x[3 * (size_t)i + 1], zi = x[3 * (size_t)i + 2];
    for (int k = lane; k < cnt; k += CHX_WAVE) {
      const uint32_t j = nl[(size_t)i * m + k];
      float rx, ry, rz;  // r_ij = x_i - x_j with i < j, as chiron/neighbors.py:777-779
      displacement(box, xi, yi, zi, x[3 * (size_t)j], x[3 * (size_t)j + 1], x[3 * (size_t)j + 2], rx, ry, rz);
      const float r2 = norm2(rx, ry, rz);
      if (r2 <= lj.T) {
        float e, fs;
        lj_pair(r2, lj.sig2, lj.eps4, lj.eps24, e, fs);
        en += e;
        if (WANT_F) {
          fx = __builtin_fmaf(fs, rx, fx);
          fy = __builtin_fmaf(fs, ry, fy);
          fz = __builtin_fmaf(fs, rz, fz);
          atomicAdd(&f3[3 * (size_t)j], -fs * rx);
          atomicAdd(&f3[3 * (size_t)j + 1], -fs * ry);
          atomicAdd(&f3[3 * (size_t)j + 2], -fs * rz);
        }
      }
    }
    if (WANT_F) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        fx += __shfl_xor(fx, o, CHX_WAVE);
        fy += __shfl_xor(fy, o, CHX_WAVE);
        fz += __shfl_xor(fz, o, CHX_WAVE);
      }
      if (lane == 0 && cnt > 0) {
        atomicAdd(&f3[3 * (size_t)i], fx);
        atomicAdd(&f3[3 * (size_t)i + 1], fy);
        atomicAdd(&f3[3 * (size_t)i + 2], fz);
      }
    }
  }
  if (WANT_E) {
    double mine = block_sum<256>((double)en);
    if (threadIdx.x == 0) ws[blockIdx.x] = mine;
  }
}

extern "C" int chx_lj_nl_energy_force(const float* xyz, const uint32_t* nl, const int32_t* n_nbr,
                                      int32_t n, int32_t m, chx_box_t box, float sigma, float epsilon,
                                      float cutoff, float* energy_out, float* force_out, double* ws,
                                      chx_stream_t stream) {
  CHX_REQUIRE(n >= 0 && m >= 1, "need n >= 0, m >= 1");
  CHX_REQUIRE(force_out || energy_out, "nothing to compute");
  CHX_REQUIRE(!energy_out || ws, "energy needs a workspace");
  hipStream_t st = chx_stream(stream);
  if (n == 0) {
    if (energy_out) (void)hipMemsetAsync(energy_out, 0, sizeof(float), st);
    return CHX_OK;
  }
  CHX_REQUIRE(xyz && nl && n_nbr, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  if (force_out && hipMemsetAsync(force_out, 0, sizeof(float) * 3 * (size_t)n, st) != hipSuccess) {
    chx_set_error("chx_lj_nl_energy_force: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  const Box3 b3 = make_box3(box);
  const LJParams lj = chx_make_lj(sigma, epsilon, cutoff);
  const unsigned blocks = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);  // <= chx_lj_ws_count(n)
  if (force_out && energy_out) {
    hipLaunchKernelGGL((k_lj_nl<true, true>), dim3(blocks), dim3(256), 0, st, xyz, nl, n_nbr, n, m, b3, lj, force_out, ws);
  } else if (force_out) {
    hipLaunchKernelGGL((k_lj_nl<true, false>), dim3(blocks), dim3(256), 0, st, xyz, nl, n_nbr, n, m, b3, lj, force_out, ws);
  } else {
    hipLaunchKernelGGL((k_lj_nl<false, true>), dim3(blocks), dim3(256), 0, st, xyz, nl, n_nbr, n, m, b3, lj, force_out, ws);
  }
  if (energy_out) hipLaunchKernelGGL(k_sum_ws, dim3(1), dim3(256), 0, st, ws, (int)blocks, energy_out);
  return chx_check_launch("chx_lj_nl_energy_force");
}

// ---------------------------------------------------------------------------
// all-pairs kernel: 16 lanes per particle (4 particles per wavefront, 16 per
// block), j-tiles of 256 particles staged through LDS
// ---------------------------------------------------------------------------

template <bool WANT_F, bool WANT_E>
__global__ __launch_bounds__(256) void k_lj_allpairs(const float* __restrict__ x, int stride, int n,
                                                     Box3 box, LJParams lj, int use_cutoff,
                                                     float* __restrict__ f3, double* __restrict__ ws,
                                                     int32_t* ctrl) {
  if (ctrl != nullptr && ctrl[0] != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl[1] = 1;
    return;
  }
  __shared__ float4 tile[AP_TILE];
  const int il = threadIdx.x / AP_SLICES;
  const int sl = threadIdx.x % AP_SLICES;
  const int i = blockIdx.x * (256 / AP_SLICES) + il;
  const bool valid = i < n;
  float xi = 0.f, yi = 0.f, zi = 0.f;
  if (valid) {
    xi = x[(size_t)stride * i];
    yi = x[(size_t)stride * i + 1];
    zi = x[(size_t)stride * i + 2];
  }
  float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
  for (int j0 = 0; j0 < n; j0 += AP_TILE) {
    __syncthreads();
    const int jj = j0 + threadIdx.x;
    if (jj < n) {
      tile[threadIdx.x] = make_float4(x[(size_t)stride * jj], x[(size_t)stride * jj + 1],
                                      x[(size_t)stride * jj + 2], 0.f);
    }
    __syncthreads();
    const int cnt = (n - j0) < AP_TILE ? (n - j0) : AP_TILE;
    if (valid) {
      for (int u = sl; u < cnt; u += AP_SLICES) {
        const int j = j0 + u;
        if (j == i) continue;
        const float4 xj = tile[u];
        const bool hi = j > i;
        float rx, ry, rz;
        displacement_oriented(box, hi, xi, yi, zi, xj.x, xj.y, xj.z, rx, ry, rz);
        const float r2 = norm2(rx, ry, rz);
        if (!use_cutoff || r2 <= lj.T) {
          float e, fs;
          lj_pair(r2, lj.sig2, lj.eps4, lj.eps24, e, fs);
          if (!hi) fs = -fs;
          fx += fs * rx;
          fy += fs * ry;
          fz += fs * rz;
          en += e;
        }
      }
    }
  }
#pragma unroll
  for (int o = AP_SLICES / 2; o > 0; o >>= 1) {
    if (WANT_F) {
      fx += __shfl_xor(fx, o, CHX_WAVE);
      fy += __shfl_xor(fy, o, CHX_WAVE);
      fz += __shfl_xor(fz, o, CHX_WAVE);
    }
    if (WANT_E) en += __shfl_xor(en, o, CHX_WAVE);
  }
  if (WANT_F && valid && sl == 0) {
    f3[3 * (size_t)i] = fx;
    f3[3 * (size_t)i + 1] = fy;
    f3[3 * (size_t)i + 2] = fz;
  }
  if (WANT_E) {
    double mine = (valid && sl == 0) ? 0.5 * (double)en : 0.0;
    mine = block_sum<256>(mine);
    if (threadIdx.x == 0) ws[blockIdx.x] = mine;
  }
}

int chx_launch_lj_allpairs(const float* x, int stride, int n, chx_box_t box, float sigma,
                           float epsilon, float cutoff, float* f3, float* energy, double* ws,
                           int32_t* ctrl, hipStream_t st) {
  const Box3 b3 = make_box3(box);
  const int use_cutoff = cutoff >= 0.0f ? 1 : 0;
  const LJParams lj = chx_make_lj(sigma, epsilon, use_cutoff ? cutoff : 1.0f);
  const unsigned blocks = (unsigned)((n + 15) / 16);
  if (f3 && energy) {
    hipLaunchKernelGGL((k_lj_allpairs<true, true>), dim3(blocks), dim3(256), 0, st, x, stride, n, b3,
                       lj, use_cutoff, f3, ws, ctrl);
  } else if (f3) {
    hipLaunchKernelGGL((k_lj_allpairs<true, false>), dim3(blocks), dim3(256), 0, st, x, stride, n,
                       b3, lj, use_cutoff, f3, ws, ctrl);
  } else {
    hipLaunchKernelGGL((k_lj_allpairs<false, true>), dim3(blocks), dim3(256), 0, st, x, stride, n,
                       b3, lj, use_cutoff, f3, ws, ctrl);
  }
  if (energy) hipLaunchKernelGGL(k_sum_ws, dim3(1), dim3(256), 0, st, ws, (int)blocks, energy);
  return CHX_OK;
}

extern "C" int chx_lj_allpairs_energy_force(const float* xyz, int32_t n, chx_box_t box, float sigma,
                                            float epsilon, float cutoff, float* force_out,
                                            float* energy_out, double* ws, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  CHX_REQUIRE(force_out || energy_out, "nothing to compute");
  CHX_REQUIRE(!energy_out || ws, "energy needs a workspace");
  if (n == 0) {
    if (energy_out) (void)hipMemsetAsync(energy_out, 0, sizeof(float), chx_stream(stream));
    return CHX_OK;
  }
  CHX_REQUIRE(xyz, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  chx_launch_lj_allpairs(xyz, 3, n, box, sigma, epsilon, cutoff, force_out, energy_out, ws, nullptr,
                         chx_stream(stream));
  return chx_check_launch("chx_lj_allpairs_energy_force");
}
