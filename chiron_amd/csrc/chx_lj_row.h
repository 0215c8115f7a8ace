// The row loop of the list kernels: one particle's share of its force-list row (G lanes per particle, packed fp32 pair
// arithmetic, branch-free minimum image, optional pruning pass), shared by the pair-force / fused-step kernels
// (chx_lj.hip) and the fused Monte Carlo trial kernel (chx_mc.hip).  Internal, device code only.
#pragma once
#include "chx_common.h"
#include "chx_internal.h"
#include "chx_step.h"

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one).  Give every
// XCD a contiguous range of logical blocks: particles are in cell order, so each private 4 MB L2
// then caches one spatial slab of positions instead of all of them.  Bijective for any grid size;
// placement is a speed matter only.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb) {
  const unsigned q = nb / 8u, r = nb % 8u, x = b % 8u, i = b / 8u;
  return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i;
}

#define CHX_FL_INVALID 0xFFFFFFFFu

// ---------------------------------------------------------------------------
// Two list entries per lane at a time, in packed fp32 (v_pk_add/mul/fma_f32: two IEEE
// operations per issue slot, each rounded exactly like its scalar twin).
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 f2_splat(float v) { return (f2){v, v}; }

// chiron/neighbors.py:79 for both halves: mod(d + L/2, L) - L/2, valid for |d| < L (the caller
// checks that and falls back to the general path otherwise).  With t = d + L/2 in (-L/2, 3L/2)
// jnp.mod picks, of {t - L, t, t + L}, the one in [0, L]: exactly the SMALLEST NON-NEGATIVE of
// the three, i.e. their minimum as unsigned bit patterns (negative floats have the top bit set).
// Same roundings as min_image(): t - L is exact (Sterbenz) and t + L rounds once.
__device__ __forceinline__ f2 min_image2_t(f2 t, float L, float hL) {  // t = d + L/2, already rounded
  const f2 a = t - L;
  const f2 c = t + L;
  f2 m;
  m.x = __uint_as_float(min(min(__float_as_uint(a.x), __float_as_uint(t.x)), __float_as_uint(c.x)));
  m.y = __uint_as_float(min(min(__float_as_uint(a.y), __float_as_uint(t.y)), __float_as_uint(c.y)));
  return m - hL;
}

// {a.x, b.x} and {a.y, b.y} in one instruction each (v_pk_mov_b32: D.lo = S0[op_sel[0]],
// D.hi = S1[op_sel[1]]; probed on the hardware, scripts/ub/pk_mov_probe.hip).  The gathers land
// as (x, y, z, w) per neighbor; the packed arithmetic wants (x_u, x_u+1): without this the
// compiler shuffles single registers, ~10 v_mov per pair of entries.
__device__ __forceinline__ f2 pk_xx(f2 a, f2 b) {
  f2 r;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_yy(f2 a, f2 b) {
  f2 r;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Minimum image of a wavefront's pairs (launch-uniform):
//   LJ_OPEN   non-periodic space, r = x_lo - x_hi;
//   LJ_SHIFT  periodic: min_image2_t (the three candidates), |d| < L checked per pair.
// (A third form for particles far from every box face -- t = d + L/2 lies in [0, L), so
// r = t - L/2 with a range check on t -- saves a quarter of the arithmetic and was measured to
// change nothing: at N = 262144 the kernel waits on the L1's gather rate)
#define LJ_OPEN 0
#define LJ_SHIFT 2

struct RowAcc {
  f2 ax, ay, az, ae;
  bool bad;
  bool keep0, keep1;  // PRUNE: entries u / u+1 of the last lj_pair2 call stay in the inner list
};

// what the pruning variant of the step kernel needs besides the list it walks (see k_lj_step)
struct PruneOut {
  uint32_t* fli;   // inner list, same geometry as the force list
  float T_inner;   // r2 <= T_inner: kept
};

// two list entries (u, u+1) of one lane; xj*: the gathered neighbor positions as (x,y) and (z,w)
// (ACT0: the first entry may be a padding slot as well -- the Monte Carlo kernel's per-lane inner list, chx_mc.hip)
template <int MODE, bool PRUNE, bool ACT0 = false>
__device__ __forceinline__ void lj_pair2(const float4& xi, f2 xy0, f2 zw0, f2 xy1, f2 zw1, uint32_t e0, uint32_t e1,
                                         bool act1, const Box3& box, const LJParams& lj, float T_inner, RowAcc& A,
                                         bool act0 = true) {
  // orientation: r = x_lo - x_hi by ORIGINAL id = +-(xi - xj); multiplying by +-1 is exact
  const f2 sgn = {(e0 & CHX_FL_HI) ? 1.0f : -1.0f, (e1 & CHX_FL_HI) ? 1.0f : -1.0f};
  const f2 dx = xi.x - pk_xx(xy0, xy1);
  const f2 dy = xi.y - pk_yy(xy0, xy1);
  const f2 dz = xi.z - pk_xx(zw0, zw1);
  f2 rx, ry, rz;
  if (MODE == LJ_OPEN) {
    rx = dx * sgn, ry = dy * sgn, rz = dz * sgn;
  } else {
    // t = (+-d) + L/2 with ONE rounding: +-d is exact, so the fma equals the reference's add
    const f2 tx = __builtin_elementwise_fma(dx, sgn, f2_splat(box.hx));
    const f2 ty = __builtin_elementwise_fma(dy, sgn, f2_splat(box.hy));
    const f2 tz = __builtin_elementwise_fma(dz, sgn, f2_splat(box.hz));
    const bool bad0 = !(fabsf(dx.x) < box.lx) | !(fabsf(dy.x) < box.ly) | !(fabsf(dz.x) < box.lz);
    const bool bad1 = !(fabsf(dx.y) < box.lx) | !(fabsf(dy.y) < box.ly) | !(fabsf(dz.y) < box.lz);
    A.bad |= bad0 | bad1;
    // unwrapped coordinates: the fast minimum image says nothing about such a pair, keep it listed
    if (PRUNE) A.keep0 = bad0, A.keep1 = bad1;
    rx = min_image2_t(tx, box.lx, box.hx);
    ry = min_image2_t(ty, box.ly, box.hy);
    rz = min_image2_t(tz, box.lz, box.hz);
  }
  const f2 r2 = (rx * rx + ry * ry) + rz * rz;  // norm2() in both halves
  if (PRUNE) {
    if (MODE == LJ_OPEN) A.keep0 = A.keep1 = false;
    A.keep0 |= r2.x <= T_inner;
    A.keep1 = act1 && (A.keep1 || r2.y <= T_inner);
  }
  const bool in0 = (ACT0 ? act0 : true) && r2.x <= lj.T;
  const bool in1 = act1 && r2.y <= lj.T;
  const f2 inv = {__builtin_amdgcn_rcpf(r2.x), __builtin_amdgcn_rcpf(r2.y)};
  const f2 s2 = inv * lj.sig2;
  const f2 s6 = s2 * s2 * s2;
  f2 e = (s6 * lj.eps4) * (s6 - 1.0f);
  f2 fs = (s6 * lj.eps24) * __builtin_elementwise_fma(f2_splat(2.0f), s6, f2_splat(-1.0f)) * inv;
  fs.x = in0 ? fs.x : 0.0f;
  fs.y = in1 ? fs.y : 0.0f;
  e.x = in0 ? e.x : 0.0f;
  e.y = in1 ? e.y : 0.0f;
  fs *= sgn;  // action on the lower id, reaction on the higher
  A.ax = __builtin_elementwise_fma(fs, rx, A.ax);
  A.ay = __builtin_elementwise_fma(fs, ry, A.ay);
  A.az = __builtin_elementwise_fma(fs, rz, A.az);
  A.ae += e;
}

// the chunk loop of one lane: chunk 0 (index quad e4, its four gathers) is already in flight.
// PRUNE: the entries with r2 <= T_inner are also copied, in order, into row p of the inner list
// (ordered compaction over the G lanes that share the row); `kept` returns the row's new length.
template <int G, int MODE, bool PRUNE>
__device__ __forceinline__ void lj_row_loop(const float4* __restrict__ x4, const uint4* __restrict__ col,
                                            size_t stride, int d, int nc, int q, const float4& xi, uint4 e4,
                                            float4 (&xj)[4], const Box3& box, const LJParams& lj, RowAcc& A,
                                            const PruneOut& po, int p, int n, int& kept) {
  constexpr int W = 4 * G;
  int base = 0;  // PRUNE: entries kept so far in this row (uniform over the G lanes)
  // At least one pass, also for an empty row (every `k < d` test then fails): with a zero-trip loop the
  // compiler sinks the first index quad and its four gathers below the wait for the row length `d`, which
  // puts THREE dependent memory round trips (d, indices, positions) in front of the first pair instead of two.
  // (One lane per particle -- G = 1, N >= 2^18 -- is throughput-bound and measured 6 % SLOWER that way: it keeps
  // the zero-trip form.)
  int c = 0;
  if (G > 1 || nc > 0) do {
    const int k0 = c * W + q;  // row index of this lane's word 0; word u is row entry k0 + u G (chx_fl_word)
    uint4 nx = make_uint4(0u, 0u, 0u, 0u);
    if (c + 1 < nc) nx = col[(size_t)(c + 1) * stride];
    bool kp[4] = {false, false, false, false};
    if (k0 < d) {
      lj_pair2<MODE, PRUNE>(xi, (f2){xj[0].x, xj[0].y}, (f2){xj[0].z, xj[0].w}, (f2){xj[1].x, xj[1].y},
                            (f2){xj[1].z, xj[1].w}, e4.x, e4.y, k0 + G < d, box, lj, po.T_inner, A);
      if (PRUNE) kp[0] = A.keep0, kp[1] = A.keep1;
    }
    if (k0 + 2 * G < d) {
      lj_pair2<MODE, PRUNE>(xi, (f2){xj[2].x, xj[2].y}, (f2){xj[2].z, xj[2].w}, (f2){xj[3].x, xj[3].y},
                            (f2){xj[3].z, xj[3].w}, e4.z, e4.w, k0 + 3 * G < d, box, lj, po.T_inner, A);
      if (PRUNE) kp[2] = A.keep0, kp[3] = A.keep1;
    }
    if (PRUNE) {
      // (all G lanes of a row run the same number of chunks, so the shuffles below are group-uniform)
      // kept entries go out in ROW order: word u of lanes 0..G-1, then word u + 1 (the chunk's interleaving)
      const uint32_t ent[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mine = (int)kp[u];
        int incl = mine;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
          const int v = __shfl_up(incl, o, G);
          if (q >= o) incl += v;
        }
        if (kp[u]) po.fli[chx_fl_addr(p, base + incl - mine, n, G)] = ent[u];
        base += __shfl(incl, G - 1, G);
      }
    }
    if (c + 1 < nc) {
      e4 = nx;
      xj[0] = x4[e4.x & ~CHX_FL_HI];
      xj[1] = x4[e4.y & ~CHX_FL_HI];
      xj[2] = x4[e4.z & ~CHX_FL_HI];
      xj[3] = x4[e4.w & ~CHX_FL_HI];
    }
  } while (++c < nc);
  kept = base;
}

// One particle's share of its force-list row.  A row is stored in chunks of W = 4 G entries
// (fl[chunk][particle][W]); lane q of the G cooperating lanes owns entries 4q..4q+3 of every
// chunk and fetches them with ONE 16-byte load, so a wavefront reads 1 KB of contiguous index
// data per instruction.  The first chunk is requested together with the particle's own
// position and degree (slots past a row's end hold a valid index -- the builders write the row's
// own index there, a zero-length pair that stays on the fast path -- so the speculative
// gathers are harmless); the index quad of chunk c+1 is requested before the arithmetic of
// chunk c, so only the position gathers sit on the dependent-latency chain.
//
// The kernel is VALU-bound at large N (a wave64 op takes 4 cycles on a SIMD16), so the pair
// arithmetic is branch-free and packed two entries wide; a lane whose fast form does not apply
// (unwrapped input coordinates, a list that does not match the positions) redoes its share
// through the general min_image().
// Loads the optimiser leaves where they are written: relaxed atomic loads at wavefront scope are ordinary
// global_load instructions (no cache-policy bits), but unlike plain loads they are not sunk below a branch
// whose other side does not use them.  k_lj_step issues its first row loads this way AHEAD of the halt /
// stale-list control words, so that the vector round trip overlaps the scalar ones instead of following them.
__device__ __forceinline__ uint4 pinned_load16(const void* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ int pinned_load4(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Sum over the G lanes that share a particle, result in every lane of the group.  Levels inside a row of 16
// lanes use DPP operand swizzles (quad permutes, half-row / row mirror: no LDS crossbar round trip as with
// ds_bpermute); wider groups finish with shuffles.  Fixed association, the same in every kernel.
__device__ __forceinline__ float dpp_f(float v, const int ctrl_sel) {
  const int i = __float_as_int(v);
  int r;
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]: lane ^ 1
    case 1: r = __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]: lane ^ 2
    case 2: r = __builtin_amdgcn_update_dpp(i, i, 0x141, 0xF, 0xF, true); break;  // row_half_mirror: 7 - lane in 8
    default: r = __builtin_amdgcn_update_dpp(i, i, 0x140, 0xF, 0xF, true); break; // row_mirror: 15 - lane in 16
  }
  return __int_as_float(r);
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if (G >= 2) v += dpp_f(v, 0);
  if (G >= 4) v += dpp_f(v, 1);   // every lane of a quad now holds the quad's sum
  if (G >= 8) v += dpp_f(v, 2);   // ... of a half row
  if (G >= 16) v += dpp_f(v, 3);  // ... of a row
#pragma unroll
  for (int o = 16; o < G; o <<= 1) v += __shfl_xor(v, o, CHX_WAVE);
  return v;
}

// write-through stores (see step_epilogue)
typedef float f3v __attribute__((ext_vector_type(3)));
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt4(float4* p, float4 v) {
  const f4v t = {v.x, v.y, v.z, v.w};
  // (s_nop: a VMEM store of more than 64 bits reads its data registers a cycle or two after it issues; the
  // compiler pads that hazard for its own stores but cannot see one inside an asm statement, and -- in an
  // experimental variant of the integrate tail -- was caught recycling the first data register for the next
  // value: velocities came out holding path lengths)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void store_wt3(float* p, float a, float b, float c) {
  const f3v t = {a, b, c};
  asm volatile("global_store_dwordx3 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}

// what row_begin has in flight for one lane
struct RowLoads {
  float4 xi;
  int d_raw;  // deg[p] as loaded (not yet masked by `valid`)
  uint4 e4;
  float4 xj[4];
  const uint4* col;
  int pc;
};

// Issue the loads of a lane's row: own position, row length, first index quad and its four gathers.
template <int G>
__device__ __forceinline__ RowLoads lj_row_begin(const float4* __restrict__ x4, const uint32_t* __restrict__ fl,
                                                 const int32_t* __restrict__ deg, int n, int p, int q, bool valid) {
  RowLoads L;
  L.pc = valid ? p : 0;
  L.col = reinterpret_cast<const uint4*>(fl) + (size_t)L.pc * G + q;
  L.xi = x4[L.pc];
  L.d_raw = deg[L.pc];
  L.e4 = L.col[0];
  L.xj[0] = x4[L.e4.x & ~CHX_FL_HI];
  L.xj[1] = x4[L.e4.y & ~CHX_FL_HI];
  L.xj[2] = x4[L.e4.z & ~CHX_FL_HI];
  L.xj[3] = x4[L.e4.w & ~CHX_FL_HI];
  return L;
}

// the same from values that are already on their way (k_lj_step's speculative loads): only the gathers are issued
template <int G>
__device__ __forceinline__ RowLoads lj_row_adopt(const float4* __restrict__ x4, const uint4* col, int pc, uint4 xi_bits,
                                                 int d_raw, uint4 e4) {
  RowLoads L;
  L.pc = pc;
  L.col = col;
  L.xi = make_float4(__uint_as_float(xi_bits.x), __uint_as_float(xi_bits.y), __uint_as_float(xi_bits.z),
                     __uint_as_float(xi_bits.w));
  L.d_raw = d_raw;
  L.e4 = e4;
  L.xj[0] = x4[e4.x & ~CHX_FL_HI];
  L.xj[1] = x4[e4.y & ~CHX_FL_HI];
  L.xj[2] = x4[e4.z & ~CHX_FL_HI];
  L.xj[3] = x4[e4.w & ~CHX_FL_HI];
  return L;
}

template <int G, bool PRUNE = false>
__device__ __forceinline__ void lj_row_finish(RowLoads& L, const float4* __restrict__ x4, const uint32_t* __restrict__ fl,
                                              int n, int q, bool valid, const Box3& box, bool periodic,
                                              const LJParams& lj, float4& xi_out, float& fx, float& fy, float& fz, float& en,
                                              const PruneOut& po = PruneOut{nullptr, 0.f}, int* kept_out = nullptr) {
  constexpr int W = 4 * G;
  const int pc = L.pc;
  const size_t stride = (size_t)n * G;  // uint4 units between consecutive chunks
  const float4 xi = L.xi;
  const int d = valid ? L.d_raw : 0;
  xi_out = xi;
  const int nc = (d + W - 1) / W;
  RowAcc A;
  A.ax = A.ay = A.az = A.ae = f2_splat(0.f);
  A.bad = false;
  A.keep0 = A.keep1 = false;
  int kept = 0;
  if (!periodic) {
    lj_row_loop<G, LJ_OPEN, PRUNE>(x4, L.col, stride, d, nc, q, xi, L.e4, L.xj, box, lj, A, po, pc, n, kept);
  } else {
    lj_row_loop<G, LJ_SHIFT, PRUNE>(x4, L.col, stride, d, nc, q, xi, L.e4, L.xj, box, lj, A, po, pc, n, kept);
  }
  if (PRUNE) *kept_out = kept;
  if (__builtin_expect(A.bad, 0)) {
    // general path, one entry at a time (bit-identical to the fast paths wherever those are valid)
    A.ax = A.ay = A.az = A.ae = f2_splat(0.f);
    const uint32_t* row = fl + ((size_t)pc * G + q) * 4;
    for (int c = 0; c < nc; ++c) {
      for (int u = 0; u < 4; ++u) {
        if (c * W + u * G + q >= d) break;
        const uint32_t en_t = row[(size_t)c * stride * 4 + u];
        const float4 xo = x4[en_t & ~CHX_FL_HI];
        const bool hi = (en_t & CHX_FL_HI) != 0u;
        float gx, gy, gz;
        displacement_oriented(box, hi, xi.x, xi.y, xi.z, xo.x, xo.y, xo.z, gx, gy, gz);
        const float g2 = norm2(gx, gy, gz);
        if (g2 <= lj.T) {
          float e, fs;
          lj_pair(g2, lj.sig2, lj.eps4, lj.eps24, e, fs);
          if (!hi) fs = -fs;
          const int h = u & 1;  // same accumulator parity as the fast path
          A.ax[h] = __builtin_fmaf(fs, gx, A.ax[h]);
          A.ay[h] = __builtin_fmaf(fs, gy, A.ay[h]);
          A.az[h] = __builtin_fmaf(fs, gz, A.az[h]);
          A.ae[h] += e;
        }
      }
    }
  }
  fx += A.ax.x + A.ax.y;
  fy += A.ay.x + A.ay.y;
  fz += A.az.x + A.az.y;
  en += A.ae.x + A.ae.y;
}

template <int G, bool PRUNE = false>
__device__ __forceinline__ void lj_row(const float4* __restrict__ x4, const uint32_t* __restrict__ fl,
                                       const int32_t* __restrict__ deg, int n, int chunks, int p, int q,
                                       bool valid, const Box3& box, const LJParams& lj, float4& xi_out,
                                       float& fx, float& fy, float& fz, float& en,
                                       const PruneOut& po = PruneOut{nullptr, 0.f}, int* kept_out = nullptr) {
  (void)chunks;
  RowLoads L = lj_row_begin<G>(x4, fl, deg, n, p, q, valid);
  lj_row_finish<G, PRUNE>(L, x4, fl, n, q, valid, box, box.periodic != 0, lj, xi_out, fx, fy, fz, en, po, kept_out);
}

