// libchx: one-pass cell-list build for the Langevin hot loop.
//
// chx_nl_build_cell_fl produces, from the same scan of the 27 surrounding cells,
//   (1) the reference's half neighbor list rows (ascending j > i with d < cutoff+skin,
//       chiron/neighbors.py:548-626) -- only the REAL entries of each row plus the row's
//       count and its first neighbor; padding and the (N,M) mask are materialised on demand
//       by chx_nl_finish_padding, because the hot loop never reads them, and
//   (2) the force list (full neighbor table in cell order, see include/chx.h) directly,
//       in scan order (nearly ascending neighbor index), with no atomics and no sort.
// Both come from the same fp32 predicate evaluated in the reference's orientation
// (x_lo - x_hi by original id), so the pair sets are identical by construction.
#include <cstdlib>
#include <type_traits>

#include "chx_common.h"
#include "chx_launch.h"
#include "chx_internal.h"
#include "chx_cell.h"

#define CHX_CELL_CAP 1024  // half-row candidates kept on chip per row


// One wavefront builds the rows of ONE particle (cell-order index w): scans the 27 surrounding cells, writes the
// force-list row in scan order (WRITE_FL) and the half row rank-sorted ascending, as jnp.where(mask, size=M) returns
// it.  cand = this wave's CHX_CELL_CAP words of LDS.
// the per-cell candidate lists k_cell_sort may have written (chx_cell.h); list == NULL: none
struct CandLists {
  const uint32_t* list;
  const int32_t* len;
  int cap;
};

// WRITE_HALF == false: the half row is only COUNTED (n_nbr: the capacity checks of the build need it), not stored or
// sorted -- the integrator's rebuilds, whose half rows nobody reads (chx_nl_rows_cell_fl: nl_rows_out == NULL); cand
// is then CHX_RANGE_WORDS words (the range table of a cell without a candidate list).
#define CHX_RANGE_WORDS 64
template <bool WRITE_FL, bool WRITE_HALF = true>
__device__ __forceinline__ void build_row_of_particle(
    int w, uint32_t* cand, const float4* __restrict__ xs4, int n, int m, const Box3& box, float T, const CellGrid& g,
    const int32_t* __restrict__ cell_start, const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ nl,
    uint32_t* __restrict__ first_out, int32_t* __restrict__ n_nbr, int fg, int fcap, uint32_t* __restrict__ fl,
    int32_t* __restrict__ deg, int32_t* __restrict__ status, const CandLists& cl) {
  const int lane = lane_id();
  const uint32_t i = sorted_ids[w];
  const float4 xw = xs4[w];  // positions in cell order: candidate loads below are coalesced
  const float xi = xw.x, yi = xw.y, zi = xw.z;
  const int c = __float_as_int(xw.w);  // packed by k_cell_sort: 10 bits per axis
  const int cx = c & 1023, cy = (c >> 10) & 1023, cz = c >> 20;
  int count_h = 0, count_f = 0;
  // one batch of 64 candidates: candidate index k (into the cell-ordered arrays), its original id j, its position
  auto test_batch = [&](int t, int total, int k, uint32_t j, const float4& xk) __attribute__((always_inline)) {
    bool ok = false, hi = false;
    if (t < total) {
      hi = j > i;
      float rx, ry, rz;  // x_i - x_j for row i of the half list, x_j - x_i for row j
      bool bad;
      displacement_oriented_fast(box, hi, xi, yi, zi, xk.x, xk.y, xk.z, rx, ry, rz, bad);
      if (__builtin_expect(bad, 0)) displacement_oriented(box, hi, xi, yi, zi, xk.x, xk.y, xk.z, rx, ry, rz);
      ok = (j != i) & (norm2(rx, ry, rz) <= T);
    }
    if (WRITE_FL) {
      const unsigned long long b = __ballot(ok);
      if (b) {
        const int slot = count_f + lanes_below(b);
        if (ok && slot < fcap) fl[chx_fl_addr(w, slot, n, fg)] = (uint32_t)k | (hi ? CHX_FL_HI : 0u);
        count_f += __popcll(b);
      }
    }
    const bool okh = ok && hi;
    const unsigned long long bh = __ballot(okh);
    if (!WRITE_HALF) {
      count_h += __popcll(bh);
    } else if (bh) {
      const int slot = count_h + lanes_below(bh);
      if (okh && slot < CHX_CELL_CAP - 40) cand[slot] = j;
      count_h += __popcll(bh);
    }
  };
  const int c_lin = (cz * g.ncy + cy) * g.ncx + cx;
  const int listed_total = cl.list != nullptr ? cl.len[c_lin] : 0x7FFFFFFF;
  if (listed_total <= cl.cap) {
    // The cell's neighbourhood is written out as a list of candidate indices (k_cell_sort, once per cell): read it with
    // coalesced loads.  Indices two batches ahead and the gathers of the next batch are requested before the current
    // batch is processed, so the loop's loads overlap instead of forming a chain of dependent round trips.
    const int total = listed_total;
    const uint32_t* lst = cl.list + (size_t)c_lin * cl.cap;
    uint32_t k_n = lane < total ? lst[lane] : 0u;
    uint32_t k_nn = lane + CHX_WAVE < total ? lst[lane + CHX_WAVE] : 0u;
    uint32_t j_n = sorted_ids[k_n];
    float4 x_n = xs4[k_n];
    for (int t0 = 0; t0 < total; t0 += CHX_WAVE) {
      const int t = t0 + lane;
      const uint32_t j = j_n;
      const int k = (int)k_n;
      const float4 xk = x_n;
      if (t0 + CHX_WAVE < total) {  // (wave-uniform)
        k_n = k_nn;
        j_n = sorted_ids[k_n];
        x_n = xs4[k_n];
        k_nn = t + 2 * CHX_WAVE < total ? lst[t + 2 * CHX_WAVE] : 0u;
      }
      test_batch(t, total, k, j, xk);
    }
  } else {
    // no list for this cell (none written, or the neighbourhood is longer than a list holds): work the neighbourhood's
    // index ranges out here and search them batch by batch
    int* rng_s = reinterpret_cast<int*>(cand) + (WRITE_HALF ? CHX_CELL_CAP : CHX_RANGE_WORDS) - 40;  // tail of the LDS slice
    int* rng_o = rng_s + 20;
    const int total = cell_ranges(g, cx, cy, cz, cell_start, lane, rng_s, rng_o);
    int r = 0;  // range holding this lane's candidate; candidates only move forward
    int k_n = 0;
    uint32_t j_n = 0;
    float4 x_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < total) {
      while (r < 17 && rng_o[r + 1] <= lane) ++r;
      k_n = rng_s[r] + (lane - rng_o[r]);
      j_n = sorted_ids[k_n];
      x_n = xs4[k_n];
    }
    for (int t0 = 0; t0 < total; t0 += CHX_WAVE) {
      const int t = t0 + lane;
      const uint32_t j = j_n;
      const int k = k_n;
      const float4 xk = x_n;
      if (t + CHX_WAVE < total) {
        const int tn = t + CHX_WAVE;
        while (r < 17 && rng_o[r + 1] <= tn) ++r;
        k_n = rng_s[r] + (tn - rng_o[r]);
        j_n = sorted_ids[k_n];
        x_n = xs4[k_n];
      }
      test_batch(t, total, k, j, xk);
    }
  }
  // force-list row: the force kernel gathers whole chunks (and chunk 0 speculatively), so the
  // unused slots of the chunks it can touch must hold a valid index; later chunks are never read
  if (WRITE_FL) {
    const int wd = CHX_FL_QUAD * fg;
    const int used = count_f < fcap ? count_f : fcap;
    int end = (used + wd - 1) / wd * wd;
    end = end < wd ? wd : end;
    end = end < fcap ? end : fcap;
    // (the row's own index: a zero-length pair keeps the force kernel on its fast path)
    for (int sl = used + lane; sl < end; sl += CHX_WAVE) fl[chx_fl_addr(w, sl, n, fg)] = (uint32_t)w;
  }
  if (!WRITE_HALF) {
    if (lane == 0) {
      n_nbr[i] = count_h;
      if (WRITE_FL) deg[w] = count_f < fcap ? count_f : fcap;
      if (WRITE_FL && count_f > fcap) {
        atomicAdd(&status[4], 1);
        atomicMax(&status[3], count_f);
      }
    }
    __threadfence_block();  // (the range table in cand is reused by the caller's next row)
    return;
  }
  const int kept = count_h < CHX_CELL_CAP - 40 ? count_h : CHX_CELL_CAP - 40;
  __threadfence_block();
  // half row: rank sort -> ascending ids, as jnp.where(mask, size=M) returns them
  uint32_t* nl_row = nl + (size_t)i * m;
  uint32_t mn = 0xFFFFFFFFu;
  if (kept <= CHX_WAVE) {
    // the common case: the row fits one value per lane -- ranks from lane broadcasts (v_readlane), no LDS traffic
    const uint32_t v = lane < kept ? cand[lane] : 0xFFFFFFFFu;
    mn = v;
    int rank = 0;
    for (int u = 0; u < kept; ++u) rank += (uint32_t)__builtin_amdgcn_readlane((int)v, u) < v ? 1 : 0;
    if (lane < kept && rank < m) nl_row[rank] = v;
  } else {
    for (int t = lane; t < kept; t += CHX_WAVE) {
      const uint32_t v = cand[t];
      mn = v < mn ? v : mn;
      int rank = 0;
      for (int u = 0; u < kept; ++u) rank += cand[u] < v ? 1 : 0;
      if (rank < m) nl_row[rank] = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = __shfl_xor(mn, o, CHX_WAVE);
    mn = other < mn ? other : mn;
  }
  if (lane == 0) {
    n_nbr[i] = count_h;
    first_out[i] = mn;
    if (WRITE_FL) deg[w] = count_f < fcap ? count_f : fcap;
    // only the rare failure cases touch the shared status words here (one hot address serialises atomics chip-wide:
    // spreading the maxima over 64 slots with a read-before-atomic filter still cost the build 60 us at N = 32768,
    // measured in round 4); the maxima the host needs are reduced from n_nbr / deg by k_build_epilogue
    if (count_h > CHX_CELL_CAP - 40) atomicOr(&status[0], 1);
    if (WRITE_FL && count_f > fcap) {
      atomicAdd(&status[4], 1);
      atomicMax(&status[3], count_f);
    }
  }
  __threadfence_block();  // (cand is reused by the caller's next row)
}

// one wavefront per PARTICLE (the general kernel: any row length up to the on-chip candidate buffer)
template <bool WRITE_HALF>
__global__ __launch_bounds__(256) void k_nl_build_cell_fl(
    const float4* __restrict__ xs4, int n, int m, Box3 box, float T, CellGrid g,
    const int32_t* __restrict__ cell_of, const int32_t* __restrict__ cell_start,
    const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ nl, uint32_t* __restrict__ first_out,
    int32_t* __restrict__ n_nbr, int fg, int fcap, uint32_t* __restrict__ fl, int32_t* __restrict__ deg,
    int32_t* __restrict__ status, CandLists cl) {
  __shared__ uint32_t cand_all[4][WRITE_HALF ? CHX_CELL_CAP : CHX_RANGE_WORDS];
  const int w0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w0 >= n) return;
  build_row_of_particle<true, WRITE_HALF>(w0, cand_all[threadIdx.x >> 6], xs4, n, m, box, T, g, cell_start, sorted_ids, nl,
                                          first_out, n_nbr, fg, fcap, fl, deg, status, cl);
}

// SIXTEEN LANES per particle, four particles per wavefront, candidates FILTERED before they are tested (round 5): the
// integrator's rebuilds (half rows counted only) wherever the binning has written a candidate list per cell -- at any size.
// The per-particle kernel above is bound by vector-instruction issue (PMC, C3 equilibrated: 461 instructions per row at
// 4 cycles each on 1024 SIMDs = its 25 us): ~70 per batch of 64 candidates -- the reference's predicate in the reference's
// orientation, two ballots, slot and address arithmetic, the loads of ids and positions -- although only one candidate
// in six of the 27-cell neighbourhood lies within the cutoff, and ~200 per row of prologue and epilogue.  Here
//   * a wavefront issues prologue and epilogue once per FOUR rows, and a batch of 16 wastes fewer lanes on a list's tail;
//   * phase 1 runs every candidate through a CHEAP distance test (nearest image by rounding, 17 instructions, no id) with a
//     threshold widened far beyond its rounding error, and appends the survivors -- their positions in the candidate
//     list, in candidate order -- to the row's slice of LDS; phase 2 puts only those (one in six) through the exact
//     predicate and writes the row as the per-particle kernel does: same candidates in the same order through the same
//     arithmetic, so the same bits;
//   * candidate indices and positions come through buffer descriptors (a 32-bit lane offset plus a scalar; a load past the
//     end of its array returns zero), so no load is predicated and no index clamped, and the candidate indices of the first
//     two iterations are requested together with the list's length: the positions of both are in flight in the third
//     round trip.  The loop is unrolled twice over two register sets: nothing is moved from "next" to "current".
// A row whose cell has no list (neighbourhood longer than the capacity) or with more survivors than its slice holds is
// redone by the general routine, the whole wavefront on it.
#define CHX_SUB 16
// CHX_SURV_CAP (template): survivors kept per row -- 192 where the force list's own capacity is at most 160 (the BASELINE
// geometry: 6 x 16), 640 (20 KB of LDS per workgroup) for denser fluids.
// The filter's constants (host: sub_filter_of): 1/L per axis and the widened threshold.  Its nearest image, d - L rint(d/L),
// differs per component from the reference's (chx_common.h: min_image) by a few ulp of the box length, the squared
// distance by less than 6 sqrt(T) L 2^-20; the threshold is T widened by ten times that plus a thousandth of T, and a
// pair more than a box length and a half apart on some axis (unwrapped coordinates: the rounding error grows with the
// distance) passes untested.
struct SubFilter {
  float ilx, ily, ilz, t_wide;
};
static SubFilter sub_filter_of(const chx_box_t& box, float T) {
  const float lmax = box.lx > box.ly ? (box.lx > box.lz ? box.lx : box.lz) : (box.ly > box.lz ? box.ly : box.lz);
  return SubFilter{1.0f / box.lx, 1.0f / box.ly, 1.0f / box.lz, T * 1.001f + 6e-5f * lmax * sqrtf(T)};
}
template <int CHX_SURV_CAP>
__global__ __launch_bounds__(256) void k_nl_build_sub_fl(
    const float4* __restrict__ xs4, int n, int m, Box3 box, float T, CellGrid g, const int32_t* __restrict__ cell_start,
    const uint32_t* __restrict__ sorted_ids, int32_t* __restrict__ n_nbr, int fg, int fcap, uint32_t* __restrict__ fl,
    int32_t* __restrict__ deg, int32_t* __restrict__ status, CandLists cl, SubFilter filt) {
  __shared__ uint32_t cand_all[4][CHX_RANGE_WORDS];
  __shared__ uint16_t surv_all[4][CHX_WAVE / CHX_SUB][CHX_SURV_CAP];  // (positions in a candidate list: below its capacity, <= 8192)
  constexpr int U = 2, STEP = CHX_SUB * U;  // candidates per lane and half iteration of phase 1
  const int lane = lane_id();
  const int sub = lane & (CHX_SUB - 1), shift = lane & ~(CHX_SUB - 1);
  const int w_raw = (int)((blockIdx.x * blockDim.x + threadIdx.x) / CHX_SUB);
  if (__builtin_amdgcn_readfirstlane(w_raw) >= n) return;  // (wave-uniform: the wavefront's first row)
  const bool row_live = w_raw < n;
  const int w = row_live ? w_raw : n - 1;
  const uint32_t i = sorted_ids[w];
  const float4 xw = xs4[w];
  const float xi = xw.x, yi = xw.y, zi = xw.z;
  const int c = __float_as_int(xw.w);
  const int c_lin = ((c >> 20) * g.ncy + ((c >> 10) & 1023)) * g.ncx + (c & 1023);
  uint16_t* surv = surv_all[threadIdx.x >> 6][lane / CHX_SUB];
  const int ncell = g.ncx * g.ncy * g.ncz;
  const __amdgpu_buffer_rsrc_t r_lst = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(cl.list), 0, ncell * cl.cap * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(xs4), 0, n * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_id = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(sorted_ids), 0, n * 4, 0x00020000);
  const int row0 = c_lin * cl.cap * 4;  // byte offset of the row's candidate list
  const int lo = row0 + sub * 4;        // ... of this lane's first candidate index
  auto load_xyz = [&](uint32_t k, float* x) __attribute__((always_inline)) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(r_xs, (int)(k << 4), 0, 0);
    x[0] = __uint_as_float(v[0]);
    x[1] = __uint_as_float(v[1]);
    x[2] = __uint_as_float(v[2]);
  };
  // second round trip: the list's length and, beside it, the candidate indices of the first two half iterations (an index
  // read past the end of a row's list -- the next cell's, or stale words: a list is allocated to its capacity -- fetches
  // some position or zero, and the comparison with the list's length drops it)
  const int listed = cl.len[c_lin];
  uint32_t k0[U], kx[U];  // kx: the indices whose positions the next half iteration requests
  float xa[U][3], xb[U][3];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    k0[u] = __builtin_amdgcn_raw_buffer_load_b32(r_lst, lo + 4 * CHX_SUB * u, 0, 0);
    kx[u] = __builtin_amdgcn_raw_buffer_load_b32(r_lst, lo + 4 * CHX_SUB * u, 4 * STEP, 0);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) load_xyz(k0[u], xa[u]);
  bool redo = row_live && listed > cl.cap;
  const int total = (row_live && !redo) ? listed : 0;
  const uint32_t below = (1u << sub) - 1u;
  int ns = 0;  // survivors of this row so far
  // one half iteration: the candidates [t0, t0 + STEP) of every row, whose positions are in xc; the positions of the next
  // STEP (indices in k: loaded a half iteration ago) are requested into xn, then k is reloaded with the indices after those
  auto half = [&](int t0, float (*xc)[3], float (*xn)[3], uint32_t* k) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) load_xyz(k[u], xn[u]);
    const int ahead = __builtin_amdgcn_readfirstlane(4 * (t0 + 2 * STEP));  // (uniform: said so, or the load is issued lane by lane)
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = __builtin_amdgcn_raw_buffer_load_b32(r_lst, lo + 4 * CHX_SUB * u, ahead, 0);
    const int left = total - t0;  // candidates of this row not yet looked at
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float dx = xi - xc[u][0], dy = yi - xc[u][1], dz = zi - xc[u][2];
      const float qx = rintf(dx * filt.ilx), qy = rintf(dy * filt.ily), qz = rintf(dz * filt.ilz);
      const float rx = fmaf(-qx, box.lx, dx), ry = fmaf(-qy, box.ly, dy), rz = fmaf(-qz, box.lz, dz);
      const float d2 = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
      const float far = fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));
      const bool pass = (sub + CHX_SUB * u < left) & ((d2 <= filt.t_wide) | (far > 1.5f));
      const uint32_t b = (uint32_t)(__ballot(pass) >> shift) & 0xFFFFu;
      const int slot = ns + __popc(b & below);
      if (pass && slot < CHX_SURV_CAP) surv[slot] = (uint16_t)(t0 + sub + CHX_SUB * u);
      ns += __popc(b);
    }
  };
  for (int t0 = 0; __ballot(t0 < total) != 0ull; t0 += 2 * STEP) {
    half(t0, xa, xb, kx);
    if (__ballot(t0 + STEP < total) == 0ull) break;
    half(t0 + STEP, xb, xa, kx);
  }
  if (ns > CHX_SURV_CAP) {
    redo = true;
    ns = 0;
  }
  __threadfence_block();  // (the survivors are read back by other lanes of this wavefront)
  // phase 2: the survivors through the reference's predicate, 16 per row at a time (list position -> index -> id and
  // position: the position of the next batch and the index of the one after it are in flight while a batch is worked)
  const int sh = 31 - __clz(CHX_FL_QUAD * fg), lg = 31 - __clz(fg);
  uint32_t* fl_row = fl + ((size_t)w << sh);
  const uint32_t chunk_stride = (uint32_t)n << sh;  // (the launcher checks: the table has fewer than 2^31 words, this fits 24 bits)
  int count_f = 0, count_h = 0;
  auto index_of = [&](int sx) __attribute__((always_inline)) {  // (past the row's survivors: its first candidate, never tested)
    const uint32_t t = sx < ns ? surv[sx] : 0u;
    return __builtin_amdgcn_raw_buffer_load_b32(r_lst, row0 + (int)(t << 2), 0, 0);
  };
  uint32_t k_c = index_of(sub), k_n = index_of(sub + CHX_SUB);
  uint32_t j_c = __builtin_amdgcn_raw_buffer_load_b32(r_id, (int)(k_c << 2), 0, 0);
  float x_c[3];
  load_xyz(k_c, x_c);
  for (int s0 = 0; __ballot(s0 < ns) != 0ull; s0 += CHX_SUB) {
    const int sx = s0 + sub;
    const uint32_t k = k_c, j = j_c;
    const float xk = x_c[0], yk = x_c[1], zk = x_c[2];
    k_c = k_n;
    j_c = __builtin_amdgcn_raw_buffer_load_b32(r_id, (int)(k_c << 2), 0, 0);
    load_xyz(k_c, x_c);
    k_n = index_of(sx + 2 * CHX_SUB);
    const bool hi = j > i;
    float rx, ry, rz;
    bool bad;
    displacement_oriented_fast(box, hi, xi, yi, zi, xk, yk, zk, rx, ry, rz, bad);
    if (__builtin_expect(bad, 0)) displacement_oriented(box, hi, xi, yi, zi, xk, yk, zk, rx, ry, rz);
    const bool ok = (sx < ns) & (j != i) & (norm2(rx, ry, rz) <= T);
    const uint32_t b = (uint32_t)(__ballot(ok) >> shift) & 0xFFFFu;
    const int slot = count_f + __popc(b & below);
    if (ok && slot < fcap) {
      const uint32_t e = (uint32_t)slot & ((1u << sh) - 1u);
      fl_row[__umul24((uint32_t)slot >> sh, chunk_stride) + (CHX_FL_QUAD * (e & (uint32_t)(fg - 1)) + (e >> lg))] = k | (hi ? CHX_FL_HI : 0u);
    }
    count_f += __popc(b);
    count_h += (ok && hi) ? 1 : 0;  // (per lane: summed over the row's lanes once, below)
  }
#pragma unroll
  for (int o = CHX_SUB / 2; o > 0; o >>= 1) count_h += __shfl_xor(count_h, o, CHX_WAVE);
  if (row_live && !redo) {
    // unused slots of the chunks the force kernel can touch: the row's own index (build_row_of_particle)
    const int wd = CHX_FL_QUAD * fg;
    const int used = count_f < fcap ? count_f : fcap;
    int end = (used + wd - 1) / wd * wd;
    end = end < wd ? wd : end;
    end = end < fcap ? end : fcap;
    for (int sl = used + sub; sl < end; sl += CHX_SUB) {
      const uint32_t e = (uint32_t)sl & ((1u << sh) - 1u);
      fl_row[__umul24((uint32_t)sl >> sh, chunk_stride) + (CHX_FL_QUAD * (e & (uint32_t)(fg - 1)) + (e >> lg))] = (uint32_t)w;
    }
    if (sub == 0) {
      n_nbr[i] = count_h;
      deg[w] = used;
      if (count_f > fcap) {
        atomicAdd(&status[4], 1);
        atomicMax(&status[3], count_f);
      }
    }
  }
  unsigned long long todo = __ballot(redo && sub == 0);
  while (todo) {  // (wave-uniform)
    const int l = __builtin_ctzll(todo);
    todo &= todo - 1;
    build_row_of_particle<true, false>(__builtin_amdgcn_readlane(w, l), cand_all[threadIdx.x >> 6], xs4, n, m, box, T, g, cell_start,
                                       sorted_ids, nullptr, nullptr, n_nbr, fg, fcap, fl, deg, status, cl);
  }
}

// One wavefront per CELL: the (about ten at the BASELINE density) particles of a cell share their 27-cell
// neighbourhood, so the candidates are fetched once per cell -- one per lane, 64 at a time -- and tested against
// every particle of the cell in turn (its position broadcast from a lane: v_readlane), instead of once per particle.
// The per-particle kernel above is bound by VALU issue (PMC: ~1000 vector instructions per row, of which the range
// search, the candidate loads and the prologue are more than half); here those are shared by the cell's rows.
// Rows come out as the per-particle kernel writes them: force-list rows in the same scan order, half rows
// rank-sorted ascending.  Half-row hits wait for the sort in this wave's LDS slice (one 64-word row per particle, up
// to 15 particles per pass; C4: 309 -> 299 us per build call against the scheme below); a pass with more particles
// writes them to their row of `nl` unsorted first and sorts in place at the end; a row too long for either (> m, > 64
// resp. CHX_ROW_SORT_CAP entries) is redone by build_row_of_particle.
#define CHX_ROW_SORT_CAP 256
#define CHX_LDS_ROW_WORDS 64  // on-chip half rows: words per row ...
#define CHX_LDS_ROWS 15       // ... and rows per pass (15 x 64 words + the 40 words of the range table <= CHX_CELL_CAP)
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <bool WRITE_HALF>
__global__ __launch_bounds__(256) void k_nl_build_cellwise_fl(
    const float4* __restrict__ xs4, int n, int m, Box3 box, float T, CellGrid g, int ncell, int split,
    const int32_t* __restrict__ cell_start, const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ nl,
    uint32_t* __restrict__ first_out, int32_t* __restrict__ n_nbr, int fg, int fcap, uint32_t* __restrict__ fl,
    int32_t* __restrict__ deg, int32_t* __restrict__ status) {
  // sort staging (first CHX_ROW_SORT_CAP words) + ranges (tail)
  constexpr int SLICE = WRITE_HALF ? CHX_CELL_CAP : CHX_RANGE_WORDS;
  __shared__ uint32_t cand_all[4][SLICE];
  const int lane = lane_id();
  // `split` wavefronts share a cell, each taking a contiguous slice of its particles (few cells = few wavefronts:
  // the launch then lives on latency, and more, shorter waves beat fewer candidate fetches)
  const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int c = wid / split;
  if (c >= ncell) return;
  int h0 = cell_start[c], h1 = cell_start[c + 1];
  {
    const int per = (h1 - h0 + split - 1) / split;
    h0 += (wid % split) * per;
    h1 = h0 + per < h1 ? h0 + per : h1;
  }
  if (h1 <= h0) return;
  uint32_t* cand = cand_all[threadIdx.x >> 6];
  int* rng_s = reinterpret_cast<int*>(cand) + SLICE - 40;
  int* rng_o = rng_s + 20;
  const int cx = c % g.ncx, cy = (c / g.ncx) % g.ncy, cz = c / (g.ncx * g.ncy);
  const int total = cell_ranges(g, cx, cy, cz, cell_start, lane, rng_s, rng_o);
  for (int hb = h0; hb < h1; hb += CHX_WAVE) {  // the cell's particles, 64 at a time: lane l holds particle hb + l
    const int nh = h1 - hb < CHX_WAVE ? h1 - hb : CHX_WAVE;
    float4 xh = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t ih = 0;
    if (lane < nh) {
      xh = xs4[hb + lane];
      ih = sorted_ids[hb + lane];
    }
    int cf = 0, ch = 0;  // entries written so far to the force-list row / half row of this lane's particle
    // Half-row hits wait for their rank sort on chip when the pass has few enough particles (the BASELINE density:
    // ~10 per cell): row ii owns CHX_LDS_ROW_WORDS words of this wave's LDS slice, nothing is written to `nl`
    // unsorted and read back.  Denser cells keep the older scheme (unsorted to the row of `nl`, sorted in place).
    const bool lds_rows = WRITE_HALF && nh <= CHX_LDS_ROWS;
    int r = 0, k_n = 0;
    uint32_t j_n = 0;
    float4 x_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < total) {
      while (r < 17 && rng_o[r + 1] <= lane) ++r;
      k_n = rng_s[r] + (lane - rng_o[r]);
      j_n = sorted_ids[k_n];
      x_n = xs4[k_n];
    }
    auto scan = [&](auto on_chip) __attribute__((always_inline)) {
      constexpr bool ON_CHIP = decltype(on_chip)::value;
      for (int t0 = 0; t0 < total; t0 += CHX_WAVE) {
        const int t = t0 + lane;
        const uint32_t j = j_n;
        const int k = k_n;
        const float4 xk = x_n;
        if (t + CHX_WAVE < total) {  // next 64 candidates: in flight while this batch meets the cell's particles
          const int tn = t + CHX_WAVE;
          while (r < 17 && rng_o[r + 1] <= tn) ++r;
          k_n = rng_s[r] + (tn - rng_o[r]);
          j_n = sorted_ids[k_n];
          x_n = xs4[k_n];
        }
        const bool live = t < total;
        for (int ii = 0; ii < nh; ++ii) {
          const float xi = rl_f(xh.x, ii), yi = rl_f(xh.y, ii), zi = rl_f(xh.z, ii);
          const uint32_t i = (uint32_t)rl_i((int)ih, ii);
          bool ok = false, hi = false;
          if (live) {
            hi = j > i;
            float rx, ry, rz;  // x_i - x_j for row i of the half list, x_j - x_i for row j
            bool bad;
            displacement_oriented_fast(box, hi, xi, yi, zi, xk.x, xk.y, xk.z, rx, ry, rz, bad);
            if (__builtin_expect(bad, 0)) displacement_oriented(box, hi, xi, yi, zi, xk.x, xk.y, xk.z, rx, ry, rz);
            ok = (j != i) & (norm2(rx, ry, rz) <= T);
          }
          // (no branch on "any hit in this batch": two thirds of the (particle, batch) pairs have one, and a
          // straight-line body lets the compiler overlap consecutive particles)
          const unsigned long long b = __ballot(ok);
          {
            const int base = rl_i(cf, ii);
            const int slot = base + lanes_below(b);
            if (ok && slot < fcap) fl[chx_fl_addr(hb + ii, slot, n, fg)] = (uint32_t)k | (hi ? CHX_FL_HI : 0u);
            cf = lane == ii ? base + __popcll(b) : cf;
          }
          const bool okh = ok && hi;
          const unsigned long long bh = __ballot(okh);
          if (!WRITE_HALF) {
            ch = lane == ii ? ch + __popcll(bh) : ch;  // counted only
          } else {
            const int base = rl_i(ch, ii);
            const int slot = base + lanes_below(bh);
            if (ON_CHIP) {
              if (okh && slot < CHX_LDS_ROW_WORDS) cand[ii * CHX_LDS_ROW_WORDS + slot] = j;
            } else {
              if (okh && slot < m) nl[(size_t)i * m + slot] = j;  // unsorted; sorted in place below
            }
            ch = lane == ii ? base + __popcll(bh) : ch;
          }
        }
      }
    };
    if (lds_rows) scan(std::true_type{});
    else scan(std::false_type{});
    // the half-row entries written above are read back below by other lanes of this wavefront: same CU, same
    // (write-through) L1, so a workgroup-scope fence is all it takes.  (A device-scope fence here writes back the
    // XCD's whole L2 once per cell: measured 3x slower than the per-particle kernel.)
    __threadfence_block();
    unsigned long long redo = 0;  // rows the general routine has to rebuild (it uses this wave's whole LDS slice)
    if (!WRITE_HALF && lane < nh) n_nbr[ih] = ch;
    for (int ii = 0; ii < nh; ++ii) {
      const int count_f = rl_i(cf, ii), count_h = rl_i(ch, ii);
      const int w = hb + ii;
      const uint32_t i = (uint32_t)rl_i((int)ih, ii);
      {
        // force-list row: the force kernel gathers whole chunks (and chunk 0 speculatively), so the unused slots of
        // the chunks it can touch must hold a valid index (the row's own: a zero-length pair stays on the fast path)
        const int wd = CHX_FL_QUAD * fg;
        const int used = count_f < fcap ? count_f : fcap;
        int end = (used + wd - 1) / wd * wd;
        end = end < wd ? wd : end;
        end = end < fcap ? end : fcap;
        for (int sl = used + lane; sl < end; sl += CHX_WAVE) fl[chx_fl_addr(w, sl, n, fg)] = (uint32_t)w;
      }
      if (lane == 0) {
        deg[w] = count_f < fcap ? count_f : fcap;
        if (count_f > fcap) {
          atomicAdd(&status[4], 1);
          atomicMax(&status[3], count_f);
        }
      }
      if (!WRITE_HALF) {
        // (nothing stored, nothing to sort)
      } else if (count_h > m || count_h > (lds_rows ? CHX_LDS_ROW_WORDS : CHX_ROW_SORT_CAP)) {
        redo |= 1ull << ii;  // too long to sort this way (and not stored completely)
      } else if (lds_rows) {
        uint32_t* nl_row = nl + (size_t)i * m;
        const uint32_t* row = cand + ii * CHX_LDS_ROW_WORDS;
        const uint32_t v = lane < count_h ? row[lane] : 0xFFFFFFFFu;
        int rank = 0;
        for (int u = 0; u < count_h; ++u) rank += row[u] < v ? 1 : 0;
        if (lane < count_h) nl_row[rank] = v;
        uint32_t mn = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t other = __shfl_xor(mn, o, CHX_WAVE);
          mn = other < mn ? other : mn;
        }
        if (lane == 0) {
          n_nbr[i] = count_h;
          first_out[i] = mn;
        }
      } else {
        uint32_t* nl_row = nl + (size_t)i * m;
        uint32_t v[CHX_ROW_SORT_CAP / CHX_WAVE];
        uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < CHX_ROW_SORT_CAP / CHX_WAVE; ++q) {
          const int t = q * CHX_WAVE + lane;
          v[q] = 0xFFFFFFFFu;
          if (q * CHX_WAVE < count_h && t < count_h) {
            v[q] = nl_row[t];
            cand[t] = v[q];
            mn = v[q] < mn ? v[q] : mn;
          }
        }
        __threadfence_block();
#pragma unroll
        for (int q = 0; q < CHX_ROW_SORT_CAP / CHX_WAVE; ++q) {
          if (q * CHX_WAVE < count_h) {  // wave-uniform
            int rank = 0;
            for (int u = 0; u < count_h; ++u) rank += cand[u] < v[q] ? 1 : 0;
            if (q * CHX_WAVE + lane < count_h) nl_row[rank] = v[q];
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t other = __shfl_xor(mn, o, CHX_WAVE);
          mn = other < mn ? other : mn;
        }
        if (lane == 0) {
          n_nbr[i] = count_h;
          first_out[i] = mn;
        }
        __threadfence_block();  // (cand is reused for the next row)
      }
    }
    while (redo) {
      // the general path rebuilds these half rows from scratch (it writes n_nbr, first_out and the status flags itself;
      // cell_ranges inside rewrites the same values into the tail of cand)
      const int ii = __builtin_ctzll(redo);
      redo &= redo - 1;
      build_row_of_particle<false, true>(hb + ii, cand, xs4, n, m, box, T, g, cell_start, sorted_ids, nl, first_out, n_nbr, fg, fcap,
                                   fl, deg, status, CandLists{nullptr, nullptr, 0});
    }
  }
}

// The chain's last launch: status[1] = max n_nbr, status[2] = #rows with n_nbr == m, status[3] = max full degree,
// reduced from the arrays the build has written (a few dozen workgroups, one atomic each), and -- by whichever workgroup
// finishes last (ticket in status[7]) -- status[0..7] copied into host-mapped memory with system scope, so the host can
// read the block after waiting for the stream, or once it knows a later launch has begun: no publish launch.
__global__ __launch_bounds__(256) void k_build_epilogue(const int32_t* __restrict__ n_nbr, const int32_t* __restrict__ deg,
                                                        int n, int m, int32_t* __restrict__ status,
                                                        int32_t* __restrict__ host_dst8) {
  __shared__ int32_t red[3][4];
  __shared__ int last_block;
  int mx = 0, eq = 0, md = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int c = n_nbr[i];
    mx = c > mx ? c : mx;
    eq += c == m ? 1 : 0;
    const int d = deg[i];
    md = d > md ? d : md;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(mx, o, CHX_WAVE), b = __shfl_xor(md, o, CHX_WAVE);
    mx = a > mx ? a : mx;
    md = b > md ? b : md;
    eq += __shfl_xor(eq, o, CHX_WAVE);
  }
  const int wv = threadIdx.x >> 6;
  if (lane_id() == 0) {
    red[0][wv] = mx;
    red[1][wv] = eq;
    red[2][wv] = md;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      mx = red[0][k] > mx ? red[0][k] : mx;
      eq += red[1][k];
      md = red[2][k] > md ? red[2][k] : md;
    }
    atomicMax(&status[1], mx);
    if (eq) atomicAdd(&status[2], eq);
    atomicMax(&status[3], md);
    __threadfence();
    last_block = atomicAdd(&status[7], 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (last_block && host_dst8 != nullptr && threadIdx.x < 8) {
    const int v = threadIdx.x == 7 ? 0 : __hip_atomic_load(&status[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&host_dst8[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static int riders_to_bin(const chx_build_riders_t* riders, BinRiders* rd) {
  *rd = BinRiders{};
  if (riders == nullptr) return CHX_OK;
  rd->xyz_copy = riders->xyz_copy_out;
  if (riders->x4_out != nullptr) {
    CHX_REQUIRE(riders->v3 && riders->mass && riders->v_out && riders->mass_out, "pack rider: null pointer");
    rd->v3 = riders->v3;
    rd->mass = riders->mass;
    rd->ref3 = riders->ref3;
    rd->x4 = reinterpret_cast<float4*>(riders->x4_out);
    rd->v = riders->v_out;
    rd->mass_out = riders->mass_out;
    rd->ref4 = reinterpret_cast<float4*>(riders->ref4_out);
  }
  rd->ctrl = riders->ctrl_reset;
  rd->ctrl_keep = riders->ctrl_keep;
  CHX_REQUIRE(riders->cell_slot_cap >= 0 && (riders->cell_slots != nullptr || riders->cell_slot_cap == 0), "cell slots: bad capacity");
  rd->cell_slots = riders->cell_slots;
  rd->cell_slot_cap = riders->cell_slot_cap;
  return CHX_OK;
}

// first half of chx_nl_build_cell_fl: the binning launches (and their riders)
extern "C" int chx_cell_bin_fl(const float* xyz, int32_t n, chx_box_t box, float cutoff_plus_skin, const int32_t* nc_h,
                               int32_t* cell_of, int32_t* cell_count, int32_t* cell_start, uint32_t* sorted_ids,
                               int32_t* status8, float* xs4_out, const chx_build_riders_t* riders, chx_stream_t stream) {
  CHX_REQUIRE(n >= 1, "need n >= 1");
  CHX_REQUIRE(xyz && nc_h && cell_of && cell_count && cell_start && sorted_ids && status8 && xs4_out, "null pointer");
  BinRiders rd;
  int rc = riders_to_bin(riders, &rd);
  if (rc != CHX_OK) return rc;
  CellGrid grid;
  rc = chx_cell_prepare(xyz, n, box, cutoff_plus_skin, nc_h, cell_of, cell_count, cell_start, sorted_ids, status8, 8, &grid,
                        chx_stream_q(stream), xs4_out, &rd,  // (the cell sort also emits xs4 and the candidate lists)
                        riders ? riders->cand_list : nullptr, riders ? riders->cand_len : nullptr,
                        riders ? riders->cand_cap : 0);
  if (rc != CHX_OK) return rc;
  return chx_check_launch("chx_cell_bin_fl");
}

// second half: the rows (half list + force list) from the binned particles, the summary and its publish
extern "C" int chx_nl_rows_cell_fl(int32_t n, int32_t m, chx_box_t box, float cutoff_plus_skin, const int32_t* nc_h,
                                   const int32_t* cell_of, const int32_t* cell_start, const uint32_t* sorted_ids,
                                   int32_t* status8, const float* xs4, uint32_t* nl_rows_out, uint32_t* first_out,
                                   int32_t* n_nbr_out, int32_t g, int32_t chunks, uint32_t* fl_out, int32_t* deg_out,
                                   int32_t* publish_host_dst8, const uint32_t* cand_list, const int32_t* cand_len,
                                   int32_t cand_cap, chx_stream_t stream) {
  CHX_REQUIRE(n >= 1 && m >= 1, "need n >= 1, m >= 1");
  CHX_REQUIRE((cand_list == nullptr) == (cand_len == nullptr) && cand_cap >= 0, "candidate lists need both arrays");
  CHX_REQUIRE(nc_h && cell_of && cell_start && sorted_ids && status8 && xs4 && n_nbr_out && fl_out && deg_out, "null pointer");
  CHX_REQUIRE((nl_rows_out == nullptr) == (first_out == nullptr), "half rows: nl_rows_out and first_out go together");
  const bool half = nl_rows_out != nullptr;  // else: the half rows are counted (n_nbr_out), not written
  CHX_REQUIRE(g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64,
              "g must be a power of two <= 64");
  CHX_REQUIRE(chunks >= 1 && (int64_t)chunks * CHX_FL_QUAD * g < (1 << 20), "bad force-list capacity");
  CHX_REQUIRE(box.periodic && box.lx > 0 && box.ly > 0 && box.lz > 0, "periodic box required");
  CHX_REQUIRE(nc_h[0] >= 3 && nc_h[1] >= 3 && nc_h[2] >= 3 && nc_h[0] <= 1024 && nc_h[1] <= 1024 && nc_h[2] <= 1024,
              "need 3 .. 1024 cells per axis");
  hipStream_t st = chx_stream_q(stream);  // (CHX_LAUNCH below: rides on the step queue while a run() is in progress)
  CellGrid grid;
  grid.ncx = nc_h[0]; grid.ncy = nc_h[1]; grid.ncz = nc_h[2];
  grid.sx = (float)nc_h[0] / box.lx; grid.sy = (float)nc_h[1] / box.ly; grid.sz = (float)nc_h[2] / box.lz;
  const float T = chx_cutoff_threshold_h(cutoff_plus_skin);
  const unsigned wb = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);
  const float4* xs = reinterpret_cast<const float4*>(xs4);
  // One wavefront per particle, or -- when there are enough cells to fill the chip with one wavefront each -- one
  // per cell, the cell's particles sharing their candidate scan (measured: N = 262144 / 27000 cells 355 -> 263 us;
  // N = 32768 / 3375 cells 51 -> 83 us, or 51 -> 51 with four wavefronts per cell: too few, too long wavefronts
  // live on latency).  Both write the same bits; CHX_BUILD_KERNEL=particle|cell forces one (tests, A/B).
  static const int forced = [] { const char* e = getenv("CHX_BUILD_KERNEL"); return !e ? 0 : (e[0] == 'p' ? 1 : (e[0] == 'c' ? 2 : 0)); }();
  const CandLists cl = {cand_cap > 0 ? cand_list : nullptr, cand_len, cand_cap};
  const int ncell = grid.ncx * grid.ncy * grid.ncz;
  const int fcap = chunks * CHX_FL_QUAD * g;
  // The integrator's rebuilds (no half rows) with candidate lists, at any size: sixteen lanes per row and a filter in front
  // of the predicate (k_nl_build_sub_fl; its 32-bit offsets and 24-bit chunk stride bound the size; C4: 229 -> 161 us per
  // build against the cell-wise kernel, C3: 47 -> 39 against the per-particle one).  A row with more survivors than the
  // kernel keeps on chip is done twice: only where the force list's capacity -- sized from the longest row of earlier
  // builds -- says that rows stay below it.  CHX_BUILD_SUB=0: the other kernels (tests, A/B).
  static const bool sub_rows = [] { const char* e = getenv("CHX_BUILD_SUB"); return !e || e[0] != '0'; }();
  static const int sub_max = [] { const char* e = getenv("CHX_BUILD_SUB_MAX"); return e ? atoi(e) : 576; }();
  const bool sub = !half && cl.list != nullptr && sub_rows && forced != 2 && fcap <= sub_max && (int64_t)fcap * n < (int64_t)1 << 31 &&
                   (int64_t)CHX_FL_QUAD * g * n < 1 << 24 && (int64_t)ncell * cand_cap * 4 < (int64_t)1 << 31 && n < 1 << 27 &&
                   cand_cap <= 65536;  // (a survivor is kept as a 16-bit position in its candidate list)
  const bool per_particle = forced == 1 || (forced == 0 && ncell < 8192);
  if (sub) {
    const unsigned qb = (unsigned)(((int64_t)n * CHX_SUB + 255) / 256);
    CHX_LAUNCH(fcap <= 160 ? k_nl_build_sub_fl<192> : k_nl_build_sub_fl<640>, dim3(qb), dim3(256), st, xs, n, m, make_box3(box), T, grid,
               cell_start, sorted_ids, n_nbr_out, g, fcap, fl_out, deg_out, status8, cl, sub_filter_of(box, T));
  } else if (per_particle) {
    CHX_LAUNCH(half ? k_nl_build_cell_fl<true> : k_nl_build_cell_fl<false>, dim3(wb), dim3(256), st, xs, n, m, make_box3(box), T,
               grid, cell_of, cell_start, sorted_ids, nl_rows_out, first_out, n_nbr_out, g, fcap, fl_out, deg_out, status8, cl);
  } else {
    static const int split_env = [] { const char* e = getenv("CHX_BUILD_SPLIT"); return e ? atoi(e) : 0; }();
    int split = split_env > 0 ? split_env : (8192 + ncell - 1) / ncell;  // (only > 1 when the kernel is forced)
    split = split < 1 ? 1 : (split > 16 ? 16 : split);
    const unsigned cb = (unsigned)(((int64_t)ncell * split * CHX_WAVE + 255) / 256);
    CHX_LAUNCH(half ? k_nl_build_cellwise_fl<true> : k_nl_build_cellwise_fl<false>, dim3(cb), dim3(256), st, xs, n, m,
               make_box3(box), T, grid, ncell, split, cell_start, sorted_ids, nl_rows_out, first_out, n_nbr_out, g, fcap, fl_out,
               deg_out, status8);
  }
  const unsigned sb = (unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
  CHX_LAUNCH(k_build_epilogue, dim3(sb), dim3(256), st, n_nbr_out, deg_out, n, m, status8, publish_host_dst8);
  // (on the step queue the first step launch behind the build need not wait for this summary: it reads n_nbr / deg, which
  // the steps only read, and writes the status block, which they never touch)
  chx_aql_allow_overlap(st);
  return chx_check_launch("chx_nl_rows_cell_fl");
}

extern "C" int chx_nl_build_cell_fl(const float* xyz, int32_t n, int32_t m, chx_box_t box,
                                    float cutoff_plus_skin, const int32_t* nc_h, int32_t* cell_of,
                                    int32_t* cell_count, int32_t* cell_start, uint32_t* sorted_ids,
                                    int32_t* status8, uint32_t* nl_rows_out, uint32_t* first_out,
                                    int32_t* n_nbr_out, int32_t g, int32_t chunks, uint32_t* fl_out,
                                    int32_t* deg_out, float* xs4_out, const chx_build_riders_t* riders,
                                    chx_stream_t stream) {
  // (argument checks of both halves up front: nothing is launched for a call that the second half would refuse)
  CHX_REQUIRE(n >= 1 && m >= 1, "need n >= 1, m >= 1");
  CHX_REQUIRE(xyz && nc_h && cell_of && cell_count && cell_start && sorted_ids && status8 && n_nbr_out && fl_out && deg_out &&
                  xs4_out,
              "null pointer");
  CHX_REQUIRE((nl_rows_out == nullptr) == (first_out == nullptr), "half rows: nl_rows_out and first_out go together");
  CHX_REQUIRE(g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64,
              "g must be a power of two <= 64");
  CHX_REQUIRE(chunks >= 1 && (int64_t)chunks * CHX_FL_QUAD * g < (1 << 20), "bad force-list capacity");
  int rc = chx_cell_bin_fl(xyz, n, box, cutoff_plus_skin, nc_h, cell_of, cell_count, cell_start, sorted_ids, status8, xs4_out,
                           riders, stream);
  if (rc != CHX_OK) return rc;
  return chx_nl_rows_cell_fl(n, m, box, cutoff_plus_skin, nc_h, cell_of, cell_start, sorted_ids, status8, xs4_out,
                             nl_rows_out, first_out, n_nbr_out, g, chunks, fl_out, deg_out,
                             riders ? riders->publish_host_dst8 : nullptr, riders ? riders->cand_list : nullptr,
                             riders ? riders->cand_len : nullptr, riders ? riders->cand_cap : 0, stream);
}

// padding (chiron/neighbors.py:603-620) and list mask (:623) for rows that hold only their real entries
__global__ __launch_bounds__(256) void k_nl_pad(uint32_t* __restrict__ nl, const uint32_t* __restrict__ first,
                                                const int32_t* __restrict__ n_nbr, int n, int m,
                                                int32_t* __restrict__ nl_mask) {
  const int lane = lane_id();
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const int count = n_nbr[i];
  uint32_t fill = count > 0 ? first[i] : 0u;  // argmax of the bool row (:606)
  if (fill == (uint32_t)i) fill += 1u;        // :609
  const int stored = count < m ? count : m;
  for (int t = stored + lane; t < m; t += CHX_WAVE) nl[(size_t)i * m + t] = fill;
  for (int t = lane; t < m; t += CHX_WAVE) nl_mask[(size_t)i * m + t] = t < count ? 1 : 0;
}

extern "C" int chx_nl_finish_padding(uint32_t* nl_inout, const uint32_t* first, const int32_t* n_nbr,
                                     int32_t n, int32_t m, int32_t* nl_mask_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0 && m >= 1, "need n >= 0, m >= 1");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(nl_inout && first && n_nbr && nl_mask_out, "null pointer");
  const unsigned wb = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);
  hipLaunchKernelGGL(k_nl_pad, dim3(wb), dim3(256), 0, chx_stream(stream), nl_inout, first, n_nbr, n, m,
                     nl_mask_out);
  return chx_check_launch("chx_nl_finish_padding");
}

// ---------------------------------------------------------------------------
// one launch to (re)pack the Langevin state into internal order
// ---------------------------------------------------------------------------
__global__ void k_pack_langevin(const float* __restrict__ x3, const float* __restrict__ v3,
                                const float* __restrict__ mass, const float* __restrict__ ref3,
                                const uint32_t* __restrict__ perm, int n, float4* __restrict__ x4,
                                float* __restrict__ v, float* __restrict__ mass_out,
                                float4* __restrict__ ref4, int32_t* __restrict__ ctrl, int ctrl_keep,
                                const int32_t* __restrict__ pub_src, int32_t* __restrict__ pub_host) {
  // riders of the integrator's rebuild (one launch instead of three): the control-block reset of chx_ctrl_reset
  // and the copy of the build's status block to host-mapped memory of chx_publish
  if (blockIdx.x == 0) {
    const int t = threadIdx.x;
    if (ctrl != nullptr && t < CHX_CTRL_WORDS) {
      // (carried-over force list: the reference's event counters / booked events survive a renewal of the list)
      const bool keep = ((ctrl_keep & 1) && (t == 8 || t == 9)) || ((ctrl_keep & 2) && (t == 10 || t == 11));
      if (!keep) ctrl[t] = chx_ctrl_reset_word(t);
    }
    if (pub_src != nullptr && t >= 64 && t < 72)
      __hip_atomic_store(&pub_host[t - 64], pub_src[t - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t o = perm ? perm[p] : (uint32_t)p;
  x4[p] = make_float4(x3[3 * o], x3[3 * o + 1], x3[3 * o + 2], 0.f);
  v[3 * (size_t)p] = v3[3 * o];
  v[3 * (size_t)p + 1] = v3[3 * o + 1];
  v[3 * (size_t)p + 2] = v3[3 * o + 2];
  mass_out[p] = mass[o];
  // (.w: the particle's mass -- the fused step kernel reads it from here instead of loading mass[] as well)
  if (ref4) ref4[p] = make_float4(ref3[3 * o], ref3[3 * o + 1], ref3[3 * o + 2], mass[o]);
}

extern "C" int chx_pack_langevin(const float* x3, const float* v3, const float* mass,
                                 const float* ref3, const uint32_t* perm, int32_t n, float* x4_out,
                                 float* v_out, float* mass_out, float* ref4_out, int32_t* ctrl_reset,
                                 int32_t ctrl_keep, const int32_t* publish_src8, int32_t* publish_host_dst8,
                                 chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  CHX_REQUIRE((publish_src8 == nullptr) == (publish_host_dst8 == nullptr), "publish needs both a source and a destination");
  if (n == 0 && !ctrl_reset && !publish_src8) return CHX_OK;
  CHX_REQUIRE(n == 0 || (x3 && v3 && mass && x4_out && v_out && mass_out), "null pointer");
  CHX_REQUIRE(!ref4_out || ref3, "ref4_out without ref3");
  const unsigned blocks = n > 0 ? (unsigned)((n + 255) / 256) : 1u;
  CHX_LAUNCH(k_pack_langevin, dim3(blocks), dim3(256), chx_stream_q(stream), x3, v3, mass, ref3, perm, n,
             reinterpret_cast<float4*>(x4_out), v_out, mass_out, reinterpret_cast<float4*>(ref4_out), ctrl_reset, (int)ctrl_keep,
             publish_src8, publish_host_dst8);
  return chx_check_launch("chx_pack_langevin");
}

// The pack launch of a run that starts on a carried-over force list, with chx_validate_snap's question folded in: is every
// particle within skin / 2 of the position its force list was built from (snap4, internal order)?  The answer has to
// land in the halt word, which this very launch resets -- so the reset moves to whichever workgroup finishes LAST
// (ticket in sync2[0]; a workgroup that saw a violation raises sync2[1] before it takes its ticket): that workgroup
// writes the control block with the halt word already decided and leaves sync2 zeroed for the next call.
__global__ void k_pack_validate_langevin(const float* __restrict__ x3, const float* __restrict__ v3,
                                         const float* __restrict__ mass, const float* __restrict__ ref3,
                                         const uint32_t* __restrict__ perm, int n, float4* __restrict__ x4,
                                         float* __restrict__ v, float* __restrict__ mass_out, float4* __restrict__ ref4,
                                         int32_t* __restrict__ ctrl, int ctrl_keep, const float4* __restrict__ snap4,
                                         Box3 box, float T_half_skin, int step, int32_t* __restrict__ host_ctrl,
                                         int32_t* __restrict__ sync2) {
  __shared__ int sh_last;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  int bad = 0;
  if (p < n) {
    const size_t o = perm ? perm[p] : (uint32_t)p;
    const float x = x3[3 * o], y = x3[3 * o + 1], z = x3[3 * o + 2];
    x4[p] = make_float4(x, y, z, 0.f);
    v[3 * (size_t)p] = v3[3 * o];
    v[3 * (size_t)p + 1] = v3[3 * o + 1];
    v[3 * (size_t)p + 2] = v3[3 * o + 2];
    mass_out[p] = mass[o];
    if (ref4) ref4[p] = make_float4(ref3[3 * o], ref3[3 * o + 1], ref3[3 * o + 2], mass[o]);
    const float4 s = snap4[p];
    float rx, ry, rz;
    displacement(box, x, y, z, s.x, s.y, s.z, rx, ry, rz);
    bad = !(norm2(rx, ry, rz) <= T_half_skin);  // (NaN positions fail the negated comparison: they halt too)
  }
  const int any_bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    if (any_bad) atomicOr(&sync2[1], 1);
    __threadfence();
    sh_last = atomicAdd(&sync2[0], 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!sh_last) return;
  __threadfence();
  const int t = threadIdx.x;
  const int failed = __hip_atomic_load(&sync2[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t < CHX_CTRL_WORDS) {
    const bool keep = ((ctrl_keep & 1) && (t == 8 || t == 9)) || ((ctrl_keep & 2) && (t == 10 || t == 11));
    if (!keep) ctrl[t] = (t == 3 && failed) ? step : chx_ctrl_reset_word(t);
  }
  if (t == 64 && failed && host_ctrl != nullptr)
    __hip_atomic_store(&host_ctrl[3], step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (t == 65) {
    __hip_atomic_store(&sync2[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sync2[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int chx_pack_validate_langevin(chx_box_t box, float skin, int32_t step, const float* snap4, int32_t* host_ctrl,
                                          int32_t* sync2, const float* x3, const float* v3, const float* mass,
                                          const float* ref3, const uint32_t* perm, int32_t n, float* x4_out, float* v_out,
                                          float* mass_out, float* ref4_out, int32_t* ctrl_reset, int32_t ctrl_keep,
                                          chx_stream_t stream) {
  CHX_REQUIRE(n >= 1, "n < 1");
  CHX_REQUIRE(x3 && v3 && mass && x4_out && v_out && mass_out && snap4 && sync2 && ctrl_reset, "null pointer");
  CHX_REQUIRE(!ref4_out || ref3, "ref4_out without ref3");
  CHX_REQUIRE(skin > 0.f && step >= 0, "need skin > 0 and step >= 0");
  CHX_LAUNCH(k_pack_validate_langevin, dim3((unsigned)((n + 255) / 256)), dim3(256), chx_stream_q(stream), x3, v3, mass, ref3,
             perm, n, reinterpret_cast<float4*>(x4_out), v_out, mass_out, reinterpret_cast<float4*>(ref4_out), ctrl_reset,
             (int)ctrl_keep, reinterpret_cast<const float4*>(snap4), make_box3(box), chx_cutoff_threshold_h(skin * 0.5f),
             (int)step, host_ctrl, sync2);
  return chx_check_launch("chx_pack_validate_langevin");
}

// ---------------------------------------------------------------------------
// list carry-over: may launch `step` walk a force list that was built from other positions?  (include/chx.h:
// chx_validate_snap)
// ---------------------------------------------------------------------------
__global__ void k_validate_snap(const float4* __restrict__ x4, const float4* __restrict__ snap4, int n, Box3 box,
                                float T_half_skin, int step, int32_t* __restrict__ ctrl, int32_t* __restrict__ host_ctrl) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float4 x = x4[p], s = snap4[p];
  float rx, ry, rz;
  displacement(box, x.x, x.y, x.z, s.x, s.y, s.z, rx, ry, rz);
  // (NaN positions fail the comparison's negation: they halt too, and the rebuild reports them)
  if (!(norm2(rx, ry, rz) <= T_half_skin)) {
    atomicMin(&ctrl[3], step);
    if (host_ctrl != nullptr) __hip_atomic_store(&host_ctrl[3], step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

extern "C" int chx_validate_snap(const float* x4, const float* snap4, int32_t n, chx_box_t box, float skin, int32_t step,
                                 int32_t* ctrl, int32_t* host_ctrl, chx_stream_t stream) {
  CHX_REQUIRE(n >= 1, "n < 1");
  CHX_REQUIRE(x4 && snap4 && ctrl, "null pointer");
  CHX_REQUIRE(skin > 0.f && step >= 0, "need skin > 0 and step >= 0");
  CHX_LAUNCH(k_validate_snap, dim3((unsigned)((n + 255) / 256)), dim3(256), chx_stream_q(stream),
             reinterpret_cast<const float4*>(x4), reinterpret_cast<const float4*>(snap4), n, make_box3(box),
             chx_cutoff_threshold_h(skin * 0.5f), (int)step, ctrl, host_ctrl);
  return chx_check_launch("chx_validate_snap");
}
