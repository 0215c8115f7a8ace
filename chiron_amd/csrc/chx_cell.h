// Cell-grid helpers shared by the binning kernels (chx_neighbors.hip) and the row kernels (chx_cell_fl.hip).
// Internal, device code only.
#pragma once
#include "chx_common.h"
#include "chx_internal.h"

// set bits of a ballot below this lane (v_mbcnt: two instructions instead of mask, and, two popcounts)
__device__ __forceinline__ int lanes_below(unsigned long long ballot) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0u));
}

// The 27 cells around cell (cx,cy,cz) are at most 18 contiguous index ranges of the cell-ordered arrays (the three
// x-neighbours of a (y,z) row are adjacent except across the periodic seam).  Lane r < 18 fetches the bounds of
// range r -- all 18 in one round trip -- and a wave-wide prefix sum flattens them into one candidate index space
// [0, total): rng_s[r] = first array index of range r, rng_o[r] = flat index of its first candidate (LDS, per wave).
__device__ __forceinline__ int cell_ranges(const CellGrid& g, int cx, int cy, int cz, const int32_t* __restrict__ cell_start,
                                           int lane, int* rng_s, int* rng_o) {
  int rs = 0, re = 0;
  if (lane < 18) {
    const int combo = lane >> 1, sub = lane & 1;
    int zz = cz + combo / 3 - 1, yy = cy + combo % 3 - 1;  // periodic wrap without a division
    zz = zz < 0 ? zz + g.ncz : (zz >= g.ncz ? zz - g.ncz : zz);
    yy = yy < 0 ? yy + g.ncy : (yy >= g.ncy ? yy - g.ncy : yy);
    const int row = (zz * g.ncy + yy) * g.ncx;
    int a0 = 0, a1 = 0;
    if (g.ncx == 3) {
      if (sub == 0) { a0 = row; a1 = row + 3; }
    } else if (cx == 0) {
      if (sub == 0) { a0 = row + g.ncx - 1; a1 = row + g.ncx; } else { a0 = row; a1 = row + 2; }
    } else if (cx == g.ncx - 1) {
      if (sub == 0) { a0 = row + cx - 1; a1 = row + cx + 1; } else { a0 = row; a1 = row + 1; }
    } else {
      if (sub == 0) { a0 = row + cx - 1; a1 = row + cx + 2; }
    }
    if (a1 > a0) {
      rs = cell_start[a0];
      re = cell_start[a1];
    }
  }
  const int len = re - rs;
  int incl = len;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int v = __shfl_up(incl, o, CHX_WAVE);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 17, CHX_WAVE);
  if (lane < 18) {
    rng_s[lane] = rs;
    rng_o[lane] = incl - len;
  }
  __threadfence_block();
  return total;
}

// Per-cell candidate index lists (round 4).  The ~10 particles of a cell share their 27-cell neighbourhood, and so far
// every ROW worked the neighbourhood's (up to 18) index ranges out for itself and then searched them for each batch of
// candidates -- a third of the row kernel's instructions.  k_cell_sort (one wavefront per cell) now writes the
// neighbourhood out once per cell as an explicit list of candidate indices into the cell-ordered arrays, in the order
// the rows scan them; a row reads its cell's list with coalesced loads.  cand_len[c] = length; a neighbourhood longer
// than the capacity keeps the range search (the row kernels check).
struct CellRangeLoads {  // what cell_candidates_begin has in flight: lane r < 18 holds range r's bounds
  int rs, re;
};
// the loads: nothing here waits, so the caller can put other loads behind them before it calls cell_candidates_finish
__device__ __forceinline__ CellRangeLoads cell_candidates_begin(const CellGrid& g, int cx, int cy, int cz,
                                                                const int32_t* __restrict__ cell_start, int lane) {
  CellRangeLoads L = {0, 0};
  if (lane < 18) {
    const int combo = lane >> 1, sub = lane & 1;
    int zz = cz + combo / 3 - 1, yy = cy + combo % 3 - 1;
    zz = zz < 0 ? zz + g.ncz : (zz >= g.ncz ? zz - g.ncz : zz);
    yy = yy < 0 ? yy + g.ncy : (yy >= g.ncy ? yy - g.ncy : yy);
    const int row = (zz * g.ncy + yy) * g.ncx;
    int a0 = 0, a1 = 0;
    if (g.ncx == 3) {
      if (sub == 0) { a0 = row; a1 = row + 3; }
    } else if (cx == 0) {
      if (sub == 0) { a0 = row + g.ncx - 1; a1 = row + g.ncx; } else { a0 = row; a1 = row + 2; }
    } else if (cx == g.ncx - 1) {
      if (sub == 0) { a0 = row + cx - 1; a1 = row + cx + 1; } else { a0 = row; a1 = row + 1; }
    } else {
      if (sub == 0) { a0 = row + cx - 1; a1 = row + cx + 2; }
    }
    if (a1 > a0) {
      L.rs = cell_start[a0];
      L.re = cell_start[a1];
    }
  }
  return L;
}
// the ranges as cell_ranges forms them, kept in the registers of lanes 0..17 (no LDS round trips: this wavefront is one
// of a few thousand and lives on latency), flattened and written out
__device__ __forceinline__ void cell_candidates_finish(const CellRangeLoads& L, int ci, int lane,
                                                       uint32_t* __restrict__ cand_list,
                                                       int32_t* __restrict__ cand_len, int cap) {
  const int rs = L.rs;
  const int len = L.re - L.rs;
  int incl = len;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int v = __shfl_up(incl, o, CHX_WAVE);
    if (lane >= o) incl += v;
  }
  const int total = __builtin_amdgcn_readlane(incl, 17);
  if (lane == 0) cand_len[ci] = total;
  if (total > cap) return;
  uint32_t* out = cand_list + (size_t)ci * cap;
  const int off = incl - len;
  // Full-width, line-aligned stores (rows of the list start on 256-byte boundaries): lane l of batch q writes flat entry
  // t = 64 q + l, which lies in the range r(t) = #{r : end(r) <= t}; 17 broadcast compares find it.
#pragma unroll 1
  for (int t0 = 0; t0 < total; t0 += CHX_WAVE) {
    const int t = t0 + lane;
    int r = 0;
#pragma unroll 1  // (compact code: a few-microsecond kernel pays for every instruction-cache line it touches cold)
    for (int k = 0; k < 17; ++k) r += t >= __builtin_amdgcn_readlane(incl, k) ? 1 : 0;
    const int s0 = __shfl(rs, r, CHX_WAVE), o0 = __shfl(off, r, CHX_WAVE);
    if (t < total) out[t] = (uint32_t)(s0 + (t - o0));
  }
}
