// libchx: NeighborListNsqrd build (O(N^2) and cell list, bit-identical outputs),
// calculate, check; PairListNsqrd build / calculate.
//
// Every distance predicate is the reference's `sqrt(fl(rx^2+ry^2+rz^2)) < c` on
// fp32 min-imaged differences (chiron/neighbors.py:79-81, :600-602), evaluated
// as `r2 <= T` with T = chx_cutoff_threshold_h(c) -- exactly equivalent because
// the correctly-rounded sqrt is monotone.
#include <math.h>

#include "chx_common.h"
#include "chx_launch.h"
#include "chx_internal.h"
#include "chx_cell.h"

// ---------------------------------------------------------------------------
// shared row epilogue: padding (chiron/neighbors.py:603-620) and mask (:623)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void nl_finish_row(uint32_t* __restrict__ nl_row,
                                              int32_t* __restrict__ mask_row, int m, int count,
                                              uint32_t first, uint32_t i, int lane) {
  uint32_t fill = count > 0 ? first : 0u;  // argmax of the bool row (:606)
  if (fill == i) fill += 1u;               // :609
  int stored = count < m ? count : m;
  for (int t = stored + lane; t < m; t += CHX_WAVE) nl_row[t] = fill;
  for (int t = lane; t < m; t += CHX_WAVE) mask_row[t] = t < count ? 1 : 0;
}

// ---------------------------------------------------------------------------
// O(N^2) build: one wavefront per row i, lanes sweep j = i+1 .. n-1 in order,
// ballot + prefix popcount emits the ascending list directly.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nl_build_nsq(const float* __restrict__ xyz, int n, int m,
                                                      Box3 box, float T, uint32_t* __restrict__ nl,
                                                      int32_t* __restrict__ nl_mask,
                                                      int32_t* __restrict__ n_nbr) {
  const int lane = lane_id();
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const float xi = xyz[3 * (size_t)i], yi = xyz[3 * (size_t)i + 1], zi = xyz[3 * (size_t)i + 2];
  uint32_t* nl_row = nl + (size_t)i * m;
  int count = 0;
  uint32_t first = 0;
  for (int j0 = i + 1; j0 < n; j0 += CHX_WAVE) {
    const int j = j0 + lane;
    bool ok = false;
    if (j < n) {
      float rx, ry, rz;
      displacement(box, xi, yi, zi, xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1],
                   xyz[3 * (size_t)j + 2], rx, ry, rz);
      ok = norm2(rx, ry, rz) <= T;
    }
    const unsigned long long b = __ballot(ok);
    if (b) {
      if (count == 0) first = (uint32_t)(j0 + __ffsll((long long)b) - 1);
      const int slot = count + __popcll(b & ((1ull << lane) - 1ull));
      if (ok && slot < m) nl_row[slot] = (uint32_t)j;
      count += __popcll(b);
    }
  }
  nl_finish_row(nl_row, nl_mask + (size_t)i * m, m, count, first, (uint32_t)i, lane);
  if (lane == 0) n_nbr[i] = count;
}

extern "C" int chx_nl_build_nsq(const float* xyz, int32_t n, int32_t m, chx_box_t box,
                                float cutoff_plus_skin, uint32_t* nl_out, int32_t* nl_mask_out,
                                int32_t* n_nbr_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0 && m >= 1, "need n >= 0, m >= 1");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(xyz && nl_out && nl_mask_out && n_nbr_out, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  const float T = chx_cutoff_threshold_h(cutoff_plus_skin);
  const unsigned blocks = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);
  hipLaunchKernelGGL(k_nl_build_nsq, dim3(blocks), dim3(256), 0, chx_stream(stream), xyz, n, m,
                     make_box3(box), T, nl_out, nl_mask_out, n_nbr_out);
  return chx_check_launch("chx_nl_build_nsq");
}

// ---------------------------------------------------------------------------
// cell-list build
// ---------------------------------------------------------------------------
extern "C" int64_t chx_cell_grid_h(chx_box_t box, float cs, int32_t n, int32_t* nc_h) {
  if (!nc_h) return 0;
  nc_h[0] = nc_h[1] = nc_h[2] = 0;
  if (!box.periodic || !(cs > 0.0f) || n <= 0) return 0;
  const float L[3] = {box.lx, box.ly, box.lz};
  int64_t total = 1;
  for (int d = 0; d < 3; ++d) {
    if (!(L[d] > 0.0f)) return 0;
    // cell edge >= cs with a margin that dominates every fp32 rounding in the
    // wrapped coordinate, the cell index and the min-image distance
    const double ulpL = (double)(nextafterf(L[d], INFINITY) - L[d]);
    const double w = (double)cs * 1.0001 + 64.0 * ulpL;
    int64_t nc = (int64_t)floor((double)L[d] / w);
    if (nc < 3) return 0;
    if (nc > 1024) nc = 1024;
    nc_h[d] = (int32_t)nc;
    total *= nc;
  }
  // do not let an almost-empty grid dominate the workspace
  while (total > 8 * (int64_t)n + 4096) {
    int big = 0;
    for (int d = 1; d < 3; ++d)
      if (nc_h[d] > nc_h[big]) big = d;
    if (nc_h[big] <= 3) break;
    nc_h[big] -= 1;
    total = (int64_t)nc_h[0] * nc_h[1] * nc_h[2];
  }
  return total;
}

__device__ __forceinline__ int cell_coord(float x, float L, float s, int nc) {
  float w = wrap1(x, L);
  int c = (int)floorf(w * s);
  c = c < 0 ? 0 : c;
  return c >= nc ? nc - 1 : c;
}

// (cell_count is all zero on entry: the previous chain's k_cell_sort left it so -- no zeroing launch in front)
// slots != NULL: the arrival order the counting atomic returns is kept -- particle i goes to slots[c * slot_cap + order]
// (an order beyond the capacity is dropped here and reported by k_cell_sort, which sees the cell's count) -- and no
// scatter launch is needed: k_cell_sort reads a cell's ids from its slots.
__global__ void k_cell_assign(const float* __restrict__ xyz, int n, Box3 box, CellGrid g,
                              int32_t* __restrict__ cell_of, int32_t* __restrict__ cell_count,
                              float* __restrict__ xyz_copy, uint32_t* __restrict__ slots, int slot_cap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
  if (xyz_copy != nullptr) {  // the list's private copy of its reference positions rides along
    xyz_copy[3 * (size_t)i] = x;
    xyz_copy[3 * (size_t)i + 1] = y;
    xyz_copy[3 * (size_t)i + 2] = z;
  }
  int cx = cell_coord(x, box.lx, g.sx, g.ncx);
  int cy = cell_coord(y, box.ly, g.sy, g.ncy);
  int cz = cell_coord(z, box.lz, g.sz, g.ncz);
  int c = (cz * g.ncy + cy) * g.ncx + cx;
  cell_of[i] = c;
  const int order = atomicAdd(&cell_count[c], 1);
  if (slots != nullptr && order < slot_cap) slots[(size_t)c * slot_cap + order] = (uint32_t)i;
}

// single-block exclusive scan: cell_start[0..ncell] from cell_count; zeroes cell_count (reused as cursor).
// Every wavefront owns one contiguous segment of the cells (a multiple of 64): pass 1 adds the segment up with
// independent coalesced loads, ONE barrier hands the 16 segment totals around, pass 2 rereads the segment (L2) and
// writes the running prefix with wave-shuffle scans, 256 counts per trip (four independent loads per lane).  (The tile-by-tile form this replaces -- 1024 cells per trip, three barriers and one exposed load
// latency per trip -- took 30 us for the 27 000 cells of N = 262 144.)
__global__ __launch_bounds__(1024) void k_cell_scan(int32_t* __restrict__ cell_count,
                                                    int32_t* __restrict__ cell_start, int ncell,
                                                    int32_t* __restrict__ status, int status_len,
                                                    int32_t* __restrict__ ctrl, int ctrl_keep) {
  __shared__ int32_t wtot[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t < status_len) status[t] = 0;  // (the build kernels behind this one accumulate into it: no memset launch)
  // rider of the integrator's rebuild chain: the control-block reset of chx_ctrl_reset (ctrl_keep: chx_pack_langevin)
  if (ctrl != nullptr && t >= 512 && t < 512 + CHX_CTRL_WORDS) {
    const int w = t - 512;
    const bool keep = ((ctrl_keep & 1) && (w == 8 || w == 9)) || ((ctrl_keep & 2) && (w == 10 || w == 11));
    if (!keep) ctrl[w] = chx_ctrl_reset_word(w);
  }
  const int seg = ((ncell + 15) / 16 + CHX_WAVE - 1) & ~(CHX_WAVE - 1);
  const int s0 = wv * seg < ncell ? wv * seg : ncell;
  const int s1 = s0 + seg < ncell ? s0 + seg : ncell;
  int32_t acc = 0;
  for (int b = s0; b < s1; b += 4 * CHX_WAVE) {  // four independent loads in flight per lane
    int32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = b + q * CHX_WAVE + lane;
      v[q] = c < s1 ? cell_count[c] : 0;
    }
    acc += (v[0] + v[1]) + (v[2] + v[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, CHX_WAVE);
  if (lane == 0) wtot[wv] = acc;
  __syncthreads();
  int32_t carry = 0;
  for (int k = 0; k < wv; ++k) carry += wtot[k];
  for (int b = s0; b < s1; b += 4 * CHX_WAVE) {
    int32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = b + q * CHX_WAVE + lane;
      v[q] = c < s1 ? cell_count[c] : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = b + q * CHX_WAVE + lane;
      int32_t incl = v[q];
#pragma unroll
      for (int o = 1; o < CHX_WAVE; o <<= 1) {
        const int32_t u = __shfl_up(incl, o, CHX_WAVE);
        if (lane >= o) incl += u;
      }
      if (c < s1) {
        cell_start[c] = carry + incl - v[q];
        cell_count[c] = 0;
      }
      carry += __shfl(incl, CHX_WAVE - 1, CHX_WAVE);
    }
  }
  if (t == 1023) cell_start[ncell] = carry;  // (the last wavefront's segment ends at ncell, or is empty: carry = total)
}

__global__ void k_cell_scatter(const int32_t* __restrict__ cell_of, int n,
                               const int32_t* __restrict__ cell_start,
                               int32_t* __restrict__ cursor, uint32_t* __restrict__ sorted_ids) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = cell_of[i];
  int pos = cell_start[c] + atomicAdd(&cursor[c], 1);
  sorted_ids[pos] = (uint32_t)i;
}

// ascending particle id inside each cell: makes the internal order (and with it
// every floating-point summation order downstream) reproducible run to run.
// One wavefront per cell: every lane holds up to CHX_SORT_PER_LANE ids in registers, ranks
// them against the whole cell through wave shuffles and writes them back in place by rank.
// Cells with more particles than that (reduced densities far above a liquid's) fall back
// to a serial insertion sort by lane 0.
#define CHX_SORT_PER_LANE 4
// xs4 != NULL: also emit the positions in cell order, .w = the cell's packed grid coordinates as int bits (what the one-pass build reads:
// its candidate loads are then coalesced and it needs no dependent cell_of[sorted_ids[w]] round trip).
__device__ __forceinline__ void cell_emit_sorted(const float* __restrict__ xyz, float4* __restrict__ xs4, int at, uint32_t id, int c,
                                                 const BinRiders& rd) {
  const float x = xyz[3 * (size_t)id], y = xyz[3 * (size_t)id + 1], z = xyz[3 * (size_t)id + 2];
  xs4[at] = make_float4(x, y, z, __int_as_float(c));
  if (rd.x4 != nullptr) {  // the Langevin state in the new internal order (what chx_pack_langevin writes)
    const float m = rd.mass[id];
    rd.x4[at] = make_float4(x, y, z, 0.f);
    rd.v[3 * (size_t)at] = rd.v3[3 * (size_t)id];
    rd.v[3 * (size_t)at + 1] = rd.v3[3 * (size_t)id + 1];
    rd.v[3 * (size_t)at + 2] = rd.v3[3 * (size_t)id + 2];
    rd.mass_out[at] = m;
    if (rd.ref4 != nullptr) {
      if (rd.ref3 != nullptr) rd.ref4[at] = make_float4(rd.ref3[3 * (size_t)id], rd.ref3[3 * (size_t)id + 1], rd.ref3[3 * (size_t)id + 2], m);
      else rd.ref4[at] = make_float4(x, y, z, m);
    }
  }
}

__global__ __launch_bounds__(256) void k_cell_sort(const int32_t* __restrict__ cell_start, int ncell,
                                                   uint32_t* __restrict__ sorted_ids, const float* __restrict__ xyz,
                                                   float4* __restrict__ xs4, int ncx, int ncy,
                                                   int32_t* __restrict__ cell_count, BinRiders rd, CellGrid g,
                                                   uint32_t* __restrict__ cand_list, int32_t* __restrict__ cand_len,
                                                   int cand_cap, const uint32_t* __restrict__ slots, int slot_cap,
                                                   int32_t* __restrict__ status) {
  const int lane = lane_id();
  int ci = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ci >= ncell) {
    // The second half of the grid (launched only when the lists are wanted): one wavefront per cell writes the candidate
    // indices of the cell's 27-cell neighbourhood, once for all of its rows (chx_cell.h).  Wavefronts of their own, not a
    // second job of the sorting wavefronts: either job is a short chain of dependent steps, and a few thousand wavefronts
    // leave the chip mostly idle -- side by side the two chains cost the longer one, in sequence their sum (measured).
    ci -= ncell;
    if (cand_list == nullptr || ci >= ncell || cell_start[ci + 1] <= cell_start[ci]) return;  // (an empty cell has no rows)
    const CellRangeLoads ranges = cell_candidates_begin(g, ci % ncx, (ci / ncx) % ncy, ci / (ncx * ncy), cell_start, lane);
    cell_candidates_finish(ranges, ci, lane, cand_list, cand_len, cand_cap);
    return;
  }
  // the scatter's cursors are done with: leave the counters zeroed for the next build (no zeroing launch there)
  if (lane == 0) cell_count[ci] = 0;
  // what goes into xs4[].w: the cell's grid coordinates, 10 bits each (chx_cell_grid_h keeps every axis <= 1024 cells),
  // so the build kernel's rows need no integer divisions to find their 27 neighbours
  const int c = (ci % ncx) | (((ci / ncx) % ncy) << 10) | ((ci / (ncx * ncy)) << 20);
  const int s = cell_start[ci], e = cell_start[ci + 1];
  const int np = e - s;
  if (slots != nullptr) {
    // the cell's ids wait in its slots (k_cell_assign), in arrival order: bring them to their place in sorted_ids for
    // the paths below that sort in place.  More ids than slots: the build is void (status[5]; the caller repeats it
    // without slots) -- the lost entries get a valid id, so that whatever still runs on this list reads valid memory.
    const uint32_t* mine = slots + (size_t)ci * slot_cap;
    if (np > slot_cap && lane == 0) atomicOr(&status[5], 1);
    if (np <= 1 || np > CHX_WAVE * CHX_SORT_PER_LANE || np > slot_cap) {
      for (int t = lane; t < np; t += CHX_WAVE) sorted_ids[s + t] = mine[t < slot_cap ? t : 0];
      __threadfence_block();
    }
  }
  if (np <= 1) {
    if (xs4 != nullptr && np == 1 && lane == 0) cell_emit_sorted(xyz, xs4, s, sorted_ids[s], c, rd);
    return;
  }
  if (np > CHX_WAVE * CHX_SORT_PER_LANE || (slots != nullptr && np > slot_cap)) {
    if (lane == 0) {
      for (int a = s + 1; a < e; ++a) {
        const uint32_t v = sorted_ids[a];
        int b = a - 1;
        while (b >= s && sorted_ids[b] > v) {
          sorted_ids[b + 1] = sorted_ids[b];
          --b;
        }
        sorted_ids[b + 1] = v;
      }
    }
    if (xs4 != nullptr) {
      __threadfence_block();  // lane 0's stores, read back by the whole wavefront
      for (int t = lane; t < np; t += CHX_WAVE) cell_emit_sorted(xyz, xs4, s + t, sorted_ids[s + t], c, rd);
    }
    return;
  }
  uint32_t mine[CHX_SORT_PER_LANE];
  int rank[CHX_SORT_PER_LANE];
#pragma unroll
  for (int k = 0; k < CHX_SORT_PER_LANE; ++k) {
    const int t = k * CHX_WAVE + lane;
    // ids are distinct and < 2^32 - 1
    mine[k] = t < np ? (slots != nullptr ? slots[(size_t)ci * slot_cap + t] : sorted_ids[s + t]) : 0xFFFFFFFFu;
    rank[k] = 0;
  }
  const int rounds = (np + CHX_WAVE - 1) / CHX_WAVE;
  for (int kb = 0; kb < rounds; ++kb) {
    // lane-uniform choice of the register that is broadcast this round
    uint32_t src = mine[0];
#pragma unroll
    for (int k = 1; k < CHX_SORT_PER_LANE; ++k) src = kb == k ? mine[k] : src;
    const int count = np - kb * CHX_WAVE < CHX_WAVE ? np - kb * CHX_WAVE : CHX_WAVE;
    for (int l = 0; l < count; ++l) {
      const uint32_t other = __shfl(src, l, CHX_WAVE);
#pragma unroll
      for (int k = 0; k < CHX_SORT_PER_LANE; ++k) rank[k] += other < mine[k] ? 1 : 0;
    }
  }
  // every id was read into registers above before any lane writes
#pragma unroll
  for (int k = 0; k < CHX_SORT_PER_LANE; ++k) {
    if (k * CHX_WAVE + lane < np) {
      sorted_ids[s + rank[k]] = mine[k];
      if (xs4 != nullptr) cell_emit_sorted(xyz, xs4, s + rank[k], mine[k], c, rd);
    }
  }
}

#define CHX_CELL_CAP 1024  // candidates kept on chip per row

__global__ __launch_bounds__(256) void k_nl_build_cell(
    const float* __restrict__ xyz, int n, int m, Box3 box, float T, CellGrid g,
    const int32_t* __restrict__ cell_of, const int32_t* __restrict__ cell_start,
    const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ nl,
    int32_t* __restrict__ nl_mask, int32_t* __restrict__ n_nbr, int32_t* __restrict__ status) {
  __shared__ uint32_t cand_all[4][CHX_CELL_CAP];
  const int lane = lane_id();
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= n) return;
  uint32_t* cand = cand_all[threadIdx.x >> 6];
  const uint32_t i = sorted_ids[w];
  const float xi = xyz[3 * (size_t)i], yi = xyz[3 * (size_t)i + 1], zi = xyz[3 * (size_t)i + 2];
  const int c = cell_of[i];
  const int cx = c % g.ncx, cy = (c / g.ncx) % g.ncy, cz = c / (g.ncx * g.ncy);
  int count = 0;
  for (int dz = -1; dz <= 1; ++dz) {
    const int zz = (cz + dz + g.ncz) % g.ncz;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = (cy + dy + g.ncy) % g.ncy;
      const int row = (zz * g.ncy + yy) * g.ncx;
      // the three x-adjacent cells are contiguous in memory except across the periodic seam
      int r0[2], r1[2], nr;
      if (g.ncx == 3) {
        r0[0] = row; r1[0] = row + 3; nr = 1;
      } else if (cx == 0) {
        r0[0] = row + g.ncx - 1; r1[0] = row + g.ncx; r0[1] = row; r1[1] = row + 2; nr = 2;
      } else if (cx == g.ncx - 1) {
        r0[0] = row + cx - 1; r1[0] = row + cx + 1; r0[1] = row; r1[1] = row + 1; nr = 2;
      } else {
        r0[0] = row + cx - 1; r1[0] = row + cx + 2; nr = 1;
      }
      for (int q = 0; q < nr; ++q) {
        const int s = cell_start[r0[q]], e = cell_start[r1[q]];
        for (int k0 = s; k0 < e; k0 += CHX_WAVE) {
          const int k = k0 + lane;
          bool ok = false;
          uint32_t j = 0;
          if (k < e) {
            j = sorted_ids[k];
            if (j > i) {
              float rx, ry, rz;
              displacement(box, xi, yi, zi, xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1],
                           xyz[3 * (size_t)j + 2], rx, ry, rz);
              ok = norm2(rx, ry, rz) <= T;
            }
          }
          const unsigned long long b = __ballot(ok);
          if (b) {
            const int slot = count + __popcll(b & ((1ull << lane) - 1ull));
            if (ok && slot < CHX_CELL_CAP) cand[slot] = j;
            count += __popcll(b);
          }
        }
      }
    }
  }
  if (count > CHX_CELL_CAP && lane == 0) atomicOr(&status[0], 1);
  const int kept = count < CHX_CELL_CAP ? count : CHX_CELL_CAP;
  __threadfence_block();  // candidate stores visible to the whole wavefront
  // rank sort (ids are distinct): ascending j, as jnp.where(mask, size=M) returns them
  uint32_t* nl_row = nl + (size_t)i * m;
  uint32_t mn = 0xFFFFFFFFu;
  for (int t = lane; t < kept; t += CHX_WAVE) {
    const uint32_t v = cand[t];
    mn = v < mn ? v : mn;
    int rank = 0;
    for (int u = 0; u < kept; ++u) rank += cand[u] < v ? 1 : 0;
    if (rank < m) nl_row[rank] = v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t other = __shfl_xor(mn, o, CHX_WAVE);
    mn = other < mn ? other : mn;
  }
  nl_finish_row(nl_row, nl_mask + (size_t)i * m, m, count, mn, i, lane);
  if (lane == 0) n_nbr[i] = count;
}

// Bin the particles: cell_of, cell_start (exclusive scan) and sorted_ids (cell order, ascending id
// inside a cell).  Shared by both cell-list builders.  status[0..7] is zeroed.
int chx_cell_prepare(const float* xyz, int n, chx_box_t box, float cutoff_plus_skin,
                     const int32_t* nc_h, int32_t* cell_of, int32_t* cell_count, int32_t* cell_start,
                     uint32_t* sorted_ids, int32_t* status, int status_len, CellGrid* g_out,
                     hipStream_t st, float* xs4_out, const BinRiders* riders, uint32_t* cand_list, int32_t* cand_len,
                     int cand_cap) {
  CHX_REQUIRE(box.periodic && box.lx > 0 && box.ly > 0 && box.lz > 0, "periodic box required");
  CHX_REQUIRE(nc_h[0] >= 3 && nc_h[1] >= 3 && nc_h[2] >= 3, "need >= 3 cells per axis");
  // k_cell_sort packs a cell's grid coordinates into 10 bits per axis (xs4.w) and the build kernels decode them
  // with & 1023: a longer axis would alias cells silently (chx_cell_grid_h never produces one)
  CHX_REQUIRE(nc_h[0] <= 1024 && nc_h[1] <= 1024 && nc_h[2] <= 1024, "at most 1024 cells per axis");
  const int64_t ncell64 = (int64_t)nc_h[0] * nc_h[1] * nc_h[2];
  CHX_REQUIRE(ncell64 < (1ll << 30), "too many cells");
  // the grid must respect the geometric guarantee (edge >= cutoff + skin)
  CHX_REQUIRE(box.lx / nc_h[0] >= cutoff_plus_skin && box.ly / nc_h[1] >= cutoff_plus_skin &&
                  box.lz / nc_h[2] >= cutoff_plus_skin,
              "cell edge smaller than cutoff + skin");
  const int ncell = (int)ncell64;
  const Box3 b3 = make_box3(box);
  CellGrid g;
  g.ncx = nc_h[0]; g.ncy = nc_h[1]; g.ncz = nc_h[2];
  g.sx = (float)nc_h[0] / box.lx; g.sy = (float)nc_h[1] / box.ly; g.sz = (float)nc_h[2] / box.lz;
  *g_out = g;
  CHX_REQUIRE(status_len >= 0 && status_len <= 1024, "status block too long");
  BinRiders rd = {};
  if (riders) rd = *riders;
  CHX_REQUIRE(!rd.x4 || (xs4_out && rd.v3 && rd.mass && rd.v && rd.mass_out), "pack rider: null pointer");
  // (no zeroing launch: cell_count arrives zeroed -- the caller's allocation, then k_cell_sort of the previous chain)
  const unsigned pb = (unsigned)((n + 255) / 256);
  const bool slotted = rd.cell_slots != nullptr && rd.cell_slot_cap > 0 && status_len >= 8;
  // (CHX_LAUNCH: on the step queue while a run() has packets outstanding there, chx_launch.h)
  CHX_LAUNCH(k_cell_assign, dim3(pb), dim3(256), st, xyz, n, b3, g, cell_of, cell_count, rd.xyz_copy,
             slotted ? rd.cell_slots : nullptr, rd.cell_slot_cap);
  CHX_LAUNCH(k_cell_scan, dim3(1), dim3(1024), st, cell_count, cell_start, ncell, status, status_len, rd.ctrl, rd.ctrl_keep);
  if (!slotted) CHX_LAUNCH(k_cell_scatter, dim3(pb), dim3(256), st, cell_of, n, cell_start, cell_count, sorted_ids);
  const bool lists = cand_list && cand_len && cand_cap > 0;
  // (with the lists: twice the wavefronts, the second half writes them -- k_cell_sort)
  const int64_t sort_waves = (int64_t)ncell * (lists ? 2 : 1);
  CHX_LAUNCH(k_cell_sort, dim3((unsigned)((sort_waves * CHX_WAVE + 255) / 256)), dim3(256), st, cell_start, ncell, sorted_ids, xyz,
             reinterpret_cast<float4*>(xs4_out), g.ncx, g.ncy, cell_count, rd, g, lists ? cand_list : nullptr, cand_len, cand_cap,
             slotted ? rd.cell_slots : nullptr, rd.cell_slot_cap, status);
  return CHX_OK;
}

extern "C" int chx_nl_build_cell(const float* xyz, int32_t n, int32_t m, chx_box_t box,
                                 float cutoff_plus_skin, const int32_t* nc_h, int32_t* cell_of,
                                 int32_t* cell_count, int32_t* cell_start, uint32_t* sorted_ids,
                                 int32_t* status, uint32_t* nl_out, int32_t* nl_mask_out,
                                 int32_t* n_nbr_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 1 && m >= 1, "need n >= 1, m >= 1");
  CHX_REQUIRE(xyz && nc_h && cell_of && cell_count && cell_start && sorted_ids && status &&
                  nl_out && nl_mask_out && n_nbr_out,
              "null pointer");
  hipStream_t st = chx_stream(stream);
  CellGrid g;
  int rc = chx_cell_prepare(xyz, n, box, cutoff_plus_skin, nc_h, cell_of, cell_count, cell_start,
                            sorted_ids, status, 4, &g, st);
  if (rc != CHX_OK) return rc;
  const Box3 b3 = make_box3(box);
  const float T = chx_cutoff_threshold_h(cutoff_plus_skin);
  const unsigned wb = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);
  hipLaunchKernelGGL(k_nl_build_cell, dim3(wb), dim3(256), 0, st, xyz, n, m, b3, T, g, cell_of,
                     cell_start, sorted_ids, nl_out, nl_mask_out, n_nbr_out, status);
  return chx_check_launch("chx_nl_build_cell");
}

// ---------------------------------------------------------------------------
// NeighborListNsqrd.calculate / check
// ---------------------------------------------------------------------------
__global__ void k_nl_calculate(const float* __restrict__ xyz, const uint32_t* __restrict__ nl,
                               const int32_t* __restrict__ nl_mask, int n, int m, Box3 box, float T,
                               int32_t* __restrict__ n_pairs, int32_t* __restrict__ mask_out,
                               float* __restrict__ dist, float* __restrict__ rij) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * m) return;
  const int i = (int)(idx / m);
  uint32_t j = nl[idx];
  if (j >= (uint32_t)n) j = (uint32_t)(n - 1);  // JAX gathers clamp out-of-range indices
  float rx, ry, rz;
  displacement(box, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2],
               xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2], rx, ry, rz);
  const float r2 = norm2(rx, ry, rz);
  const int mk = (r2 <= T && nl_mask[idx] != 0) ? 1 : 0;  // chiron/neighbors.py:782
  mask_out[idx] = mk;
  dist[idx] = fsqrt(r2);
  rij[3 * idx] = rx;
  rij[3 * idx + 1] = ry;
  rij[3 * idx + 2] = rz;
  if (mk) atomicAdd(&n_pairs[i], 1);
}

extern "C" int chx_nl_calculate(const float* xyz, const uint32_t* nl, const int32_t* nl_mask,
                                int32_t n, int32_t m, chx_box_t box, float cutoff,
                                int32_t* n_pairs_out, int32_t* mask_out, float* dist_out,
                                float* rij_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0 && m >= 1, "need n >= 0, m >= 1");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(xyz && nl && nl_mask && n_pairs_out && mask_out && dist_out && rij_out, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  hipStream_t st = chx_stream(stream);
  if (hipMemsetAsync(n_pairs_out, 0, sizeof(int32_t) * (size_t)n, st) != hipSuccess) {
    chx_set_error("chx_nl_calculate: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  const int64_t total = (int64_t)n * m;
  hipLaunchKernelGGL(k_nl_calculate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, xyz, nl,
                     nl_mask, n, m, make_box3(box), chx_cutoff_threshold_h(cutoff), n_pairs_out,
                     mask_out, dist_out, rij_out);
  return chx_check_launch("chx_nl_calculate");
}

__global__ void k_nl_check(const float* __restrict__ xyz, const float* __restrict__ ref, int n,
                           Box3 box, float T, int32_t* __restrict__ flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float rx, ry, rz;
  displacement(box, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2],
               ref[3 * (size_t)i], ref[3 * (size_t)i + 1], ref[3 * (size_t)i + 2], rx, ry, rz);
  if (norm2(rx, ry, rz) > T) flag[0] = 1;  // d >= skin/2  (chiron/neighbors.py:868)
}

extern "C" int chx_nl_check(const float* xyz, const float* ref_xyz, int32_t n, chx_box_t box,
                            float skin, int32_t* flag_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  CHX_REQUIRE(flag_out, "null pointer");
  hipStream_t st = chx_stream(stream);
  if (hipMemsetAsync(flag_out, 0, sizeof(int32_t), st) != hipSuccess) {
    chx_set_error("chx_nl_check: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(xyz && ref_xyz, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  // skin / 2.0 is formed in fp32 inside the jitted function (chiron/neighbors.py:868)
  const float T = chx_cutoff_threshold_h(skin * 0.5f);
  hipLaunchKernelGGL(k_nl_check, dim3((n + 255) / 256), dim3(256), 0, st, xyz, ref_xyz, n,
                     make_box3(box), T, flag_out);
  return chx_check_launch("chx_nl_check");
}

// ---------------------------------------------------------------------------
// PairListNsqrd
// ---------------------------------------------------------------------------
__global__ void k_pl_build(int n, uint32_t* __restrict__ all_pairs, uint8_t* __restrict__ red) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = n - 1;
  if (idx >= (int64_t)n * w) return;
  const int i = (int)(idx / w), k = (int)(idx % w);
  const int j = k < i ? k : k + 1;
  all_pairs[idx] = (uint32_t)j;
  red[idx] = i < j ? 1 : 0;
}

extern "C" int chx_pl_build(int32_t n, uint32_t* all_pairs_out, uint8_t* reduction_mask_out,
                            chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n <= 1) return CHX_OK;
  CHX_REQUIRE(all_pairs_out && reduction_mask_out, "null pointer");
  const int64_t total = (int64_t)n * (n - 1);
  hipLaunchKernelGGL(k_pl_build, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     chx_stream(stream), n, all_pairs_out, reduction_mask_out);
  return chx_check_launch("chx_pl_build");
}

__global__ void k_pl_calculate(const float* __restrict__ xyz, int n, Box3 box, int use_cutoff, float T,
                               int32_t* __restrict__ n_pairs, int32_t* __restrict__ mask_out,
                               float* __restrict__ dist, float* __restrict__ rij) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = n - 1;
  if (idx >= (int64_t)n * w) return;
  const int i = (int)(idx / w), k = (int)(idx % w);
  const int j = k < i ? k : k + 1;
  float rx, ry, rz;
  displacement(box, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2],
               xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2], rx, ry, rz);
  const float r2 = norm2(rx, ry, rz);
  const int mk = (i < j && (!use_cutoff || r2 <= T)) ? 1 : 0;  // chiron/neighbors.py:1156, :1211
  mask_out[idx] = mk;
  dist[idx] = fsqrt(r2);
  rij[3 * idx] = rx;
  rij[3 * idx + 1] = ry;
  rij[3 * idx + 2] = rz;
  if (mk) atomicAdd(&n_pairs[i], 1);
}

extern "C" int chx_pl_calculate(const float* xyz, int32_t n, chx_box_t box, float cutoff,
                                int32_t* n_pairs_out, int32_t* mask_out, float* dist_out,
                                float* rij_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(n_pairs_out, "null pointer");
  hipStream_t st = chx_stream(stream);
  if (hipMemsetAsync(n_pairs_out, 0, sizeof(int32_t) * (size_t)n, st) != hipSuccess) {
    chx_set_error("chx_pl_calculate: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  if (n == 1) return CHX_OK;
  CHX_REQUIRE(xyz && mask_out && dist_out && rij_out, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  const int use_cutoff = cutoff >= 0.0f ? 1 : 0;
  const float T = use_cutoff ? chx_cutoff_threshold_h(cutoff) : 0.0f;
  const int64_t total = (int64_t)n * (n - 1);
  hipLaunchKernelGGL(k_pl_calculate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, xyz, n,
                     make_box3(box), use_cutoff, T, n_pairs_out, mask_out, dist_out, rij_out);
  return chx_check_launch("chx_pl_calculate");
}
