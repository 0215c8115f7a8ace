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
#include "chx_common.h"
#include "chx_internal.h"

#define CHX_CELL_CAP 1024  // half-row candidates kept on chip per row


__global__ __launch_bounds__(256) void k_nl_build_cell_fl(
    const float4* __restrict__ xs4, int n, int m, Box3 box, float T, CellGrid g,
    const int32_t* __restrict__ cell_of, const int32_t* __restrict__ cell_start,
    const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ nl, uint32_t* __restrict__ first_out,
    int32_t* __restrict__ n_nbr, int fg, int fcap, uint32_t* __restrict__ fl, int32_t* __restrict__ deg,
    int32_t* __restrict__ status) {
  __shared__ uint32_t cand_all[4][CHX_CELL_CAP];
  const int lane = lane_id();
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= n) return;
  uint32_t* cand = cand_all[threadIdx.x >> 6];
  const uint32_t i = sorted_ids[w];
  const float4 xw = xs4[w];  // positions in cell order: candidate loads below are coalesced
  const float xi = xw.x, yi = xw.y, zi = xw.z;
  const int c = __float_as_int(xw.w);
  const int cx = c % g.ncx, cy = (c / g.ncx) % g.ncy, cz = c / (g.ncx * g.ncy);
  // The 27 surrounding cells are at most 18 contiguous index ranges of the cell-ordered arrays
  // (the three x-neighbours of a (y,z) row are adjacent except across the periodic seam).  Lane r
  // fetches the bounds of range r -- all 18 in one round trip -- and a wave-wide prefix sum
  // flattens them, so the candidate loop below runs with every lane busy.
  int rs = 0, re = 0;
  if (lane < 18) {
    const int combo = lane >> 1, sub = lane & 1;
    const int zz = (cz + combo / 3 - 1 + g.ncz) % g.ncz;
    const int yy = (cy + combo % 3 - 1 + g.ncy) % g.ncy;
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
  int* rng_s = reinterpret_cast<int*>(cand) + CHX_CELL_CAP - 40;  // tail of this wave's LDS slice
  int* rng_o = rng_s + 20;
  if (lane < 18) {
    rng_s[lane] = rs;
    rng_o[lane] = incl - len;
  }
  __threadfence_block();
  int count_h = 0, count_f = 0;
  int r = 0;  // range holding this lane's candidate; candidates only move forward
  // candidate of the NEXT iteration is requested before the current one is processed, so the
  // loop's global loads overlap instead of forming a chain of dependent round trips
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
    bool ok = false, hi = false;
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
    if (t < total) {
      hi = j > i;
      float rx, ry, rz;  // x_i - x_j for row i of the half list, x_j - x_i for row j
      bool bad;
      displacement_oriented_fast(box, hi, xi, yi, zi, xk.x, xk.y, xk.z, rx, ry, rz, bad);
      if (__builtin_expect(bad, 0)) displacement_oriented(box, hi, xi, yi, zi, xk.x, xk.y, xk.z, rx, ry, rz);
      ok = (j != i) & (norm2(rx, ry, rz) <= T);
#ifdef CHX_EXP_BUILD_NOTEST
      ok = ok && xk.x == 12345.f;  // (timing decomposition: no hits, nothing stored)
#endif
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long b = __ballot(ok);
    if (b) {
      const int slot = count_f + __popcll(b & lt);
#ifndef CHX_EXP_BUILD_NOFL
      if (ok && slot < fcap) fl[chx_fl_addr(w, slot, n, fg)] = (uint32_t)k | (hi ? CHX_FL_HI : 0u);
#endif
      count_f += __popcll(b);
    }
    const bool okh = ok && hi;
    const unsigned long long bh = __ballot(okh);
    if (bh) {
      const int slot = count_h + __popcll(bh & lt);
      if (okh && slot < CHX_CELL_CAP - 40) cand[slot] = j;
      count_h += __popcll(bh);
    }
  }
  // force-list row: the force kernel gathers whole chunks (and chunk 0 speculatively), so the
  // unused slots of the chunks it can touch must hold a valid index; later chunks are never read
  {
    const int wd = CHX_FL_QUAD * fg;
    const int used = count_f < fcap ? count_f : fcap;
    int end = (used + wd - 1) / wd * wd;
    end = end < wd ? wd : end;
    end = end < fcap ? end : fcap;
    // (the row's own index: a zero-length pair keeps the force kernel on its fast path)
    for (int sl = used + lane; sl < end; sl += CHX_WAVE) fl[chx_fl_addr(w, sl, n, fg)] = (uint32_t)w;
  }
  const int kept = count_h < CHX_CELL_CAP - 40 ? count_h : CHX_CELL_CAP - 40;
  __threadfence_block();
  // half row: rank sort -> ascending ids, as jnp.where(mask, size=M) returns them
  uint32_t* nl_row = nl + (size_t)i * m;
  uint32_t mn = 0xFFFFFFFFu;
  for (int t = lane; t < kept; t += CHX_WAVE) {
    const uint32_t v = cand[t];
    mn = v < mn ? v : mn;
    int rank = 0;
#ifdef CHX_EXP_BUILD_NOSORT
    rank = t;
#else
    for (int u = 0; u < kept; ++u) rank += cand[u] < v ? 1 : 0;
#endif
#ifndef CHX_EXP_BUILD_NONL
    if (rank < m) nl_row[rank] = v;
#endif
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = __shfl_xor(mn, o, CHX_WAVE);
    mn = other < mn ? other : mn;
  }
  if (lane == 0) {
    n_nbr[i] = count_h;
    first_out[i] = mn;
    deg[w] = count_f < fcap ? count_f : fcap;
    // only the rare failure cases touch the shared status words here (one hot address serialises
    // atomics chip-wide); the maxima the host needs are reduced by k_nl_summary
    if (count_h > CHX_CELL_CAP - 40) atomicOr(&status[0], 1);
    if (count_f > fcap) {
      atomicAdd(&status[4], 1);
      atomicMax(&status[3], count_f);
    }
  }
}

// positions in cell order; .w carries the particle's cell index (as int bits) so the build kernel
// needs no dependent cell_of[sorted_ids[w]] round trip
__global__ void k_gather_sorted(const float* __restrict__ xyz, const uint32_t* __restrict__ sorted_ids,
                                const int32_t* __restrict__ cell_of, int n, float4* __restrict__ xs4) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const size_t j = sorted_ids[k];
  xs4[k] = make_float4(xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2], __int_as_float(cell_of[j]));
}

// status[1] = max n_nbr, status[2] = #rows with n_nbr == m, status[3] = max full degree
__global__ __launch_bounds__(256) void k_nl_summary(const int32_t* __restrict__ n_nbr,
                                                    const int32_t* __restrict__ deg, int n, int m,
                                                    int32_t* __restrict__ status) {
  __shared__ int32_t red[3][4];
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
  }
}

extern "C" int chx_nl_build_cell_fl(const float* xyz, int32_t n, int32_t m, chx_box_t box,
                                    float cutoff_plus_skin, const int32_t* nc_h, int32_t* cell_of,
                                    int32_t* cell_count, int32_t* cell_start, uint32_t* sorted_ids,
                                    int32_t* status8, uint32_t* nl_rows_out, uint32_t* first_out,
                                    int32_t* n_nbr_out, int32_t g, int32_t chunks, uint32_t* fl_out,
                                    int32_t* deg_out, float* xs4_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 1 && m >= 1, "need n >= 1, m >= 1");
  CHX_REQUIRE(xyz && nc_h && cell_of && cell_count && cell_start && sorted_ids && status8 &&
                  nl_rows_out && first_out && n_nbr_out && fl_out && deg_out && xs4_out,
              "null pointer");
  CHX_REQUIRE(g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64,
              "g must be a power of two <= 64");
  CHX_REQUIRE(chunks >= 1 && (int64_t)chunks * CHX_FL_QUAD * g < (1 << 20), "bad force-list capacity");
  hipStream_t st = chx_stream(stream);
  CellGrid grid;
  int rc = chx_cell_prepare(xyz, n, box, cutoff_plus_skin, nc_h, cell_of, cell_count, cell_start,
                            sorted_ids, status8, 8, &grid, st);
  if (rc != CHX_OK) return rc;
  const float T = chx_cutoff_threshold_h(cutoff_plus_skin);
  const unsigned wb = (unsigned)(((int64_t)n * CHX_WAVE + 255) / 256);
  float4* xs4 = reinterpret_cast<float4*>(xs4_out);
  hipLaunchKernelGGL(k_gather_sorted, dim3((n + 255) / 256), dim3(256), 0, st, xyz, sorted_ids, cell_of, n, xs4);
  hipLaunchKernelGGL(k_nl_build_cell_fl, dim3(wb), dim3(256), 0, st, xs4, n, m, make_box3(box), T, grid,
                     cell_of, cell_start, sorted_ids, nl_rows_out, first_out, n_nbr_out, g, chunks * CHX_FL_QUAD * g,
                     fl_out, deg_out, status8);
  const unsigned sb = (unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
  hipLaunchKernelGGL(k_nl_summary, dim3(sb), dim3(256), 0, st, n_nbr_out, deg_out, n, m, status8);
  return chx_check_launch("chx_nl_build_cell_fl");
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
                                float4* __restrict__ ref4, int32_t* __restrict__ ctrl,
                                const int32_t* __restrict__ pub_src, int32_t* __restrict__ pub_host) {
  // riders of the integrator's rebuild (one launch instead of three): the control-block reset of chx_ctrl_reset
  // and the copy of the build's status block to host-mapped memory of chx_publish
  if (blockIdx.x == 0) {
    const int t = threadIdx.x;
    if (ctrl != nullptr && t < 8) ctrl[t] = t == 3 ? 0x7FFFFFFF : ((t == 4 || t == 5) ? -2 : 0);
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
                                 const int32_t* publish_src8, int32_t* publish_host_dst8, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  CHX_REQUIRE((publish_src8 == nullptr) == (publish_host_dst8 == nullptr), "publish needs both a source and a destination");
  if (n == 0 && !ctrl_reset && !publish_src8) return CHX_OK;
  CHX_REQUIRE(n == 0 || (x3 && v3 && mass && x4_out && v_out && mass_out), "null pointer");
  CHX_REQUIRE(!ref4_out || ref3, "ref4_out without ref3");
  const unsigned blocks = n > 0 ? (unsigned)((n + 255) / 256) : 1u;
  hipLaunchKernelGGL(k_pack_langevin, dim3(blocks), dim3(256), 0, chx_stream(stream), x3, v3, mass,
                     ref3, perm, n, reinterpret_cast<float4*>(x4_out), v_out, mass_out,
                     reinterpret_cast<float4*>(ref4_out), ctrl_reset, publish_src8, publish_host_dst8);
  return chx_check_launch("chx_pack_langevin");
}
