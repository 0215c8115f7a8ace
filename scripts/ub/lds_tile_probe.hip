// Feasibility probe (GPU box only, not part of libchx): does staging a block's NEIGHBOURHOOD in LDS -- north_star's
// "LDS-staged particle tiles", round-2 review item 2 -- shorten the step launch at config C3?
//
// What the product's step kernel does per lane (csrc/chx_lj.hip: k_lj_step, G = 4 lanes per particle, 64 particles per
// 256-thread block): load an index quad of its row (round trip 1), gather the four neighbour positions it names from
// global memory (round trip 2, dependent on 1), pair arithmetic, reduce, integrate.  The proposal: a block's 64
// consecutive particles (cell order) have their partners in a handful of contiguous runs of the cell-ordered position
// array; load those runs into LDS with coalesced loads, let the list hold 16-bit tile slots, read partners with
// ds_read_b128 -- the gathers, and with them round trip 2, are gone.
//
// The catch this probe measures: the runs of a block are not a function of its block id alone (cells hold varying
// numbers of particles), so the block must first fetch its RUN TABLE -- a dependent round trip again -- and at
// rho* = 0.1 the tile is mostly particles nobody in the block interacts with (bytes from L2 that the gathers, 88 % of
// which hit L1, never asked for).
//
//   A        index quad -> global gathers (the product's shape), one launch per step
//   B        run table (scalar loads) -> coalesced tile loads into LDS -> ds_read_b128 partners, one launch per step
//   B-const  as B with the SAME run table for every block taken from kernel arguments (no table round trip;
//            physically meaningless -- every block stages block 0's runs -- a lower bound for B's latency chain)
//
// Geometry = config C3: N = 32768 particles, rho* = 0.1 (box edge 23.4402 nm), cell order on the 15^3 grid the product
// uses, pruned rows (partners within cutoff + 0.12 nm = 1.14 nm, ~16 per particle, both directions), rows in chunks of
// 16 entries interleaved over the 4 lanes exactly as include/chx.h lays them out.  A and B must end in identical bits.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lds_tile_probe lds_tile_probe.hip && ./lds_tile_probe [steps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int G = 4;           // lanes per particle
constexpr int PPB = 256 / G;   // 64 particles per block
constexpr int W = 4 * G;       // row entries per chunk
constexpr int MAXRUNS = 32;
constexpr int TILE_CAP = 1280;  // float4 slots of LDS per block (20 KB)

typedef float f4 __attribute__((ext_vector_type(4)));

struct RunTable {
  int nruns;
  int start[MAXRUNS];    // first particle (cell order) of run r
  int pre[MAXRUNS + 1];  // tile slot of run r's first particle; pre[nruns] = tile length
};

__device__ __forceinline__ float minimg(float d, float L) {
  d = d > 0.5f * L ? d - L : d;
  return d < -0.5f * L ? d + L : d;
}

// the product's pair arithmetic in spirit: minimum image, cutoff test, reciprocal-based LJ
__device__ __forceinline__ void pair(const f4& xi, const f4& xj, float L, float T, float& fx, float& fy, float& fz) {
  const float dx = minimg(xi.x - xj.x, L), dy = minimg(xi.y - xj.y, L), dz = minimg(xi.z - xj.z, L);
  const float r2 = (dx * dx + dy * dy) + dz * dz;
  const float inv = __builtin_amdgcn_rcpf(r2 + 1e-12f);
  const float s2 = 0.1156f * inv, s6 = s2 * s2 * s2;
  float fs = (s6 * 23.9f) * (2.0f * s6 - 1.0f) * inv;
  fs = (r2 <= T && r2 > 0.f) ? fs : 0.f;
  fx += fs * dx, fy += fs * dy, fz += fs * dz;
}

// stand-in for the product's Philox draw + Box-Muller in the block's first wavefront (~500 dependent-ish instructions)
__device__ __forceinline__ float burn(uint32_t a, uint32_t b) {
  uint32_t c0 = a, c1 = b, c2 = a ^ 0x9E3779B9u, c3 = b + 0xBB67AE85u, k0 = 0x1234u, k1 = 0x5678u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    c0 = h1 ^ c1 ^ k0, c1 = l1, c2 = h0 ^ c3 ^ k1, c3 = l0;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  const float u = ((float)(c0 >> 9) + 0.5f) * (1.0f / 8388608.0f), v = ((float)(c1 >> 9) + 0.5f) * (1.0f / 8388608.0f);
  return __builtin_sqrtf(-2.0f * __logf(u)) * __cosf(6.2831853f * v) + 0.f * (float)(c2 ^ c3);
}

__device__ __forceinline__ float wrapc(float x, float L) {
  x = x >= L ? x - L : x;
  return x < 0.f ? x + L : x;
}

__device__ __forceinline__ void finish(float fx, float fy, float fz, float* sF, int p0, int n, const f4& own, float z, float L,
                                       f4* __restrict__ xn) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    fx += __shfl_xor(fx, o, 64);
    fy += __shfl_xor(fy, o, 64);
    fz += __shfl_xor(fz, o, 64);
  }
  if ((threadIdx.x % G) == 0) {
    const int pl = threadIdx.x / G;
    sF[3 * pl] = fx, sF[3 * pl + 1] = fy, sF[3 * pl + 2] = fz;
  }
  __syncthreads();
  if (threadIdx.x < PPB && p0 + (int)threadIdx.x < n) {
    // a tiny, deterministic move: the configuration stays what the lists were built for over thousands of steps
    f4 o;
    o.x = wrapc(own.x + 1e-9f * sF[3 * threadIdx.x] + 1e-9f * z, L);
    o.y = wrapc(own.y + 1e-9f * sF[3 * threadIdx.x + 1], L);
    o.z = wrapc(own.z + 1e-9f * sF[3 * threadIdx.x + 2], L);
    o.w = 0.f;
    xn[p0 + threadIdx.x] = o;
  }
}

// ---------------------------------------------------------------- A: index quad -> gathers
__global__ __launch_bounds__(256) void k_gather(const f4* __restrict__ xc, f4* __restrict__ xn, const uint32_t* __restrict__ fl,
                                                const int* __restrict__ deg, int n, float L, float T, int step) {
  __shared__ float sF[3 * PPB];
  const int p0 = blockIdx.x * PPB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  const uint4* col = reinterpret_cast<const uint4*>(fl) + (size_t)p * G + q;
  uint4 e = col[0];
  const f4 xi = xc[p];
  const int d = deg[p];
  f4 own = xi;
  if (threadIdx.x < PPB) own = xc[p0 + threadIdx.x];
  f4 a = xc[e.x], b = xc[e.y], c = xc[e.z], dd = xc[e.w];
  float z = 0.f;
  if (threadIdx.x < PPB) z = burn((uint32_t)(p0 + threadIdx.x), (uint32_t)step);  // under the gathers' latency
  const int nc = (d + W - 1) / W;
  float fx = 0, fy = 0, fz = 0;
  for (int ch = 0; ch < nc; ++ch) {
    uint4 nx = e;
    if (ch + 1 < nc) nx = col[(size_t)(ch + 1) * n * G];
    pair(xi, a, L, T, fx, fy, fz);
    pair(xi, b, L, T, fx, fy, fz);
    pair(xi, c, L, T, fx, fy, fz);
    pair(xi, dd, L, T, fx, fy, fz);
    if (ch + 1 < nc) {
      e = nx;
      a = xc[e.x], b = xc[e.y], c = xc[e.z], dd = xc[e.w];
    }
  }
  finish(fx, fy, fz, sF, p0, n, own, z, L, xn);
}

// ---------------------------------------------------------------- B: run table -> LDS tile -> ds_read partners
template <bool CONST_TABLE>
__global__ __launch_bounds__(256) void k_tile(const f4* __restrict__ xc, f4* __restrict__ xn, const RunTable* __restrict__ tables,
                                              RunTable ktable, const uint16_t* __restrict__ slots, const int* __restrict__ deg,
                                              int n, float L, float T, int step) {
  __shared__ f4 tile[TILE_CAP];
  __shared__ float sF[3 * PPB];
  const int p0 = blockIdx.x * PPB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  // loads that need no table: own row's slots, length, position
  const uint2* col = reinterpret_cast<const uint2*>(slots) + (size_t)p * G + q;
  uint2 s = col[0];
  const f4 xi = xc[p];
  const int d = deg[p];
  f4 own = xi;
  if (threadIdx.x < PPB) own = xc[p0 + threadIdx.x];
  // the block's runs: every wavefront fetches the table with ONE vector load per column (lane r holds run r: no serial
  // scalar loads inside the loop), then wave w stages runs w, w + 4, ... with coalesced loads
  const RunTable& tb = CONST_TABLE ? ktable : tables[blockIdx.x];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nr = tb.nruns;
  const int lr = lane < MAXRUNS ? lane : MAXRUNS - 1;
  const int my_start = tb.start[lr], my_pre = tb.pre[lr], my_end = tb.pre[lr + 1];
  for (int r = wv; r < nr; r += 4) {
    const int rs = __builtin_amdgcn_readlane(my_start, r), rp = __builtin_amdgcn_readlane(my_pre, r);
    const int len = __builtin_amdgcn_readlane(my_end, r) - rp;
    for (int k = lane; k < len; k += 64) tile[rp + k] = xc[rs + k];
  }
  float z = 0.f;
  if (threadIdx.x < PPB) z = burn((uint32_t)(p0 + threadIdx.x), (uint32_t)step);  // under the tile loads' latency
  __syncthreads();
  const int nc = (d + W - 1) / W;
  float fx = 0, fy = 0, fz = 0;
  for (int ch = 0; ch < nc; ++ch) {
    uint2 nx = s;
    if (ch + 1 < nc) nx = col[(size_t)(ch + 1) * n * G];
    const f4 a = tile[s.x & 0xFFFFu], b = tile[s.x >> 16], c = tile[s.y & 0xFFFFu], dd = tile[s.y >> 16];
    pair(xi, a, L, T, fx, fy, fz);
    pair(xi, b, L, T, fx, fy, fz);
    pair(xi, c, L, T, fx, fy, fz);
    pair(xi, dd, L, T, fx, fy, fz);
    s = nx;
  }
  finish(fx, fy, fz, sF, p0, n, own, z, L, xn);
}

// ---------------------------------------------------------------- host
int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 2000;
  const int n = 32768, nc = 15;
  const float L = 23.4402f, rin = 1.14f, T = 1.02f * 1.02f;
  // jittered lattice + thermal scatter, then cell order (x fastest, as the product bins)
  std::vector<float> x(3 * n);
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-0.3f, 0.3f);
  const float a = L / 32.f;
  for (int i = 0; i < n; ++i) {
    const int ix = i % 32, iy = (i / 32) % 32, iz = i / 1024;
    const float p[3] = {(ix + 0.5f + U(rng)) * a, (iy + 0.5f + U(rng)) * a, (iz + 0.5f + U(rng)) * a};
    for (int c = 0; c < 3; ++c) x[3 * i + c] = p[c] - floorf(p[c] / L) * L;
  }
  std::vector<int> cell(n), order(n);
  for (int i = 0; i < n; ++i) {
    int c[3];
    for (int k = 0; k < 3; ++k) c[k] = std::min(nc - 1, (int)(x[3 * i + k] * nc / L));
    cell[i] = (c[2] * nc + c[1]) * nc + c[0];
    order[i] = i;
  }
  std::stable_sort(order.begin(), order.end(), [&](int u, int v) { return cell[u] < cell[v]; });
  std::vector<f4> xs(n);
  std::vector<int> cs(nc * nc * nc + 1, 0);
  for (int p = 0; p < n; ++p) {
    const int i = order[p];
    xs[p] = (f4){x[3 * i], x[3 * i + 1], x[3 * i + 2], 0.f};
    cs[cell[i] + 1]++;
  }
  for (size_t c = 1; c < cs.size(); ++c) cs[c] += cs[c - 1];
  // pruned full rows in cell order
  std::vector<std::vector<int>> rows(n);
  auto mi = [&](float d) { return d > 0.5f * L ? d - L : (d < -0.5f * L ? d + L : d); };
  for (int p = 0; p < n; ++p) {
    const int ci = cell[order[p]], cx = ci % nc, cy = (ci / nc) % nc, cz = ci / (nc * nc);
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const int o = (((cz + dz + nc) % nc) * nc + (cy + dy + nc) % nc) * nc + (cx + dx + nc) % nc;
      for (int t = cs[o]; t < cs[o + 1]; ++t) {
        if (t == p) continue;
        const float rx = mi(xs[p].x - xs[t].x), ry = mi(xs[p].y - xs[t].y), rz = mi(xs[p].z - xs[t].z);
        if (rx * rx + ry * ry + rz * rz < rin * rin) rows[p].push_back(t);
      }
    }
    std::sort(rows[p].begin(), rows[p].end());
  }
  int maxdeg = 0;
  double avg = 0;
  for (auto& r : rows) maxdeg = std::max(maxdeg, (int)r.size()), avg += r.size();
  const int chunks = (maxdeg + W - 1) / W;
  const int nblk = n / PPB;
  // A: global-index list, chunk-interleaved (include/chx.h "Force list"); pad = own index
  std::vector<uint32_t> fl((size_t)chunks * n * W);
  std::vector<int> deg(n);
  for (int p = 0; p < n; ++p) {
    deg[p] = (int)rows[p].size();
    for (int k = 0; k < chunks * W; ++k) {
      const int e = k % W;
      fl[(((size_t)(k / W)) * n + p) * W + 4 * (e % G) + e / G] = k < deg[p] ? (uint32_t)rows[p][k] : (uint32_t)p;
    }
  }
  // B: per block the union of partners (and own particles) as contiguous runs; rows as 16-bit tile slots
  std::vector<RunTable> tabs(nblk);
  std::vector<uint16_t> sl((size_t)chunks * n * W);
  size_t tile_total = 0;
  int tile_max = 0, runs_max = 0;
  double runs_avg = 0;
  for (int b = 0; b < nblk; ++b) {
    std::vector<int> ids;
    for (int p = b * PPB; p < (b + 1) * PPB; ++p) {
      ids.push_back(p);
      ids.insert(ids.end(), rows[p].begin(), rows[p].end());
    }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    RunTable& t = tabs[b];
    t.nruns = 0;
    int pos = 0;
    for (size_t k = 0; k < ids.size();) {
      size_t e = k;
      while (e + 1 < ids.size() && ids[e + 1] - ids[e] <= 12) ++e;  // bridge small gaps: fewer, longer runs
      if (t.nruns >= MAXRUNS) { fprintf(stderr, "too many runs\n"); return 1; }
      t.start[t.nruns] = ids[k];
      t.pre[t.nruns] = pos;
      pos += ids[e] - ids[k] + 1;
      ++t.nruns;
      k = e + 1;
    }
    t.pre[t.nruns] = pos;
    if (pos > TILE_CAP) { fprintf(stderr, "tile of %d slots exceeds the LDS cap\n", pos); return 1; }
    tile_total += pos, tile_max = std::max(tile_max, pos), runs_max = std::max(runs_max, t.nruns), runs_avg += t.nruns;
    auto slot_of = [&](int j) {
      for (int r = t.nruns - 1; r >= 0; --r)
        if (j >= t.start[r]) return t.pre[r] + (j - t.start[r]);
      return 0;
    };
    for (int p = b * PPB; p < (b + 1) * PPB; ++p)
      for (int k = 0; k < chunks * W; ++k) {
        const int e = k % W;
        sl[(((size_t)(k / W)) * n + p) * W + 4 * (e % G) + e / G] = (uint16_t)slot_of(k < deg[p] ? rows[p][k] : p);
      }
  }
  printf("N = %d, %d blocks of %d particles; rows: %.1f entries on average, %d at most (%d chunks of %d)\n", n, nblk, PPB,
         avg / n, maxdeg, chunks, W);
  printf("tiles: %.0f positions per block on average (%d at most) in %.1f runs (%d at most) for %.0f listed partners per block: "
         "%.2f MB of tile loads per launch against %.2f MB of lane gathers\n",
         (double)tile_total / nblk, tile_max, runs_avg / nblk, runs_max, avg / n * PPB, tile_total * 16e-6, avg * 16e-6);

  f4 *d_x[2], *d_y[2];
  uint32_t* d_fl;
  uint16_t* d_sl;
  int* d_deg;
  RunTable* d_tab;
  for (int k = 0; k < 2; ++k) {
    CK(hipMalloc(&d_x[k], n * sizeof(f4)));
    CK(hipMalloc(&d_y[k], n * sizeof(f4)));
  }
  CK(hipMalloc(&d_fl, fl.size() * 4));
  CK(hipMalloc(&d_sl, sl.size() * 2));
  CK(hipMalloc(&d_deg, n * 4));
  CK(hipMalloc(&d_tab, nblk * sizeof(RunTable)));
  CK(hipMemcpy(d_fl, fl.data(), fl.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sl, sl.data(), sl.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_deg, deg.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tab, tabs.data(), nblk * sizeof(RunTable), hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  auto run = [&](int variant, f4** buf, int count) {
    for (int s = 0; s < count; ++s) {
      const f4* xc = buf[s & 1];
      f4* xn = buf[(s + 1) & 1];
      if (variant == 0) hipLaunchKernelGGL(k_gather, dim3(nblk), dim3(256), 0, st, xc, xn, d_fl, d_deg, n, L, T, s);
      else if (variant == 1) hipLaunchKernelGGL((k_tile<false>), dim3(nblk), dim3(256), 0, st, xc, xn, d_tab, tabs[0], d_sl, d_deg, n, L, T, s);
      else hipLaunchKernelGGL((k_tile<true>), dim3(nblk), dim3(256), 0, st, xc, xn, d_tab, tabs[0], d_sl, d_deg, n, L, T, s);
    }
  };
  auto timed = [&](int variant, f4** buf) {
    // clock warm-up, then seven samples; the median
    CK(hipMemcpy(buf[0], xs.data(), n * sizeof(f4), hipMemcpyHostToDevice));
    run(variant, buf, 20000);
    CK(hipStreamSynchronize(st));
    std::vector<float> us;
    for (int rep = 0; rep < 7; ++rep) {
      CK(hipMemcpy(buf[0], xs.data(), n * sizeof(f4), hipMemcpyHostToDevice));
      run(variant, buf, 64);
      CK(hipEventRecord(e0, st));
      run(variant, buf, steps);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      us.push_back(1e3f * ms / steps);
    }
    std::sort(us.begin(), us.end());
    return std::make_pair(us[3], us);
  };
  // bit comparison over 64 steps from the same start
  std::vector<f4> ra(n), rb(n);
  CK(hipMemcpy(d_x[0], xs.data(), n * sizeof(f4), hipMemcpyHostToDevice));
  run(0, d_x, 64);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ra.data(), d_x[0], n * sizeof(f4), hipMemcpyDeviceToHost));
  CK(hipMemcpy(d_y[0], xs.data(), n * sizeof(f4), hipMemcpyHostToDevice));
  run(1, d_y, 64);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(rb.data(), d_y[0], n * sizeof(f4), hipMemcpyDeviceToHost));
  const bool same = memcmp(ra.data(), rb.data(), n * sizeof(f4)) == 0;
  double moved = 0;
  for (int p = 0; p < n; ++p) moved = std::max(moved, (double)fabsf(ra[p].x - xs[p].x));
  printf("A and B after 64 steps: %s (largest |dx| from the start %.3g nm)\n", same ? "bit-identical" : "DIFFERENT", moved);
  const char* names[3] = {"A  index quad -> global gathers", "B  run table -> LDS tile -> ds_read", "B-const  table from kernel arguments"};
  for (int round = 0; round < 2; ++round)
    for (int v = 0; v < 3; ++v) {
      auto r = timed(v, v == 0 ? d_x : d_y);
      printf("%-40s %6.3f us per launch (median of 7:", names[v], r.first);
      for (float u : r.second) printf(" %.3f", u);
      printf(")\n");
    }
  return same ? 0 : 2;
}
