// Feasibility probe (GPU box only, not part of libchx): a cluster-pair (8 x 8) list against the per-particle gather list
// at LIQUID density -- round-2 review item 7; SURVEY.md 8(d)'s secondary runs (rho* = 0.8, chiron/tests/conftest.py:48).
//
// At rho* = 0.8 a particle has ~126 listed partners inside cutoff + 0.12 nm; the product's list kernel reads one
// 4-byte index and gathers one 16-byte position per listed pair (132 MB of index words per step at N = 262144) and is
// bound by the L1's gather rate and VALU issue together.  A cluster-pair list (clusters of 8 spatially sorted
// particles; for every i-cluster the j-clusters whose bounding boxes come within the list radius) reads ONE index and
// 128 contiguous bytes of positions per 64 pair evaluations -- but evaluates every pair of the two clusters, also those
// out of range.  Which effect wins on this part is what the probe measures:
//
//   A   per-particle rows, G lanes per particle, index quad -> four gathers per lane and chunk (the product's shape)
//   B   one wavefront per i-cluster: lane (i, j) evaluates particle i of the cluster against particle j of each listed
//       j-cluster (coalesced 128-byte position loads, no per-pair index), 8-lane reduction at the end
//
// Same pair arithmetic in both (minimum image, cutoff test, reciprocal-based LJ); forces must agree to fp32 summation
// order.  Only the force kernels are timed (no integrate tail: at this size the step is throughput-bound).
//
//   hipcc --offload-arch=gfx950 -O3 -o cluster_pair_probe cluster_pair_probe.hip && ./cluster_pair_probe [n_side] [reps]
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

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float minimg(float d, float L) {
  d = d > 0.5f * L ? d - L : d;
  return d < -0.5f * L ? d + L : d;
}

__device__ __forceinline__ void pair(const f4& xi, const f4& xj, float L, float T, bool live, float& fx, float& fy, float& fz) {
  const float dx = minimg(xi.x - xj.x, L), dy = minimg(xi.y - xj.y, L), dz = minimg(xi.z - xj.z, L);
  const float r2 = (dx * dx + dy * dy) + dz * dz;
  const float inv = __builtin_amdgcn_rcpf(r2 + 1e-12f);
  const float s2 = 0.1156f * inv, s6 = s2 * s2 * s2;
  float fs = (s6 * 23.9f) * (2.0f * s6 - 1.0f) * inv;
  fs = (live && r2 <= T && r2 > 0.f) ? fs : 0.f;
  fx += fs * dx, fy += fs * dy, fz += fs * dz;
}

// ---------------------------------------------------------------- A: per-particle rows, gathers
template <int G>
__global__ __launch_bounds__(256) void k_gather(const f4* __restrict__ x, const uint32_t* __restrict__ fl, const int* __restrict__ deg,
                                                int n, float L, float T, float* __restrict__ f3) {
  constexpr int W = 4 * G;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int p = tid / G, q = tid % G;
  if (p >= n) return;
  const uint4* col = reinterpret_cast<const uint4*>(fl) + (size_t)p * G + q;
  const f4 xi = x[p];
  const int d = deg[p];
  const int nc = (d + W - 1) / W;
  float fx = 0, fy = 0, fz = 0;
  uint4 e = col[0];
  f4 a = x[e.x], b = x[e.y], c = x[e.z], dd = x[e.w];
  for (int ch = 0; ch < nc; ++ch) {
    uint4 nx = e;
    if (ch + 1 < nc) nx = col[(size_t)(ch + 1) * n * G];
    const int k0 = ch * W + q;  // word u of lane q is row entry k0 + u G
    pair(xi, a, L, T, k0 < d, fx, fy, fz);
    pair(xi, b, L, T, k0 + G < d, fx, fy, fz);
    pair(xi, c, L, T, k0 + 2 * G < d, fx, fy, fz);
    pair(xi, dd, L, T, k0 + 3 * G < d, fx, fy, fz);
    if (ch + 1 < nc) {
      e = nx;
      a = x[e.x], b = x[e.y], c = x[e.z], dd = x[e.w];
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    fx += __shfl_xor(fx, o, 64);
    fy += __shfl_xor(fy, o, 64);
    fz += __shfl_xor(fz, o, 64);
  }
  if (q == 0) f3[3 * (size_t)p] = fx, f3[3 * (size_t)p + 1] = fy, f3[3 * (size_t)p + 2] = fz;
}

// ---------------------------------------------------------------- B: cluster pairs, one wavefront per i-cluster
template <int UNROLL>
__global__ __launch_bounds__(256) void k_cluster(const f4* __restrict__ x, const uint32_t* __restrict__ cj_list,
                                                 const int* __restrict__ cj_start, int nclusters, float L, float T,
                                                 float* __restrict__ f3) {
  const int ci = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (ci >= nclusters) return;
  const int lane = threadIdx.x & 63, li = lane >> 3, lj = lane & 7;
  const f4 xi = x[8 * ci + li];
  const int s0 = cj_start[ci], s1 = cj_start[ci + 1];
  float fx = 0, fy = 0, fz = 0;
  int s = s0;
  for (; s + UNROLL <= s1; s += UNROLL) {
    f4 xj[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) xj[u] = x[8 * cj_list[s + u] + lj];  // 8 distinct positions = one 128-byte line
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) pair(xi, xj[u], L, T, true, fx, fy, fz);  // (a particle against itself has r2 = 0)
  }
  for (; s < s1; ++s) pair(xi, x[8 * cj_list[s] + lj], L, T, true, fx, fy, fz);
  fx += __shfl_xor(fx, 1, 64), fy += __shfl_xor(fy, 1, 64), fz += __shfl_xor(fz, 1, 64);
  fx += __shfl_xor(fx, 2, 64), fy += __shfl_xor(fy, 2, 64), fz += __shfl_xor(fz, 2, 64);
  fx += __shfl_xor(fx, 4, 64), fy += __shfl_xor(fy, 4, 64), fz += __shfl_xor(fz, 4, 64);
  if (lj == 0) {
    const size_t p = 8 * (size_t)ci + li;
    f3[3 * p] = fx, f3[3 * p + 1] = fy, f3[3 * p + 2] = fz;
  }
}

static uint32_t morton3(uint32_t a, uint32_t b, uint32_t c) {  // 3 bits per axis
  uint32_t m = 0;
  for (int k = 0; k < 3; ++k) m |= ((a >> k) & 1u) << (3 * k) | ((b >> k) & 1u) << (3 * k + 1) | ((c >> k) & 1u) << (3 * k + 2);
  return m;
}

template <int G>
static float time_gather(const std::vector<std::vector<int>>& rows, int n, float L, float T, const f4* d_x, float* d_f,
                         hipStream_t st, int reps, double* avg_out) {
  constexpr int W = 4 * G;
  int maxdeg = 0;
  double avg = 0;
  for (auto& r : rows) maxdeg = std::max(maxdeg, (int)r.size()), avg += r.size();
  *avg_out = avg / n;
  const int chunks = (maxdeg + W - 1) / W;
  std::vector<uint32_t> fl((size_t)chunks * n * W);
  std::vector<int> deg(n);
  for (int p = 0; p < n; ++p) {
    deg[p] = (int)rows[p].size();
    for (int k = 0; k < chunks * W; ++k) {
      const int e = k % W;
      fl[(((size_t)(k / W)) * n + p) * W + 4 * (e % G) + e / G] = k < deg[p] ? (uint32_t)rows[p][k] : (uint32_t)p;
    }
  }
  uint32_t* d_fl;
  int* d_deg;
  CK(hipMalloc(&d_fl, fl.size() * 4));
  CK(hipMalloc(&d_deg, n * 4));
  CK(hipMemcpy(d_fl, fl.data(), fl.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_deg, deg.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned blocks = (unsigned)(((size_t)n * G + 255) / 256);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k_gather<G>), dim3(blocks), dim3(256), 0, st, d_x, d_fl, d_deg, n, L, T, d_f);
  std::vector<float> us;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gather<G>), dim3(blocks), dim3(256), 0, st, d_x, d_fl, d_deg, n, L, T, d_f);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    us.push_back(1e3f * ms / reps);
  }
  std::sort(us.begin(), us.end());
  printf("A  gather list, G = %d lanes per particle: %8.2f us per launch (index words %.1f MB, %d chunks of %d)\n", G, us[2],
         fl.size() * 4e-6, chunks, W);
  CK(hipFree(d_fl));
  CK(hipFree(d_deg));
  return us[2];
}

int main(int argc, char** argv) {
  const int side = argc > 1 ? atoi(argv[1]) : 64;  // particles per edge: 64 -> N = 262144, 32 -> 32768
  const int reps = argc > 2 ? atoi(argv[2]) : 200;
  const int n = side * side * side;
  const float sigma = 0.34f, rho = 0.8f, rlist = 1.14f, rc = 1.02f;
  const float L = cbrtf(n * sigma * sigma * sigma / rho), T = rc * rc;
  const int nc = (int)floorf(L / rlist);
  std::vector<float> x(3 * n);
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-0.15f, 0.15f);
  const float a = L / side;
  for (int i = 0; i < n; ++i) {
    const int ix = i % side, iy = (i / side) % side, iz = i / (side * side);
    const float p[3] = {(ix + 0.5f + U(rng)) * a, (iy + 0.5f + U(rng)) * a, (iz + 0.5f + U(rng)) * a};
    for (int c = 0; c < 3; ++c) x[3 * i + c] = p[c] - floorf(p[c] / L) * L;
  }
  // spatial order: cells (x fastest), Morton order of the position inside the cell
  std::vector<int> cell(n), order(n);
  std::vector<uint32_t> sub(n);
  for (int i = 0; i < n; ++i) {
    int c[3];
    uint32_t sb[3];
    for (int k = 0; k < 3; ++k) {
      const float u = x[3 * i + k] * nc / L;
      c[k] = std::min(nc - 1, (int)u);
      sb[k] = std::min(7u, (uint32_t)((u - c[k]) * 8.f));
    }
    cell[i] = (c[2] * nc + c[1]) * nc + c[0];
    sub[i] = morton3(sb[0], sb[1], sb[2]);
    order[i] = i;
  }
  std::stable_sort(order.begin(), order.end(), [&](int u, int v) { return cell[u] != cell[v] ? cell[u] < cell[v] : sub[u] < sub[v]; });
  std::vector<f4> xs(n);
  std::vector<int> cs((size_t)nc * nc * nc + 1, 0), cellp(n);
  for (int p = 0; p < n; ++p) {
    const int i = order[p];
    xs[p] = (f4){x[3 * i], x[3 * i + 1], x[3 * i + 2], 0.f};
    cellp[p] = cell[i];
    cs[cell[i] + 1]++;
  }
  for (size_t c = 1; c < cs.size(); ++c) cs[c] += cs[c - 1];
  auto mi = [&](float d) { return d > 0.5f * L ? d - L : (d < -0.5f * L ? d + L : d); };
  // A: full rows within rlist
  std::vector<std::vector<int>> rows(n);
  size_t in_cutoff = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : in_cutoff)
  for (int p = 0; p < n; ++p) {
    const int ci = cellp[p], cx = ci % nc, cy = (ci / nc) % nc, cz = ci / (nc * nc);
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const int o = (((cz + dz + nc) % nc) * nc + (cy + dy + nc) % nc) * nc + (cx + dx + nc) % nc;
      for (int t = cs[o]; t < cs[o + 1]; ++t) {
        if (t == p) continue;
        const float rx = mi(xs[p].x - xs[t].x), ry = mi(xs[p].y - xs[t].y), rz = mi(xs[p].z - xs[t].z);
        const float r2 = rx * rx + ry * ry + rz * rz;
        if (r2 < rlist * rlist) rows[p].push_back(t);
        if (r2 <= T) ++in_cutoff;
      }
    }
    std::sort(rows[p].begin(), rows[p].end());
  }
  // B: clusters of 8 consecutive particles, bounding boxes (relative to the cluster's first particle: periodic-safe)
  if (n % 8) { fprintf(stderr, "n must be a multiple of 8\n"); return 1; }
  const int ncl = n / 8;
  std::vector<float> lo(3 * ncl), hi(3 * ncl), org(3 * ncl);
  for (int c = 0; c < ncl; ++c) {
    const f4 o = xs[8 * c];
    float l[3] = {0, 0, 0}, h[3] = {0, 0, 0};
    for (int k = 1; k < 8; ++k) {
      const float d[3] = {mi(xs[8 * c + k].x - o.x), mi(xs[8 * c + k].y - o.y), mi(xs[8 * c + k].z - o.z)};
      for (int t = 0; t < 3; ++t) l[t] = std::min(l[t], d[t]), h[t] = std::max(h[t], d[t]);
    }
    org[3 * c] = o.x, org[3 * c + 1] = o.y, org[3 * c + 2] = o.z;
    for (int t = 0; t < 3; ++t) lo[3 * c + t] = l[t], hi[3 * c + t] = h[t];
  }
  // candidate j-clusters of an i-cluster: those holding a particle of the 27 cells around any of its particles' cells;
  // kept when the boxes come within rlist (every pair within rlist is then covered: checked below)
  std::vector<std::vector<uint32_t>> cjs(ncl);
#pragma omp parallel for schedule(dynamic, 64)
  for (int c = 0; c < ncl; ++c) {
    std::vector<int> cand;
    int lastcell = -1;
    for (int k = 0; k < 8; ++k) {
      const int ci = cellp[8 * c + k];
      if (ci == lastcell) continue;
      lastcell = ci;
      const int cx = ci % nc, cy = (ci / nc) % nc, cz = ci / (nc * nc);
      for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        const int o = (((cz + dz + nc) % nc) * nc + (cy + dy + nc) % nc) * nc + (cx + dx + nc) % nc;
        for (int t = cs[o] / 8; t <= (cs[o + 1] - 1) / 8 && cs[o + 1] > cs[o]; ++t) cand.push_back(t);
      }
    }
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    for (int j : cand) {
      float d2 = 0;
      for (int t = 0; t < 3; ++t) {
        const float sh = mi(org[3 * j + t] - org[3 * c + t]);  // j's origin relative to i's
        const float gap = std::max(0.f, std::max(sh + lo[3 * j + t] - hi[3 * c + t], lo[3 * c + t] - (sh + hi[3 * j + t])));
        d2 += gap * gap;
      }
      if (d2 < rlist * rlist) cjs[c].push_back((uint32_t)j);
    }
  }
  std::vector<int> cstart(ncl + 1, 0);
  for (int c = 0; c < ncl; ++c) cstart[c + 1] = cstart[c] + (int)cjs[c].size();
  std::vector<uint32_t> cjl(cstart[ncl]);
  for (int c = 0; c < ncl; ++c) std::copy(cjs[c].begin(), cjs[c].end(), cjl.begin() + cstart[c]);
  // coverage check: every listed partner of A lies in a listed j-cluster of B
  size_t missing = 0;
#pragma omp parallel for reduction(+ : missing)
  for (int p = 0; p < n; ++p)
    for (int t : rows[p])
      if (!std::binary_search(cjs[p / 8].begin(), cjs[p / 8].end(), (uint32_t)(t / 8))) ++missing;
  double avg = 0;
  for (auto& r : rows) avg += r.size();
  printf("N = %d, rho* = %.1f, box %.3f nm, %d^3 cells; listed partners per particle (< %.2f nm): %.1f, inside the cutoff: %.1f\n",
         n, rho, L, nc, rlist, avg / n, (double)in_cutoff / n);
  printf("cluster pairs: %.1f j-clusters per i-cluster = %.0f pair evaluations per particle (%.2fx the listed pairs, %.2fx the "
         "interacting ones); %.1f MB of cluster indices; pairs of A missing from B: %zu\n",
         (double)cstart[ncl] / ncl, 8.0 * cstart[ncl] / ncl, 8.0 * cstart[ncl] / ncl / (avg / n),
         8.0 * cstart[ncl] / ncl / ((double)in_cutoff / n), cjl.size() * 4e-6, missing);

  f4* d_x;
  float *d_fa, *d_fb;
  uint32_t* d_cj;
  int* d_cs;
  CK(hipMalloc(&d_x, n * sizeof(f4)));
  CK(hipMalloc(&d_fa, 3 * n * 4));
  CK(hipMalloc(&d_fb, 3 * n * 4));
  CK(hipMalloc(&d_cj, cjl.size() * 4));
  CK(hipMalloc(&d_cs, (ncl + 1) * 4));
  CK(hipMemcpy(d_x, xs.data(), n * sizeof(f4), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cj, cjl.data(), cjl.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cs, cstart.data(), (ncl + 1) * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  double dummy;
  time_gather<2>(rows, n, L, T, d_x, d_fa, st, reps, &dummy);
  time_gather<8>(rows, n, L, T, d_x, d_fa, st, reps, &dummy);
  time_gather<4>(rows, n, L, T, d_x, d_fa, st, reps, &dummy);  // (last: d_fa holds its forces for the comparison)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned cblocks = (unsigned)(((size_t)ncl * 64 + 255) / 256);
  auto time_cluster = [&](int unroll) {
    auto launch = [&]() {
      if (unroll == 1) hipLaunchKernelGGL((k_cluster<1>), dim3(cblocks), dim3(256), 0, st, d_x, d_cj, d_cs, ncl, L, T, d_fb);
      else if (unroll == 2) hipLaunchKernelGGL((k_cluster<2>), dim3(cblocks), dim3(256), 0, st, d_x, d_cj, d_cs, ncl, L, T, d_fb);
      else hipLaunchKernelGGL((k_cluster<4>), dim3(cblocks), dim3(256), 0, st, d_x, d_cj, d_cs, ncl, L, T, d_fb);
    };
    for (int i = 0; i < 100; ++i) launch();
    std::vector<float> us;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      us.push_back(1e3f * ms / reps);
    }
    std::sort(us.begin(), us.end());
    printf("B  cluster pairs 8 x 8, %d j-cluster(s) in flight:  %8.2f us per launch\n", unroll, us[2]);
  };
  time_cluster(1);
  time_cluster(2);
  time_cluster(4);
  std::vector<float> fa(3 * n), fb(3 * n);
  CK(hipMemcpy(fa.data(), d_fa, 3 * n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(fb.data(), d_fb, 3 * n * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (int i = 0; i < 3 * n; ++i) worst = std::max(worst, (double)fabsf(fa[i] - fb[i])), scale = std::max(scale, (double)fabsf(fa[i]));
  printf("forces A (G = 4) against B: largest difference %.3g of a largest component %.3g (%s)\n", worst, scale,
         worst <= 2e-5 * scale ? "agree to fp32 summation order" : "DISAGREE");
  return worst <= 2e-5 * scale && missing == 0 ? 0 : 2;
}
