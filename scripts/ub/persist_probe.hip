// Feasibility probe (GPU box only, not part of libchx): what does ONE MD-like step cost when the
// per-step kernel boundary is replaced by in-launch hand-offs of the positions?
//
//   A  baseline        one launch per step, plain loads/stores (what k_lj_step does today)
//   B  data-tagged     ONE launch for S steps; every position is a 16-B {x, y, z, tag = step} record
//                      written by one `global_store_dwordx4 sc1`; a consumer gathers its neighbors'
//                      records with `global_load_dwordx4 sc1` and re-polls the ones whose tag is old.
//                      No barrier, no flags: a lane waits only for the particles it actually reads.
//   C  block flags     ONE launch; positions sc1-stored, then (vmcnt(0), __syncthreads) one lane stores
//                      done[block] = step+1 (sc1); a consumer polls the done[] words of the blocks its
//                      rows reference (one wave-wide sc1 load), barrier, then gathers with sc1 loads.
//
// Geometry mimics config C3: N = 32768 particles on a 32^3 lattice in brick order (32 per block,
// G = 8 lanes per particle, 1024 blocks of 256 threads), neighbors = lattice points within
// r2 <= R2 (18 for R2 = 2, 32 for R2 = 4), rows padded to chunks of 32 entries with the row's own index.
// All three variants must end in bit-identical positions.
//
//   hipcc --offload-arch=gfx950 -O3 -o persist_probe persist_probe.hip && ./persist_probe [steps] [R2]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int G = 8;
constexpr int PPB = 256 / G;  // 32 particles per block
constexpr int W = 4 * G;      // entries per chunk
constexpr int MAXDEP = 64;    // blocks a block may depend on (variant C)
constexpr int SPIN_LIMIT = 200000;

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 load_sc1(const f4* p) {
  f4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ void load4_sc1(const f4* p0, const f4* p1, const f4* p2, const f4* p3, f4& a, f4& b, f4& c,
                                          f4& d) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %5, off sc1\n\t"
      "global_load_dwordx4 %2, %6, off sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
__device__ __forceinline__ void store_sc1(f4* p, f4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ int load_i32_sc1(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_i32_sc1(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float minimg(float d, float L) {
  d = d > 0.5f * L ? d - L : d;
  d = d < -0.5f * L ? d + L : d;
  return d;
}

__device__ __forceinline__ void pair(const f4& xi, const f4& xj, float L, float& fx, float& fy, float& fz) {
  const float dx = minimg(xj.x - xi.x, L), dy = minimg(xj.y - xi.y, L), dz = minimg(xj.z - xi.z, L);
  const float r2 = dx * dx + dy * dy + dz * dz;
  const float w = r2 < 6.0f ? 1e-3f * __builtin_amdgcn_rcpf(1.0f + r2) * (1.5f - r2) : 0.f;
  fx += w * dx, fy += w * dy, fz += w * dz;
}

__device__ __forceinline__ float wrapc(float x, float L) {
  x = x >= L ? x - L : x;
  return x < 0.f ? x + L : x;
}

// reduce over the G lanes, stage in LDS, first PPB threads integrate; returns the new record of thread's particle
__device__ __forceinline__ bool finish(float fx, float fy, float fz, float* sF, int p0, int n, const f4& own, float L,
                                       int tag, f4& out) {
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
    out.x = wrapc(own.x + sF[3 * threadIdx.x], L);
    out.y = wrapc(own.y + sF[3 * threadIdx.x + 1], L);
    out.z = wrapc(own.z + sF[3 * threadIdx.x + 2], L);
    out.w = __int_as_float(tag);
    return true;
  }
  return false;
}

// ---------------------------------------------------------------- A: one launch per step
template <int NT>
__global__ __launch_bounds__(NT) void k_step_nt(const f4* __restrict__ xc, f4* __restrict__ xn, const uint32_t* __restrict__ nb,
                                                const int* __restrict__ deg, int n, float L, int step) {
  constexpr int PB = NT / G;
  __shared__ float sF[3 * PB];
  const int p0 = blockIdx.x * PB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  const f4 xi = xc[p];
  const int d = deg[p];
  const int nc = (d + W - 1) / W;
  f4 own = threadIdx.x < PB ? xc[p0 + threadIdx.x] : xi;
  float fx = 0, fy = 0, fz = 0;
  for (int c = 0; c < nc; ++c) {
    const uint4 e = reinterpret_cast<const uint4*>(nb)[((size_t)c * n + p) * G + q];
    const f4 a = xc[e.x], b = xc[e.y], cc = xc[e.z], dd = xc[e.w];
    pair(xi, a, L, fx, fy, fz);
    pair(xi, b, L, fx, fy, fz);
    pair(xi, cc, L, fx, fy, fz);
    pair(xi, dd, L, fx, fy, fz);
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    fx += __shfl_xor(fx, o, 64);
    fy += __shfl_xor(fy, o, 64);
    fz += __shfl_xor(fz, o, 64);
  }
  if (q == 0) {
    const int pl = threadIdx.x / G;
    sF[3 * pl] = fx, sF[3 * pl + 1] = fy, sF[3 * pl + 2] = fz;
  }
  __syncthreads();
  if (threadIdx.x < PB) {
    f4 out;
    out.x = wrapc(own.x + sF[3 * threadIdx.x], L);
    out.y = wrapc(own.y + sF[3 * threadIdx.x + 1], L);
    out.z = wrapc(own.z + sF[3 * threadIdx.x + 2], L);
    out.w = __int_as_float(step + 1);
    xn[p0 + threadIdx.x] = out;
  }
}

__global__ __launch_bounds__(256) void k_step(const f4* __restrict__ xc, f4* __restrict__ xn, const uint32_t* __restrict__ nb,
                                              const int* __restrict__ deg, int n, float L, int step) {
  __shared__ float sF[3 * PPB];
  const int p0 = blockIdx.x * PPB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  const f4 xi = xc[p];
  const int d = deg[p];
  const int nc = (d + W - 1) / W;
  f4 own = threadIdx.x < PPB ? xc[p0 + threadIdx.x] : xi;
  float fx = 0, fy = 0, fz = 0;
  for (int c = 0; c < nc; ++c) {
    const uint4 e = reinterpret_cast<const uint4*>(nb)[((size_t)c * n + p) * G + q];
    const f4 a = xc[e.x], b = xc[e.y], cc = xc[e.z], dd = xc[e.w];
    pair(xi, a, L, fx, fy, fz);
    pair(xi, b, L, fx, fy, fz);
    pair(xi, cc, L, fx, fy, fz);
    pair(xi, dd, L, fx, fy, fz);
  }
  f4 out;
  if (finish(fx, fy, fz, sF, p0, n, own, L, step + 1, out)) xn[p0 + threadIdx.x] = out;
}

// ---------------------------------------------------------------- B: persistent, data-tagged records
// status[0]: abort flag (a spin ran out), status[1]: total re-polls
__global__ __launch_bounds__(256) void k_persist_tag(f4* __restrict__ x0, f4* __restrict__ x1, const uint32_t* __restrict__ nb,
                                                     const int* __restrict__ deg, int n, float L, int steps, int* status) {
  __shared__ float sF[3 * PPB];
  const int p0 = blockIdx.x * PPB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  const int d = deg[p];
  const int nc = (d + W - 1) / W;
  int repolls = 0;
  bool dead = false;
  // own record stays in registers from step to step
  f4 own = load_sc1(&x0[threadIdx.x < PPB ? p0 + threadIdx.x : p]);
  for (int s = 0; s < steps && !dead; ++s) {
    f4* xc = (s & 1) ? x1 : x0;
    f4* xn = (s & 1) ? x0 : x1;
    // the row owner's position: lanes of the group read it from the group's integrate thread through LDS? keep simple:
    // poll it like a neighbor (it was stored by this block, same protocol)
    f4 xi;
    int spins = 0;
    for (;;) {
      xi = load_sc1(&xc[p]);
      if (__float_as_int(xi.w) == s) break;
      if (++spins > SPIN_LIMIT || load_i32_sc1(&status[0])) { dead = true; break; }
    }
    float fx = 0, fy = 0, fz = 0;
    for (int c = 0; c < nc && !dead; ++c) {
      const uint4 e = reinterpret_cast<const uint4*>(nb)[((size_t)c * n + p) * G + q];
      f4 a, b, cc, dd;
      for (;;) {
        load4_sc1(&xc[e.x], &xc[e.y], &xc[e.z], &xc[e.w], a, b, cc, dd);
        const bool ok = (__float_as_int(a.w) == s) & (__float_as_int(b.w) == s) & (__float_as_int(cc.w) == s) &
                        (__float_as_int(dd.w) == s);
        if (ok) break;
        ++repolls;
        if (++spins > SPIN_LIMIT || load_i32_sc1(&status[0])) { dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      pair(xi, a, L, fx, fy, fz);
      pair(xi, b, L, fx, fy, fz);
      pair(xi, cc, L, fx, fy, fz);
      pair(xi, dd, L, fx, fy, fz);
    }
    // a dead lane must still reach the barrier inside finish(); the whole block then leaves together
    dead = __syncthreads_or(dead);
    if (dead) {
      if (threadIdx.x == 0) store_i32_sc1(&status[0], 1);
      break;
    }
    f4 out;
    if (finish(fx, fy, fz, sF, p0, n, own, L, s + 1, out)) {
      store_sc1(&xn[p0 + threadIdx.x], out);
      own = out;
    }
    __syncthreads();  // sF is reused by the next step
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (repolls) atomicAdd(&status[1], repolls);
}

// ---------------------------------------------------------------- B2: as B, but a lane group gets its row owner's position
// from the block's LDS (written by the integrate threads) and the (single-chunk) index quad lives in registers:
// one fabric round trip per step is left on the dependent chain.
__global__ __launch_bounds__(256) void k_persist_tag2(f4* __restrict__ x0, f4* __restrict__ x1, const uint32_t* __restrict__ nb,
                                                      const int* __restrict__ deg, int n, float L, int steps, int* status) {
  __shared__ float sF[3 * PPB];
  __shared__ f4 sX[PPB];
  const int p0 = blockIdx.x * PPB;
  const int pl = threadIdx.x / G;
  const int p = p0 + pl, q = threadIdx.x % G;
  const uint4 e = reinterpret_cast<const uint4*>(nb)[(size_t)p * G + q];  // chunk 0 only (deg <= 32 in this probe)
  int repolls = 0;
  bool dead = false;
  f4 own = load_sc1(&x0[threadIdx.x < PPB ? p0 + threadIdx.x : p]);
  if (threadIdx.x < PPB) sX[threadIdx.x] = own;
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    f4* xc = (s & 1) ? x1 : x0;
    f4* xn = (s & 1) ? x0 : x1;
    const f4 xi = sX[pl];
    float fx = 0, fy = 0, fz = 0;
    f4 a, b, cc, dd;
    int spins = 0;
    for (;;) {
      load4_sc1(&xc[e.x], &xc[e.y], &xc[e.z], &xc[e.w], a, b, cc, dd);
      const bool ok = (__float_as_int(a.w) == s) & (__float_as_int(b.w) == s) & (__float_as_int(cc.w) == s) &
                      (__float_as_int(dd.w) == s);
      if (ok) break;
      ++repolls;
      if (++spins > SPIN_LIMIT || load_i32_sc1(&status[0])) { dead = true; break; }
    }
    pair(xi, a, L, fx, fy, fz);
    pair(xi, b, L, fx, fy, fz);
    pair(xi, cc, L, fx, fy, fz);
    pair(xi, dd, L, fx, fy, fz);
    dead = __syncthreads_or(dead);
    if (dead) {
      if (threadIdx.x == 0) store_i32_sc1(&status[0], 1);
      break;
    }
    f4 out;
    if (finish(fx, fy, fz, sF, p0, n, own, L, s + 1, out)) {
      store_sc1(&xn[p0 + threadIdx.x], out);
      own = out;
      sX[threadIdx.x] = out;
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (repolls) atomicAdd(&status[1], repolls);
}

// ---------------------------------------------------------------- C: persistent, per-block done[] words
__global__ __launch_bounds__(256) void k_persist_flag(f4* __restrict__ x0, f4* __restrict__ x1, const uint32_t* __restrict__ nb,
                                                      const int* __restrict__ deg, const int* __restrict__ dep,
                                                      const int* __restrict__ ndep, int* done, int n, float L, int steps,
                                                      int* status) {
  __shared__ float sF[3 * PPB];
  const int p0 = blockIdx.x * PPB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  const int d = deg[p];
  const int nc = (d + W - 1) / W;
  const int nd = ndep[blockIdx.x];
  const int my_dep = threadIdx.x < nd ? dep[blockIdx.x * MAXDEP + threadIdx.x] : -1;
  int repolls = 0;
  f4 own = load_sc1(&x0[threadIdx.x < PPB ? p0 + threadIdx.x : p]);
  for (int s = 0; s < steps; ++s) {
    f4* xc = (s & 1) ? x1 : x0;
    f4* xn = (s & 1) ? x0 : x1;
    // wait until every block this one reads from has published step s (done[b] >= s)
    bool dead = false;
    if (threadIdx.x < 64) {
      int spins = 0;
      for (;;) {
        const int v = my_dep >= 0 ? load_i32_sc1(&done[my_dep * 32]) : 0x7fffffff;
        if (__all(v >= s)) break;
        ++repolls;
        if (++spins > SPIN_LIMIT || load_i32_sc1(&status[0])) { dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    dead = __syncthreads_or(dead);
    if (dead) {
      if (threadIdx.x == 0) store_i32_sc1(&status[0], 1);
      break;
    }
    const f4 xi = load_sc1(&xc[p]);
    float fx = 0, fy = 0, fz = 0;
    for (int c = 0; c < nc; ++c) {
      const uint4 e = reinterpret_cast<const uint4*>(nb)[((size_t)c * n + p) * G + q];
      f4 a, b, cc, dd;
      load4_sc1(&xc[e.x], &xc[e.y], &xc[e.z], &xc[e.w], a, b, cc, dd);
      pair(xi, a, L, fx, fy, fz);
      pair(xi, b, L, fx, fy, fz);
      pair(xi, cc, L, fx, fy, fz);
      pair(xi, dd, L, fx, fy, fz);
    }
    f4 out;
    if (finish(fx, fy, fz, sF, p0, n, own, L, s + 1, out)) {
      store_sc1(&xn[p0 + threadIdx.x], out);
      own = out;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) store_i32_sc1(&done[blockIdx.x * 32], s + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (repolls && threadIdx.x == 0) atomicAdd(&status[1], repolls);
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 256;
  const int R2 = argc > 2 ? atoi(argv[2]) : 2;
  const int E = 32, n = E * E * E;
  const float L = (float)E;
  // brick order: bricks of 4 x 4 x 2 lattice sites = 32 particles = one block
  std::vector<int> id_of(n);  // lattice site -> particle index
  std::vector<int> site_of(n);
  {
    int idx = 0;
    for (int bz = 0; bz < E / 2; ++bz)
      for (int by = 0; by < E / 4; ++by)
        for (int bx = 0; bx < E / 4; ++bx)
          for (int z = 0; z < 2; ++z)
            for (int y = 0; y < 4; ++y)
              for (int x = 0; x < 4; ++x) {
                const int s = ((bz * 2 + z) * E + (by * 4 + y)) * E + bx * 4 + x;
                id_of[s] = idx;
                site_of[idx] = s;
                ++idx;
              }
  }
  std::vector<std::vector<uint32_t>> rows(n);
  int maxdeg = 0;
  for (int i = 0; i < n; ++i) {
    const int s = site_of[i], x = s % E, y = (s / E) % E, z = s / (E * E);
    for (int dz = -2; dz <= 2; ++dz)
      for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
          const int r2 = dx * dx + dy * dy + dz * dz;
          if (r2 == 0 || r2 > R2) continue;
          const int t = (((z + dz + E) % E) * E + (y + dy + E) % E) * E + (x + dx + E) % E;
          rows[i].push_back((uint32_t)id_of[t]);
        }
    std::sort(rows[i].begin(), rows[i].end());
    maxdeg = std::max(maxdeg, (int)rows[i].size());
  }
  const int chunks = (maxdeg + W - 1) / W;
  std::vector<uint32_t> nb((size_t)chunks * n * W);
  std::vector<int> deg(n);
  for (int i = 0; i < n; ++i) {
    deg[i] = (int)rows[i].size();
    for (int k = 0; k < chunks * W; ++k)
      nb[((size_t)(k / W) * n + i) * W + k % W] = k < deg[i] ? rows[i][k] : (uint32_t)i;
  }
  const int blocks = n / PPB;
  std::vector<int> dep((size_t)blocks * MAXDEP, -1), ndep(blocks);
  int maxdep = 0;
  for (int b = 0; b < blocks; ++b) {
    std::vector<int> s;
    for (int i = b * PPB; i < (b + 1) * PPB; ++i)
      for (uint32_t j : rows[i]) s.push_back((int)j / PPB);
    s.push_back(b);
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    if ((int)s.size() > MAXDEP) { fprintf(stderr, "too many deps %zu\n", s.size()); return 1; }
    ndep[b] = (int)s.size();
    maxdep = std::max(maxdep, ndep[b]);
    for (size_t k = 0; k < s.size(); ++k) dep[(size_t)b * MAXDEP + k] = s[k];
  }
  std::vector<float> xh(4 * (size_t)n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    const int s = site_of[i];
    xh[4 * i] = (float)(s % E) + 0.2f * ((float)rand() / RAND_MAX - 0.5f) + 0.5f;
    xh[4 * i + 1] = (float)((s / E) % E) + 0.2f * ((float)rand() / RAND_MAX - 0.5f) + 0.5f;
    xh[4 * i + 2] = (float)(s / (E * E)) + 0.2f * ((float)rand() / RAND_MAX - 0.5f) + 0.5f;
    int tag = 0;
    memcpy(&xh[4 * i + 3], &tag, 4);
  }
  printf("n %d blocks %d steps %d R2 %d maxdeg %d chunks %d max block deps %d\n", n, blocks, steps, R2, maxdeg, chunks, maxdep);

  int dev_blocks = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&dev_blocks, k_persist_tag, 256, 0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("occupancy api: %d blocks/CU x %d CUs = %d resident (need %d)\n", dev_blocks, prop.multiProcessorCount,
         dev_blocks * prop.multiProcessorCount, blocks);
  // margin: the API can be one block per CU high (MI355X_MICROARCH.md, residency)
  if ((dev_blocks - 1) * prop.multiProcessorCount < blocks) { fprintf(stderr, "grid would not be co-resident\n"); return 1; }

  f4 *x0, *x1, *xa, *xb;
  uint32_t* nbd;
  int *degd, *depd, *ndepd, *done, *status;
  CK(hipMalloc(&x0, 16 * (size_t)n));
  CK(hipMalloc(&x1, 16 * (size_t)n));
  CK(hipMalloc(&xa, 16 * (size_t)n));
  CK(hipMalloc(&xb, 16 * (size_t)n));
  CK(hipMalloc(&nbd, nb.size() * 4));
  CK(hipMalloc(&degd, 4 * (size_t)n));
  CK(hipMalloc(&depd, dep.size() * 4));
  CK(hipMalloc(&ndepd, 4 * (size_t)blocks));
  CK(hipMalloc(&done, 4 * 32 * (size_t)blocks));
  CK(hipMalloc(&status, 64));
  CK(hipMemcpy(nbd, nb.data(), nb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(degd, deg.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
  CK(hipMemcpy(depd, dep.data(), dep.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ndepd, ndep.data(), 4 * (size_t)blocks, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ra(4 * (size_t)n), rb(4 * (size_t)n), rc(4 * (size_t)n);
  float ms;

  // ---- A
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(xa, xh.data(), 16 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemset(xb, 0xff, 16 * (size_t)n));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int s = 0; s < steps; ++s)
      hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, 0, (s & 1) ? xb : xa, (s & 1) ? xa : xb, nbd, degd, n, L, s);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("A per-step launches : %8.3f us/step\n", 1e3 * ms / steps);
  }
  CK(hipMemcpy(ra.data(), (steps & 1) ? xb : xa, 16 * (size_t)n, hipMemcpyDeviceToHost));

  // ---- A2: the same step with 1024-thread blocks (one per CU)
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(xa, xh.data(), 16 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemset(xb, 0xff, 16 * (size_t)n));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int s = 0; s < steps; ++s)
      hipLaunchKernelGGL(k_step_nt<1024>, dim3(n / (1024 / G)), dim3(1024), 0, 0, (s & 1) ? xb : xa, (s & 1) ? xa : xb, nbd,
                         degd, n, L, s);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("A2 launches, 1024 thr: %7.3f us/step\n", 1e3 * ms / steps);
  }
  CK(hipMemcpy(rb.data(), (steps & 1) ? xb : xa, 16 * (size_t)n, hipMemcpyDeviceToHost));
  printf("A2 == A bitwise: %s\n", memcmp(ra.data(), rb.data(), 16 * (size_t)n) == 0 ? "yes" : "NO");
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(xa, xh.data(), 16 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int s = 0; s < steps; ++s)
      hipLaunchKernelGGL(k_step_nt<512>, dim3(n / (512 / G)), dim3(512), 0, 0, (s & 1) ? xb : xa, (s & 1) ? xa : xb, nbd,
                         degd, n, L, s);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("A3 launches, 512 thr : %7.3f us/step\n", 1e3 * ms / steps);
  }

  // ---- B2
  if (chunks == 1)
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(x0, xh.data(), 16 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemset(x1, 0xff, 16 * (size_t)n));
    CK(hipMemset(status, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_persist_tag2, dim3(blocks), dim3(256), 0, 0, x0, x1, nbd, degd, n, L, steps, status);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    int st[2];
    CK(hipMemcpy(st, status, 8, hipMemcpyDeviceToHost));
    printf("B2 persistent, tags, LDS own: %7.3f us/step   abort %d  re-polls/step/block %.2f\n", 1e3 * ms / steps, st[0],
           (double)st[1] / steps / blocks);
    CK(hipMemcpy(rb.data(), (steps & 1) ? x1 : x0, 16 * (size_t)n, hipMemcpyDeviceToHost));
    printf("B2 == A bitwise: %s\n", memcmp(ra.data(), rb.data(), 16 * (size_t)n) == 0 ? "yes" : "NO");
  }

  // ---- B
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(x0, xh.data(), 16 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemset(x1, 0xff, 16 * (size_t)n));
    CK(hipMemset(status, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_persist_tag, dim3(blocks), dim3(256), 0, 0, x0, x1, nbd, degd, n, L, steps, status);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    int st[2];
    CK(hipMemcpy(st, status, 8, hipMemcpyDeviceToHost));
    printf("B persistent, tags  : %8.3f us/step   abort %d  re-polls/step/block %.2f\n", 1e3 * ms / steps, st[0],
           (double)st[1] / steps / blocks);
  }
  CK(hipMemcpy(rb.data(), (steps & 1) ? x1 : x0, 16 * (size_t)n, hipMemcpyDeviceToHost));
  printf("B == A bitwise: %s\n", memcmp(ra.data(), rb.data(), 16 * (size_t)n) == 0 ? "yes" : "NO");

  // ---- C
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(x0, xh.data(), 16 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemset(x1, 0xff, 16 * (size_t)n));
    CK(hipMemset(status, 0, 64));
    CK(hipMemset(done, 0, 4 * 32 * (size_t)blocks));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_persist_flag, dim3(blocks), dim3(256), 0, 0, x0, x1, nbd, degd, depd, ndepd, done, n, L, steps,
                       status);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    int st[2];
    CK(hipMemcpy(st, status, 8, hipMemcpyDeviceToHost));
    printf("C persistent, flags : %8.3f us/step   abort %d  re-polls/step/block %.2f\n", 1e3 * ms / steps, st[0],
           (double)st[1] / steps / blocks);
  }
  CK(hipMemcpy(rc.data(), (steps & 1) ? x1 : x0, 16 * (size_t)n, hipMemcpyDeviceToHost));
  printf("C == A bitwise: %s\n", memcmp(ra.data(), rc.data(), 16 * (size_t)n) == 0 ? "yes" : "NO");
  return 0;
}
