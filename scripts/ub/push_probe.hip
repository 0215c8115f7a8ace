// Feasibility probe (GPU box only, not part of libchx): can the step kernel's SECOND dependent round trip -- index
// quad, then the gathered neighbor positions -- be removed by PUSHING positions instead of pulling them?
//
//   A  pull (today)   lane loads its index quad, gathers four float4 positions, pair math, group reduce, LDS,
//                     one thread per particle integrates and stores the new position
//   B  push           every row owns a MIRROR of its neighbors' positions (float4 per list entry, contiguous per
//                     row); a lane loads its four mirror entries together with everything else in the launch's
//                     first round trip; after the integrate every lane stores the particle's new position into the
//                     mirror slots of the rows that list it (reverse-slot table, built with the list)
//
// Geometry mimics config C3 with the pruned list: N = 32768 on a 32^3 lattice in brick order, G = 4 lanes per
// particle, 64 particles per block, 14 symmetric neighbors per particle, rows padded to 16 entries with the row's
// own index.  Both variants must end in bit-identical positions.
//
//   hipcc --offload-arch=gfx950 -O3 -o push_probe push_probe.hip && ./push_probe [launches]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int G = 4;
constexpr int PPB = 256 / G;  // 64 particles per block
constexpr int W = 4 * G;      // 16 entries per row (one chunk)
typedef float f4 __attribute__((ext_vector_type(4)));

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
// some filler arithmetic of about the weight of the Philox draw + BAOAB (so the tail is not unrealistically short)
__device__ __forceinline__ float filler(float v, int p, int step) {
  uint32_t h = (uint32_t)p * 0x9E3779B9u + (uint32_t)step;
#pragma unroll
  for (int r = 0; r < 10; ++r) h = __umulhi(h | 1u, 0xD2511F53u) ^ (h * 0xCD9E8D57u);
  return v + 1e-9f * (float)(h >> 20);
}
__device__ __forceinline__ void store_wt4(f4* p, f4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <bool WT>
__global__ __launch_bounds__(256) void k_pull(const f4* __restrict__ xc, f4* __restrict__ xn, const uint32_t* __restrict__ nb,
                                              int n, float L, int step) {
  __shared__ float sF[3 * PPB];
  const int p0 = blockIdx.x * PPB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  const uint4 e = reinterpret_cast<const uint4*>(nb)[(size_t)p * G + q];
  const f4 xi = xc[p];
  const f4 own = threadIdx.x < PPB ? xc[p0 + threadIdx.x] : xi;
  const f4 a = xc[e.x], b = xc[e.y], c = xc[e.z], d = xc[e.w];
  float fx = 0, fy = 0, fz = 0;
  pair(xi, a, L, fx, fy, fz);
  pair(xi, b, L, fx, fy, fz);
  pair(xi, c, L, fx, fy, fz);
  pair(xi, d, L, fx, fy, fz);
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
  if (threadIdx.x < PPB) {
    f4 out;
    out.x = wrapc(filler(own.x + sF[3 * threadIdx.x], p0 + threadIdx.x, step), L);
    out.y = wrapc(own.y + sF[3 * threadIdx.x + 1], L);
    out.z = wrapc(own.z + sF[3 * threadIdx.x + 2], L);
    out.w = 0.f;
    if (WT) store_wt4(&xn[p0 + threadIdx.x], out);
    else xn[p0 + threadIdx.x] = out;
  }
}

template <bool WT>
__global__ __launch_bounds__(256) void k_push(const f4* __restrict__ xc, f4* __restrict__ xn, const f4* __restrict__ mc,
                                              f4* __restrict__ mn, const uint32_t* __restrict__ rs, int n, float L, int step) {
  __shared__ float sF[3 * PPB];
  __shared__ f4 sX[PPB];
  const int p0 = blockIdx.x * PPB;
  const int p = p0 + threadIdx.x / G, q = threadIdx.x % G;
  const f4* row = mc + ((size_t)p * G + q) * 4;
  const uint4 r4 = reinterpret_cast<const uint4*>(rs)[(size_t)p * G + q];
  const f4 xi = xc[p];
  const f4 own = threadIdx.x < PPB ? xc[p0 + threadIdx.x] : xi;
  const f4 a = row[0], b = row[1], c = row[2], d = row[3];
  float fx = 0, fy = 0, fz = 0;
  pair(xi, a, L, fx, fy, fz);
  pair(xi, b, L, fx, fy, fz);
  pair(xi, c, L, fx, fy, fz);
  pair(xi, d, L, fx, fy, fz);
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
  if (threadIdx.x < PPB) {
    f4 out;
    out.x = wrapc(filler(own.x + sF[3 * threadIdx.x], p0 + threadIdx.x, step), L);
    out.y = wrapc(own.y + sF[3 * threadIdx.x + 1], L);
    out.z = wrapc(own.z + sF[3 * threadIdx.x + 2], L);
    out.w = 0.f;
    sX[threadIdx.x] = out;
    if (WT) store_wt4(&xn[p0 + threadIdx.x], out);
    else xn[p0 + threadIdx.x] = out;
  }
  __syncthreads();
  const f4 mine = sX[threadIdx.x / G];
  if (WT) {
    store_wt4(&mn[r4.x], mine), store_wt4(&mn[r4.y], mine), store_wt4(&mn[r4.z], mine), store_wt4(&mn[r4.w], mine);
  } else {
    mn[r4.x] = mine, mn[r4.y] = mine, mn[r4.z] = mine, mn[r4.w] = mine;
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000;
  const int S = 32, n = S * S * S;
  const float L = (float)S;
  // brick order: 4 x 4 x 4 bricks of 64 lattice points = one block
  std::vector<int> id_of(n), cell_of(n);
  {
    int k = 0;
    for (int bz = 0; bz < S; bz += 4)
      for (int by = 0; by < S; by += 4)
        for (int bx = 0; bx < S; bx += 4)
          for (int z = bz; z < bz + 4; ++z)
            for (int y = by; y < by + 4; ++y)
              for (int x = bx; x < bx + 4; ++x) {
                id_of[(z * S + y) * S + x] = k;
                cell_of[k] = (z * S + y) * S + x;
                ++k;
              }
  }
  // 14 symmetric neighbor offsets: the 6 faces + 8 of the 12 edges
  const int off[14][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 0, 1},  {-1, 0, -1},
                          {1, 0, -1}, {-1, 0, 1}, {0, 1, 1}, {0, -1, -1}, {0, 1, -1}, {0, -1, 1}};
  std::vector<uint32_t> nb((size_t)n * W), rs((size_t)n * W);
  for (int p = 0; p < n; ++p) {
    const int c = cell_of[p], x = c % S, y = (c / S) % S, z = c / (S * S);
    for (int k = 0; k < W; ++k) {
      if (k < 14) {
        const int xx = (x + off[k][0] + S) % S, yy = (y + off[k][1] + S) % S, zz = (z + off[k][2] + S) % S;
        nb[(size_t)p * W + k] = (uint32_t)id_of[(zz * S + yy) * S + xx];
      } else {
        nb[(size_t)p * W + k] = (uint32_t)p;
      }
    }
  }
  // reverse slots: entry k of row p lists j; p sits in row j at the slot of the opposite offset (k ^ 1)
  for (int p = 0; p < n; ++p)
    for (int k = 0; k < W; ++k) {
      const uint32_t j = nb[(size_t)p * W + k];
      rs[(size_t)p * W + k] = k < 14 ? j * W + (uint32_t)(k ^ 1) : (uint32_t)p * W + k;
    }
  std::vector<f4> x0(n), m0((size_t)n * W);
  for (int p = 0; p < n; ++p) {
    const int c = cell_of[p];
    x0[p] = f4{(float)(c % S) + 0.25f * (float)((p * 7919) % 13) / 13.f, (float)((c / S) % S) + 0.1f, (float)(c / (S * S)) + 0.2f, 0.f};
  }
  for (int p = 0; p < n; ++p)
    for (int k = 0; k < W; ++k) m0[(size_t)p * W + k] = x0[nb[(size_t)p * W + k]];
  f4 *xa, *xb, *ma, *mb;
  uint32_t *dnb, *drs;
  CK(hipMalloc(&xa, sizeof(f4) * n));
  CK(hipMalloc(&xb, sizeof(f4) * n));
  CK(hipMalloc(&ma, sizeof(f4) * n * W));
  CK(hipMalloc(&mb, sizeof(f4) * n * W));
  CK(hipMalloc(&dnb, 4 * (size_t)n * W));
  CK(hipMalloc(&drs, 4 * (size_t)n * W));
  CK(hipMemcpy(dnb, nb.data(), 4 * (size_t)n * W, hipMemcpyHostToDevice));
  CK(hipMemcpy(drs, rs.data(), 4 * (size_t)n * W, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<f4> ref(n), got(n);
  auto run = [&](const char* name, int variant, bool wt) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemcpy(xa, x0.data(), sizeof(f4) * n, hipMemcpyHostToDevice));
      CK(hipMemcpy(ma, m0.data(), sizeof(f4) * n * W, hipMemcpyHostToDevice));
      CK(hipMemcpy(mb, m0.data(), sizeof(f4) * n * W, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int s = 0; s < launches; ++s) {
        f4 *xc = (s & 1) ? xb : xa, *xn = (s & 1) ? xa : xb, *mc = (s & 1) ? mb : ma, *mn = (s & 1) ? ma : mb;
        if (variant == 0) {
          if (wt) hipLaunchKernelGGL(k_pull<true>, dim3(n / PPB), dim3(256), 0, 0, xc, xn, dnb, n, L, s);
          else hipLaunchKernelGGL(k_pull<false>, dim3(n / PPB), dim3(256), 0, 0, xc, xn, dnb, n, L, s);
        } else {
          if (wt) hipLaunchKernelGGL(k_push<true>, dim3(n / PPB), dim3(256), 0, 0, xc, xn, mc, mn, drs, n, L, s);
          else hipLaunchKernelGGL(k_push<false>, dim3(n / PPB), dim3(256), 0, 0, xc, xn, mc, mn, drs, n, L, s);
        }
      }
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-44s %7.3f us per launch\n", name, 1e3 * ms / launches);
    }
    CK(hipMemcpy(got.data(), (launches & 1) ? xb : xa, sizeof(f4) * n, hipMemcpyDeviceToHost));
  };
  run("A  pull (index quad -> gathers), plain", 0, false);
  ref = got;
  run("A  pull, write-through stores", 0, true);
  printf("   same bits as A: %s\n", memcmp(ref.data(), got.data(), sizeof(f4) * n) == 0 ? "yes" : "NO");
  run("B  push (mirror rows + scatter), plain", 1, false);
  printf("   same bits as A: %s\n", memcmp(ref.data(), got.data(), sizeof(f4) * n) == 0 ? "yes" : "NO");
  run("B  push, write-through stores", 1, true);
  printf("   same bits as A: %s\n", memcmp(ref.data(), got.data(), sizeof(f4) * n) == 0 ? "yes" : "NO");
  return 0;
}
