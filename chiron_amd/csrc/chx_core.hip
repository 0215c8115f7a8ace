// libchx: error plumbing, host helpers, Space kernels (displacement / wrap),
// permutation helpers and small elementwise kernels.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "chx_common.h"
#include "chx_launch.h"

static thread_local char g_err[512] = "";

void chx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// A dispatch through the step queue that failed (chx_launch.h: a kernarg block that does not fit the kernel, a queue in
// error) is remembered per thread until the entry point that made it reaches its chx_check_launch: the kernel was
// dropped, and what follows on the queue would otherwise run on arrays nobody wrote.
static thread_local int g_launch_rc = CHX_OK;
void chx_note_launch_failure(int rc) {
  if (rc != CHX_OK && g_launch_rc == CHX_OK) g_launch_rc = rc;
}

int chx_check_launch(const char* what) {
  if (g_launch_rc != CHX_OK) {
    const int rc = g_launch_rc;
    g_launch_rc = CHX_OK;
    (void)hipGetLastError();
    char keep[400];
    snprintf(keep, sizeof(keep), "%s", g_err);  // (the dispatcher's own text)
    chx_set_error("%s: a launch was dropped (%s)", what, keep);
    return rc;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    chx_set_error("%s: %s", what, hipGetErrorString(e));
    return CHX_ERR_LAUNCH;
  }
  return CHX_OK;
}

extern "C" int chx_version(void) { return CHX_ABI_VERSION; }
extern "C" const char* chx_last_error(void) { return g_err; }

#ifndef CHX_SOURCE_HASH
#define CHX_SOURCE_HASH "unhashed"
#endif
// the marker prefix lets build.py find the hash in the binary without loading it
static const char g_source_hash[] = "CHX_SOURCE_HASH=" CHX_SOURCE_HASH;
extern "C" const char* chx_source_hash(void) { return g_source_hash + 16; }
extern "C" int64_t chx_sizeof_langevin_t(void) { return (int64_t)sizeof(chx_langevin_t); }
extern "C" int64_t chx_sizeof_mc_t(void) { return (int64_t)sizeof(chx_mc_t); }
extern "C" int64_t chx_sizeof_build_riders_t(void) { return (int64_t)sizeof(chx_build_riders_t); }

// Largest fp32 t with sqrtf(t) < c.  Host sqrtf is correctly rounded and
// monotone, as is the device __fsqrt_rn, so the equivalence is exact.
extern "C" float chx_cutoff_threshold_h(float c) {
  if (!(c > 0.0f)) return -1.0f;  // sqrt(t) < c impossible for t >= 0 -> r2 <= -1 never true
  if (isinf(c)) return INFINITY;
  float t = c * c;
  while (sqrtf(t) < c) t = nextafterf(t, INFINITY);
  while (!(sqrtf(t) < c)) t = nextafterf(t, -INFINITY);
  return t;
}

// ---------------------------------------------------------------------------
// Space.displacement / Space.wrap
// ---------------------------------------------------------------------------
__global__ void k_displacement(const float* __restrict__ x1, const float* __restrict__ x2, int64_t n,
                               Box3 box, float* __restrict__ rij, float* __restrict__ dist) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float rx, ry, rz;
  displacement(box, x1[3 * i], x1[3 * i + 1], x1[3 * i + 2], x2[3 * i], x2[3 * i + 1],
               x2[3 * i + 2], rx, ry, rz);
  rij[3 * i] = rx;
  rij[3 * i + 1] = ry;
  rij[3 * i + 2] = rz;
  dist[i] = fsqrt(norm2(rx, ry, rz));
}

extern "C" int chx_displacement(const float* x1, const float* x2, int64_t n, chx_box_t box,
                                float* rij_out, float* dist_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(x1 && x2 && rij_out && dist_out, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_displacement, dim3((unsigned)blocks), dim3(256), 0, chx_stream(stream), x1, x2,
                     n, make_box3(box), rij_out, dist_out);
  return chx_check_launch("chx_displacement");
}

__global__ void k_wrap(const float* __restrict__ xin, float* __restrict__ xout, int64_t n, Box3 box) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = xin[3 * i], y = xin[3 * i + 1], z = xin[3 * i + 2];
  if (box.periodic) {
    x = wrap1(x, box.lx);
    y = wrap1(y, box.ly);
    z = wrap1(z, box.lz);
  }
  xout[3 * i] = x;
  xout[3 * i + 1] = y;
  xout[3 * i + 2] = z;
}

extern "C" int chx_wrap(const float* xin, float* xout, int64_t n, chx_box_t box,
                        chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(xin && xout, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_wrap, dim3((unsigned)blocks), dim3(256), 0, chx_stream(stream), xin, xout, n,
                     make_box3(box));
  return chx_check_launch("chx_wrap");
}

// ---------------------------------------------------------------------------
// permutation helpers (internal particle order <-> user order)
// ---------------------------------------------------------------------------
__global__ void k_permute_pack(const float* __restrict__ src3, const uint32_t* __restrict__ perm,
                               int n, float* __restrict__ dst, int stride) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t o = perm ? perm[p] : (uint32_t)p;
  float a = src3[3 * (size_t)o], b = src3[3 * (size_t)o + 1], c = src3[3 * (size_t)o + 2];
  if (stride == 4) {
    reinterpret_cast<float4*>(dst)[p] = make_float4(a, b, c, 0.0f);
  } else {
    dst[3 * (size_t)p] = a;
    dst[3 * (size_t)p + 1] = b;
    dst[3 * (size_t)p + 2] = c;
  }
}

extern "C" int chx_permute_pack(const float* src3, const uint32_t* perm, int32_t n, float* dst,
                                int32_t dst_stride, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  CHX_REQUIRE(dst_stride == 3 || dst_stride == 4, "dst_stride must be 3 or 4");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(src3 && dst, "null pointer");
  hipLaunchKernelGGL(k_permute_pack, dim3((n + 255) / 256), dim3(256), 0, chx_stream(stream), src3,
                     perm, n, dst, dst_stride);
  return chx_check_launch("chx_permute_pack");
}

__global__ void k_permute_unpack(const float* __restrict__ src, int stride,
                                 const uint32_t* __restrict__ perm, int n, float* __restrict__ dst3) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t o = perm ? perm[p] : (uint32_t)p;
  float a = src[(size_t)stride * p], b = src[(size_t)stride * p + 1], c = src[(size_t)stride * p + 2];
  dst3[3 * (size_t)o] = a;
  dst3[3 * (size_t)o + 1] = b;
  dst3[3 * (size_t)o + 2] = c;
}

extern "C" int chx_permute_unpack(const float* src, int32_t src_stride, const uint32_t* perm,
                                  int32_t n, float* dst3, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  CHX_REQUIRE(src_stride == 3 || src_stride == 4, "src_stride must be 3 or 4");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(src && dst3, "null pointer");
  CHX_LAUNCH(k_permute_unpack, dim3((n + 255) / 256), dim3(256), chx_stream_q(stream), src, src_stride, perm, n, dst3);
  return chx_check_launch("chx_permute_unpack");
}

// inverse of chx_pack_langevin in one launch: positions (rows of 4 floats) and velocities (rows of 3) back to user order
__global__ void k_unpack_langevin(const float4* __restrict__ x4, const float* __restrict__ v,
                                  const uint32_t* __restrict__ perm, int n, float* __restrict__ x3,
                                  float* __restrict__ v3) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t o = perm ? perm[p] : (uint32_t)p;
  const float4 x = x4[p];
  const float a = v[3 * (size_t)p], b = v[3 * (size_t)p + 1], c = v[3 * (size_t)p + 2];
  x3[3 * o] = x.x;
  x3[3 * o + 1] = x.y;
  x3[3 * o + 2] = x.z;
  v3[3 * o] = a;
  v3[3 * o + 1] = b;
  v3[3 * o + 2] = c;
}

extern "C" int chx_unpack_langevin(const float* x4, const float* v, const uint32_t* perm, int32_t n,
                                   float* x3_out, float* v3_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(x4 && v && x3_out && v3_out, "null pointer");
  CHX_LAUNCH(k_unpack_langevin, dim3((n + 255) / 256), dim3(256), chx_stream_q(stream), reinterpret_cast<const float4*>(x4), v,
             perm, n, x3_out, v3_out);
  return chx_check_launch("chx_unpack_langevin");
}

__global__ void k_invert_perm(const uint32_t* __restrict__ perm, int n, uint32_t* __restrict__ inv) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) inv[perm[p]] = (uint32_t)p;
}

extern "C" int chx_invert_perm(const uint32_t* perm, int32_t n, uint32_t* inv_out,
                               chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(perm && inv_out, "null pointer");
  hipLaunchKernelGGL(k_invert_perm, dim3((n + 255) / 256), dim3(256), 0, chx_stream(stream), perm, n,
                     inv_out);
  return chx_check_launch("chx_invert_perm");
}

// ---------------------------------------------------------------------------
// barostat geometry: out = in * s  (chiron/mcmc.py:977-983)
// ---------------------------------------------------------------------------
__global__ void k_scale(const float* __restrict__ in, float* __restrict__ out, int64_t count, float s) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = fmul(in[i], s);
}

extern "C" int chx_scale(const float* in, float* out, int64_t count, float s, chx_stream_t stream) {
  CHX_REQUIRE(count >= 0, "count < 0");
  if (count == 0) return CHX_OK;
  CHX_REQUIRE(in && out, "null pointer");
  int64_t blocks = (count + 255) / 256;
  hipLaunchKernelGGL(k_scale, dim3((unsigned)blocks), dim3(256), 0, chx_stream(stream), in, out,
                     count, s);
  return chx_check_launch("chx_scale");
}

// ---------------------------------------------------------------------------
// control-block plumbing of the Langevin loop.  Both are ordinary kernels on the caller's
// stream, so the hot loop never touches the runtime's copy engines / blit queues.
// ---------------------------------------------------------------------------
__global__ void k_ctrl_reset(int32_t* __restrict__ ctrl) {
  // [3] = no halt; [4],[5] = -2: "inner list stale" words hold the refresh epoch they were raised in
  // (k_lj_step), -2 is no epoch
  if (threadIdx.x < CHX_CTRL_WORDS) ctrl[threadIdx.x] = chx_ctrl_reset_word((int)threadIdx.x);
}

extern "C" int chx_ctrl_reset(int32_t* ctrl, chx_stream_t stream) {
  CHX_REQUIRE(ctrl, "null pointer");
  hipLaunchKernelGGL(k_ctrl_reset, dim3(1), dim3(64), 0, chx_stream(stream), ctrl);
  return chx_check_launch("chx_ctrl_reset");
}

// ---------------------------------------------------------------------------
// The step queue's start-up check (chx_aql.hip: selfcheck_fences): before it sends steady-state packets without cache
// fences it makes sure, on THIS card with THIS firmware, that a dispatch without an acquire fence reads no stale data.
// k_selfcheck_fill (one workgroup) rewrites a 4 KB buffer with write-through stores; k_selfcheck_read (every CU, several
// workgroups each) reads all of it back with plain vector loads and one scalar load and counts the words that are not
// the value just written -- after the previous round left the old value in every L1, scalar cache and L2.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_selfcheck_fill(int32_t* buf, int val) {
  typedef int i4v __attribute__((ext_vector_type(4)));
  const i4v t = {val, val, val, val};
  int32_t* p = buf + 4 * threadIdx.x;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__global__ __launch_bounds__(256) void k_selfcheck_read(const int32_t* buf, int val, int32_t* bad) {
  const int4 t = reinterpret_cast<const int4*>(buf)[threadIdx.x];
  const int s0 = buf[(blockIdx.x & 7) * 32];  // (uniform address: a scalar load)
  const int n = (t.x != val) + (t.y != val) + (t.z != val) + (t.w != val);
  if (n) atomicAdd(bad, n);
  if (threadIdx.x == 0 && s0 != val) atomicAdd(bad + 1, 1);
}
// (referenced from the host so that the stubs, and with them the names hipKernelNameRefByPtr hands out, exist)
// The step queue's debug mode (chx_queue_debug, include/chx.h): a packet that does nothing for `ticks` periods of the
// 100 MHz wall clock.  One lane watches the clock; the clock only moves forward, so the wave ends.
__global__ __launch_bounds__(256) void k_debug_spin(long long ticks) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (ticks > 100000000ll) ticks = 100000000ll;  // one second at most
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
const void* chx_debug_spin_fn() { return reinterpret_cast<const void*>(&k_debug_spin); }
const void* chx_selfcheck_fill_fn() { return reinterpret_cast<const void*>(&k_selfcheck_fill); }
const void* chx_selfcheck_read_fn() { return reinterpret_cast<const void*>(&k_selfcheck_read); }

// copy n <= 64 ints to host-mapped pinned memory with system-scope visibility
__global__ void k_publish(const int32_t* __restrict__ src, int32_t* __restrict__ host_dst, int n) {
  if ((int)threadIdx.x < n) {
    __hip_atomic_store(&host_dst[threadIdx.x], src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
}

extern "C" int chx_publish(const int32_t* src, int32_t* host_mapped_dst, int32_t n, chx_stream_t stream) {
  CHX_REQUIRE(src && host_mapped_dst, "null pointer");
  CHX_REQUIRE(n >= 1 && n <= 64, "n must be in 1..64");
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, chx_stream(stream), src, host_mapped_dst, n);
  return chx_check_launch("chx_publish");
}

// ---------------------------------------------------------------------------
// Optional JAX-compatible stream: jax.random.normal(key, (count,)) for a classic
// (non-partitionable) threefry key, so that a run seeded like the reference draws the reference's numbers
// (chiron/utils.py:29-38,142; chiron/integrators.py:124,179,185; chiron/mcmc.py:731).  Algorithms restated
// from their publications: Threefry-2x32-20 (Salmon et al., SC'11); jax's random_bits layout (the counters
// 0..count-1, padded to even length, are split in HALVES that form the two input words, the two output halves
// are concatenated); uniform = bitcast((bits >> 9) | 0x3F800000) - 1 mapped to [nextafter(-1, 0), 1);
// normal = sqrt(2) erfinv(u) with the single-precision polynomial of M. Giles (GPU Computing Gems, 2010)
// that XLA evaluates for float32.  The product's own generator stays Philox (chx_common.h).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int d) { return (x << d) | (x >> (32 - d)); }

__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0];
  x1 += ks[1];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      x0 += x1;
      x1 = rotl32(x1, rot[i % 2][r]) ^ x0;
    }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
}

__device__ __forceinline__ float erfinv_giles(float x) {
  float w = -logf(fmul(fsub(1.0f, x), fadd(1.0f, x)));
  float p;
  if (w < 5.0f) {
    w = fsub(w, 2.5f);
    p = 2.81022636e-08f;
    p = fadd(3.43273939e-07f, fmul(p, w));
    p = fadd(-3.5233877e-06f, fmul(p, w));
    p = fadd(-4.39150654e-06f, fmul(p, w));
    p = fadd(0.00021858087f, fmul(p, w));
    p = fadd(-0.00125372503f, fmul(p, w));
    p = fadd(-0.00417768164f, fmul(p, w));
    p = fadd(0.246640727f, fmul(p, w));
    p = fadd(1.50140941f, fmul(p, w));
  } else {
    w = fsub(fsqrt(w), 3.0f);
    p = -0.000200214257f;
    p = fadd(0.000100950558f, fmul(p, w));
    p = fadd(0.00134934322f, fmul(p, w));
    p = fadd(-0.00367342844f, fmul(p, w));
    p = fadd(0.00573950773f, fmul(p, w));
    p = fadd(-0.0076224613f, fmul(p, w));
    p = fadd(0.00943887047f, fmul(p, w));
    p = fadd(1.00167406f, fmul(p, w));
    p = fadd(2.83297682f, fmul(p, w));
  }
  return fmul(p, x);
}

__global__ void k_jax_normal(uint32_t k0, uint32_t k1, int64_t count, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const int64_t half = (count + 1) / 2;  // counters are padded to an even length
  const bool second = e >= half;
  const int64_t blk = second ? e - half : e;
  uint32_t x0 = (uint32_t)blk;
  uint32_t x1 = (blk + half < count) ? (uint32_t)(blk + half) : 0u;  // the padding counter is 0
  threefry2x32(k0, k1, x0, x1);
  const uint32_t bits = second ? x1 : x0;
  const float f = fsub(__uint_as_float((bits >> 9) | 0x3F800000u), 1.0f);
  const float lo = -0.99999994f;  // nextafter(-1, 0)
  const float u = fmaxf(lo, fadd(fmul(f, fsub(1.0f, lo)), lo));
  out[e] = fmul(1.41421356237309504880f, erfinv_giles(u));
}

extern "C" int chx_jax_normal(uint32_t k0, uint32_t k1, int64_t count, float* out, chx_stream_t stream) {
  CHX_REQUIRE(count >= 0, "count < 0");
  CHX_REQUIRE(count < ((int64_t)1 << 32), "count must fit the 32-bit threefry counter");
  if (count == 0) return CHX_OK;
  CHX_REQUIRE(out, "null pointer");
  hipLaunchKernelGGL(k_jax_normal, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, chx_stream(stream), k0, k1, count,
                     out);
  return chx_check_launch("chx_jax_normal");
}

// Philox4x32-10 on the host (same constants and round structure as chx_common.h: philox4x32)
extern "C" void chx_philox4x32_h(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t* out4_h) {
  if (!out4_h) return;
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  out4_h[0] = c0, out4_h[1] = c1, out4_h[2] = c2, out4_h[3] = c3;
}
