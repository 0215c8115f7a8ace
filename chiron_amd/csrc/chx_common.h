// Shared device helpers for libchx (gfx950 only; wavefront = 64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/chx.h"

#define CHX_WAVE 64

void chx_set_error(const char* fmt, ...);
int chx_check_launch(const char* what);
// (chx_launch.h: a dropped dispatch is reported by the entry point's chx_check_launch)
void chx_note_launch_failure(int rc);

#define CHX_REQUIRE(cond, msg)             \
  do {                                     \
    if (!(cond)) {                         \
      chx_set_error("%s: %s", __func__, msg); \
      return CHX_ERR_INVALID;              \
    }                                      \
  } while (0)

// The stream an entry point launches on.  Step launches may be outstanding on the step queue paired with it (chx_aql.h):
// anything that goes to the HIP stream waits for them first, so that the caller's one-stream picture holds.
extern int chx_aql_open_count;
void chx_aql_join(hipStream_t st);
static inline hipStream_t chx_stream(chx_stream_t s) {
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  if (__atomic_load_n(&chx_aql_open_count, __ATOMIC_RELAXED) > 0) chx_aql_join(st);
  return st;
}
// (the step launches themselves: they go behind the outstanding ones)
static inline hipStream_t chx_stream_steps(chx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
// (entry points whose launches all go through CHX_LAUNCH, chx_launch.h: they ride on the queue while it is open, so
// they do not join)
static inline hipStream_t chx_stream_q(chx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// word t of a freshly reset control block (include/chx.h: chx_langevin_t::ctrl, chx_ctrl_reset)
__host__ __device__ __forceinline__ int chx_ctrl_reset_word(int t) {
  return (t == 3 || t == 12) ? 0x7FFFFFFF : ((t == 4 || t == 5) ? -2 : ((t == 9 || t == 10 || t == 11) ? -1 : 0));
}

struct Box3 {
  int periodic;
  float lx, ly, lz;
  float hx, hy, hz;  // L * 0.5f (exact)
};

static inline Box3 make_box3(chx_box_t b) {
  Box3 r;
  r.periodic = b.periodic;
  r.lx = b.lx;
  r.ly = b.ly;
  r.lz = b.lz;
  r.hx = b.lx * 0.5f;
  r.hy = b.ly * 0.5f;
  r.hz = b.lz * 0.5f;
  return r;
}

// ---------------------------------------------------------------------------
// exact fp32 building blocks (one IEEE rounding per operation, never contracted)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }
// NB: HIP's __fsqrt_rn maps to the 1-ulp native sqrt; __builtin_sqrtf is IEEE under
// -fhip-fp32-correctly-rounded-divide-sqrt (set explicitly in the build flags).
__device__ __forceinline__ float fsqrt(float a) { return __builtin_sqrtf(a); }

// jnp.mod(t, L) for L > 0 (lax.rem == fmod, then the sign fix-up).  fmod is
// exact; for |t| < 2L it reduces to a compare and one exact subtraction.
__device__ __forceinline__ float mod_floor(float t, float L) {
  float a = fabsf(t);
  float m;
  if (a < L) {
    m = t;
  } else if (a < fmul(2.0f, L)) {
    m = copysignf(fsub(a, L), t);  // exact by Sterbenz
  } else {
    m = fmodf(t, L);
  }
  if (m < 0.0f) m = fadd(m, L);  // (m != 0) & ((m < 0) != (L < 0)), L > 0
  return m;
}

// chiron/neighbors.py:79: mod(r + L/2, L) - L/2
// Same value, bit for bit, for every input; the common case -L < t < 2L (always true for
// wrapped coordinates) takes two compares and one add/sub instead of the general fmod path.
__device__ __forceinline__ float min_image(float d, float L, float hL) {
  const float t = fadd(d, hL);
  float m;
  if (__builtin_expect(!(t > -L && t < fmul(2.0f, L)), 0)) {
    m = mod_floor(t, L);
  } else {
    m = t;
    if (t >= L) m = fsub(t, L);  // fmod(t, L), exact by Sterbenz
    if (t < 0.0f) m = fadd(t, L);  // fmod(t, L) = t, then the sign fix-up of jnp.mod
  }
  return fsub(m, hL);
}

// displacement r = a - b under the space's boundary conditions
__device__ __forceinline__ void displacement(const Box3& box, float ax, float ay, float az, float bx,
                                             float by, float bz, float& rx, float& ry, float& rz) {
  rx = fsub(ax, bx);
  ry = fsub(ay, by);
  rz = fsub(az, bz);
  if (box.periodic) {
    rx = min_image(rx, box.lx, box.hx);
    ry = min_image(ry, box.ly, box.hy);
    rz = min_image(rz, box.lz, box.hz);
  }
}

// r = (forward ? a - b : b - a) under the boundary conditions.  b - a is exactly -(a - b) in
// IEEE arithmetic, so one subtraction plus a sign flip serves both orientations and the
// (not antisymmetric) minimum-image rounding is applied once, to the correctly oriented value.
__device__ __forceinline__ void displacement_oriented(const Box3& box, bool forward, float ax, float ay,
                                                      float az, float bx, float by, float bz, float& rx,
                                                      float& ry, float& rz) {
  const uint32_t flip = forward ? 0u : 0x80000000u;
  rx = __uint_as_float(__float_as_uint(fsub(ax, bx)) ^ flip);
  ry = __uint_as_float(__float_as_uint(fsub(ay, by)) ^ flip);
  rz = __uint_as_float(__float_as_uint(fsub(az, bz)) ^ flip);
  if (box.periodic) {
    rx = min_image(rx, box.lx, box.hx);
    ry = min_image(ry, box.ly, box.hy);
    rz = min_image(rz, box.lz, box.hz);
  }
}

// Branch-free minimum image for |d| < L (same value, bit for bit, as min_image()).  With
// t = d + L/2 in (-L/2, 3L/2) jnp.mod picks, of {t - L, t, t + L}, the one in [0, L]: exactly
// the SMALLEST NON-NEGATIVE of the three, i.e. their minimum as unsigned bit patterns
// (negative floats have the top bit set).  t - L is exact (Sterbenz), t + L rounds once.
__device__ __forceinline__ float min_image_fast(float d, float L, float hL) {
  const float t = fadd(d, hL);
  const uint32_t a = __float_as_uint(fsub(t, L)), b = __float_as_uint(t), c = __float_as_uint(fadd(t, L));
  return fsub(__uint_as_float(min(min(a, b), c)), hL);
}

// displacement_oriented() for the common case; `bad` is raised when some |component| >= L
// (unwrapped coordinates) and the caller must redo the pair through displacement_oriented().
__device__ __forceinline__ void displacement_oriented_fast(const Box3& box, bool forward, float ax, float ay,
                                                           float az, float bx, float by, float bz,
                                                           float& rx, float& ry, float& rz, bool& bad) {
  const uint32_t flip = forward ? 0u : 0x80000000u;
  rx = __uint_as_float(__float_as_uint(fsub(ax, bx)) ^ flip);
  ry = __uint_as_float(__float_as_uint(fsub(ay, by)) ^ flip);
  rz = __uint_as_float(__float_as_uint(fsub(az, bz)) ^ flip);
  if (box.periodic) {
    bad = !(fabsf(rx) < box.lx) | !(fabsf(ry) < box.ly) | !(fabsf(rz) < box.lz);
    rx = min_image_fast(rx, box.lx, box.hx);
    ry = min_image_fast(ry, box.ly, box.hy);
    rz = min_image_fast(rz, box.lz, box.hz);
  } else {
    bad = false;
  }
}

// jnp.linalg.norm squared: (x*x + y*y) + z*z
__device__ __forceinline__ float norm2(float rx, float ry, float rz) {
  return fadd(fadd(fmul(rx, rx), fmul(ry, ry)), fmul(rz, rz));
}

// chiron/neighbors.py:110: x - floor(x / L) * L
__device__ __forceinline__ float wrap1(float x, float L) {
  return fsub(x, fmul(floorf(fdiv(x, L)), L));
}

// ---------------------------------------------------------------------------
// wavefront helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (CHX_WAVE - 1); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, CHX_WAVE);
  return v;
}

// block-wide sum of one double per thread; result valid in thread 0
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double red[BLOCK / CHX_WAVE];
  v = wave_sum(v);
  if (lane_id() == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / CHX_WAVE; ++w) t += red[w];
  }
  return t;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Random123) and the normal draw shared by every stochastic kernel
// ---------------------------------------------------------------------------
#define CHX_STREAM_VELOCITY 0u
#define CHX_STREAM_LANGEVIN 1u
#define CHX_STREAM_MC_DISPLACE 2u
#define CHX_STREAM_UNIFORM 3u
#define CHX_STREAM_SPLIT 0x5EEDu  // key splitting (chiron_amd/rng.py: split)

__device__ __forceinline__ uint4 philox4x32(uint4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ float u32_to_uniform(uint32_t x) {
  // 23 random bits: (x>>9) + 0.5 and the 2^-23 scale are both exact in fp32
  return fmul(fadd((float)(x >> 9), 0.5f), 1.0f / 8388608.0f);
}

// three N(0,1) for (index, step) of `stream`: Box-Muller on one Philox block.
// (z0, z1) = r01 (cos, sin)(2 pi u1), z2 = r23 cos(2 pi u3), r = sqrt(-2 ln u).
// The hardware log2 / sin / cos (about 1e-6 absolute) are ample for a noise source.
__device__ __forceinline__ void philox_normals3(uint32_t index, uint32_t step, uint32_t stream,
                                                uint32_t k0, uint32_t k1, float& z0, float& z1,
                                                float& z2) {
  uint4 o = philox4x32(make_uint4(index, step, 0u, stream), k0, k1);
  float u0 = u32_to_uniform(o.x), u1 = u32_to_uniform(o.y);
  float u2 = u32_to_uniform(o.z), u3 = u32_to_uniform(o.w);
  const float two_pi = 6.283185307179586f;
  float r01 = fsqrt(fmul(-2.0f, __logf(u0)));
  float r23 = fsqrt(fmul(-2.0f, __logf(u2)));
  float t1 = fmul(two_pi, u1), t3 = fmul(two_pi, u3);
  z0 = fmul(r01, __cosf(t1));
  z1 = fmul(r01, __sinf(t1));
  z2 = fmul(r23, __cosf(t3));
}

// component c (0..2) of the same draw, for kernels that give each lane one component
__device__ __forceinline__ float philox_normal_c(uint32_t index, uint32_t step, uint32_t stream,
                                                 uint32_t k0, uint32_t k1, int c) {
  uint4 o = philox4x32(make_uint4(index, step, 0u, stream), k0, k1);
  const float ua = u32_to_uniform(c == 2 ? o.z : o.x);
  const float ub = u32_to_uniform(c == 2 ? o.w : o.y);
  const float r = fsqrt(fmul(-2.0f, __logf(ua)));
  const float t = fmul(6.283185307179586f, ub);
  return fmul(r, c == 1 ? __sinf(t) : __cosf(t));
}

// LJ pair terms from r2 (within cutoff): e = 4 eps (s^12 - s^6), fs = 24 eps (2 s^12 - s^6) / r2
__device__ __forceinline__ void lj_pair(float r2, float sig2, float eps4, float eps24, float& e,
                                        float& fs) {
  // energies/forces are compared with a tolerance (not bit-for-bit), so the 1-ulp
  // hardware reciprocal and explicit FMAs are fine here
  float inv = __builtin_amdgcn_rcpf(r2);
  float s2 = sig2 * inv;
  float s6 = s2 * s2 * s2;
  e = (eps4 * s6) * (s6 - 1.0f);
  fs = (eps24 * s6) * __builtin_fmaf(2.0f, s6, -1.0f) * inv;
}
