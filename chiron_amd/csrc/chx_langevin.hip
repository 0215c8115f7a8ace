// libchx: BAOAB Langevin integrator (chiron/integrators.py:126-195), velocity
// initialisation (chiron/utils.py:116-144), Monte-Carlo proposal / Metropolis
// kernels (chiron/mcmc.py:531-548, :724-755).
//
// One MD step = two launches: a fused per-particle integrate kernel
//   [B (trailing, previous step)] B A O A wrap skin-check
// and the pair-force kernel.  The arithmetic keeps the reference's operation
// order and one rounding per operation, so with injected noise a trajectory
// matches the fp32 oracle up to the force summation order.
#include <cstdlib>

#include "chx_common.h"
#include "chx_internal.h"

// ---------------------------------------------------------------------------
// velocities
// ---------------------------------------------------------------------------
__global__ void k_init_velocities(float* __restrict__ v, const float* __restrict__ mass, int n,
                                  float kT, uint32_t k0, uint32_t k1,
                                  const float* __restrict__ normals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float z0, z1, z2;
  if (normals) {
    z0 = normals[3 * (size_t)i];
    z1 = normals[3 * (size_t)i + 1];
    z2 = normals[3 * (size_t)i + 2];
  } else {
    philox_normals3((uint32_t)i, 0u, CHX_STREAM_VELOCITY, k0, k1, z0, z1, z2);
  }
  const float sv = fsqrt(fdiv(kT, mass[i]));  // chiron/utils.py:140
  v[3 * (size_t)i] = fmul(sv, z0);
  v[3 * (size_t)i + 1] = fmul(sv, z1);
  v[3 * (size_t)i + 2] = fmul(sv, z2);
}

extern "C" int chx_init_velocities(float* v_out, const float* mass, int32_t n, float kT,
                                   uint32_t k0, uint32_t k1, const float* normals,
                                   chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(v_out && mass, "null pointer");
  hipLaunchKernelGGL(k_init_velocities, dim3((n + 255) / 256), dim3(256), 0, chx_stream(stream),
                     v_out, mass, n, kT, k0, k1, normals);
  return chx_check_launch("chx_init_velocities");
}

__global__ void k_kick(float* __restrict__ v, const float* __restrict__ f,
                       const float* __restrict__ mass, int n, float hdt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = mass[i];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t a = 3 * (size_t)i + c;
    v[a] = fadd(v[a], fdiv(fmul(hdt, f[a]), m));  // chiron/integrators.py:195
  }
}

extern "C" int chx_kick(float* v, const float* f, const float* mass, int32_t n, float half_dt,
                        chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(v && f && mass, "null pointer");
  hipLaunchKernelGGL(k_kick, dim3((n + 255) / 256), dim3(256), 0, chx_stream(stream), v, f, mass, n,
                     half_dt);
  return chx_check_launch("chx_kick");
}

// ---------------------------------------------------------------------------
// fused integrate kernel
// ---------------------------------------------------------------------------
struct BaoabArgs {
  int n;
  float4* x4;
  float* v;
  const float* f;
  const float4* ref4;
  const float* mass;
  const uint32_t* perm;
  Box3 box;
  int wrap;
  float T_half_skin;
  float hdt, a, b, kT;
  uint32_t k0, k1;
  const float* noise;
  long long noise_step0;
  int32_t* ctrl;
};

__global__ __launch_bounds__(256) void k_baoab(BaoabArgs p, long long step, int n_kicks) {
  if (p.ctrl[1] != 0) return;  // halted by an earlier step of this chunk
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= p.n) return;
  float4 x = p.x4[s];
  float vx = p.v[3 * (size_t)s], vy = p.v[3 * (size_t)s + 1], vz = p.v[3 * (size_t)s + 2];
  const float fx = p.f[3 * (size_t)s], fy = p.f[3 * (size_t)s + 1], fz = p.f[3 * (size_t)s + 2];
  const float m = p.mass[s];
  const float hdt = p.hdt;
  // B: v += (dt/2) F / m            chiron/integrators.py:181 (and :195 of the previous step)
  for (int k = 0; k < n_kicks; ++k) {
    vx = fadd(vx, fdiv(fmul(hdt, fx), m));
    vy = fadd(vy, fdiv(fmul(hdt, fy), m));
    vz = fadd(vz, fdiv(fmul(hdt, fz), m));
  }
  // A: x += (dt/2) v                :183
  x.x = fadd(x.x, fmul(hdt, vx));
  x.y = fadd(x.y, fmul(hdt, vy));
  x.z = fadd(x.z, fmul(hdt, vz));
  // O: v = a v + b sigma_v xi       :185-186
  const uint32_t o = p.perm ? p.perm[s] : (uint32_t)s;
  float z0, z1, z2;
  if (p.noise) {
    const size_t row = ((size_t)(step - p.noise_step0) * (size_t)p.n + o) * 3;
    z0 = p.noise[row];
    z1 = p.noise[row + 1];
    z2 = p.noise[row + 2];
  } else {
    philox_normals3(o, (uint32_t)step, CHX_STREAM_LANGEVIN, p.k0, p.k1, z0, z1, z2);
  }
  const float bs = fmul(p.b, fsqrt(fdiv(p.kT, m)));  // b * sigma_v, sigma_v = sqrt(kT/m) (:131)
  vx = fadd(fmul(p.a, vx), fmul(bs, z0));
  vy = fadd(fmul(p.a, vy), fmul(bs, z1));
  vz = fadd(fmul(p.a, vz), fmul(bs, z2));
  // A                               :188
  x.x = fadd(x.x, fmul(hdt, vx));
  x.y = fadd(x.y, fmul(hdt, vy));
  x.z = fadd(x.z, fmul(hdt, vz));
  // wrap                            :239
  if (p.wrap && p.box.periodic) {
    x.x = wrap1(x.x, p.box.lx);
    x.y = wrap1(x.y, p.box.ly);
    x.z = wrap1(x.z, p.box.lz);
  }
  p.x4[s] = x;
  p.v[3 * (size_t)s] = vx;
  p.v[3 * (size_t)s + 1] = vy;
  p.v[3 * (size_t)s + 2] = vz;
  // skin check                      :241, chiron/neighbors.py:864-868
  if (p.ref4) {
    const float4 r = p.ref4[s];
    float rx, ry, rz;
    displacement(p.box, x.x, x.y, x.z, r.x, r.y, r.z, rx, ry, rz);
    if (norm2(rx, ry, rz) > p.T_half_skin) {
      p.ctrl[2] = (int32_t)step;
      p.ctrl[0] = 1;
    }
  }
}

// One BAOAB update on plain user-order (n,3) arrays: the body of the reference's loop
// (chiron/integrators.py:181-188, wrap :239, skin check :241) for callers that drive the loop
// themselves and evaluate the force in between.
__global__ __launch_bounds__(256) void k_baoab3(float* __restrict__ x, float* __restrict__ v,
                                                const float* __restrict__ f, const float* __restrict__ mass,
                                                int n, float hdt, float a, float b, float kT, Box3 box,
                                                int wrap, const float* __restrict__ ref, float T_half_skin,
                                                uint32_t k0, uint32_t k1, uint32_t step,
                                                const float* __restrict__ noise, int32_t* __restrict__ flag) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const float m = mass[s];
  float z[3];
  if (noise) {
    z[0] = noise[3 * (size_t)s], z[1] = noise[3 * (size_t)s + 1], z[2] = noise[3 * (size_t)s + 2];
  } else {
    philox_normals3((uint32_t)s, step, CHX_STREAM_LANGEVIN, k0, k1, z[0], z[1], z[2]);
  }
  const float bs = fmul(b, fsqrt(fdiv(kT, m)));
  const float L[3] = {box.lx, box.ly, box.lz};
  float xn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float vc = fadd(v[3 * (size_t)s + c], fdiv(fmul(hdt, f[3 * (size_t)s + c]), m));  // B  :181
    float xc = fadd(x[3 * (size_t)s + c], fmul(hdt, vc));                              // A  :183
    vc = fadd(fmul(a, vc), fmul(bs, z[c]));                                           // O  :186
    xc = fadd(xc, fmul(hdt, vc));                                                      // A  :188
    if (wrap && box.periodic) xc = wrap1(xc, L[c]);                                   // :239
    x[3 * (size_t)s + c] = xc;
    v[3 * (size_t)s + c] = vc;
    xn[c] = xc;
  }
  if (ref && flag) {
    float rx, ry, rz;
    displacement(box, xn[0], xn[1], xn[2], ref[3 * (size_t)s], ref[3 * (size_t)s + 1], ref[3 * (size_t)s + 2],
                 rx, ry, rz);
    if (norm2(rx, ry, rz) > T_half_skin) flag[0] = 1;  // :241, chiron/neighbors.py:868
  }
}

extern "C" int chx_baoab_step(float* x, float* v, const float* f, const float* mass, int32_t n,
                              chx_box_t box, int32_t wrap, float half_dt, float a, float b, float kT,
                              const float* ref_xyz, float skin, uint32_t k0, uint32_t k1, uint32_t step,
                              const float* noise, int32_t* rebuild_flag_out, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  hipStream_t st = chx_stream(stream);
  if (rebuild_flag_out && hipMemsetAsync(rebuild_flag_out, 0, sizeof(int32_t), st) != hipSuccess) {
    chx_set_error("chx_baoab_step: hipMemsetAsync failed");
    return CHX_ERR_LAUNCH;
  }
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(x && v && f && mass, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  hipLaunchKernelGGL(k_baoab3, dim3((n + 255) / 256), dim3(256), 0, st, x, v, f, mass, n, half_dt, a, b, kT,
                     make_box3(box), wrap, ref_xyz, chx_cutoff_threshold_h(skin * 0.5f), k0, k1, step, noise,
                     rebuild_flag_out);
  return chx_check_launch("chx_baoab_step");
}

static int validate_langevin(const chx_langevin_t* p) {
  CHX_REQUIRE(p, "null params");
  CHX_REQUIRE(p->n >= 1, "n < 1");
  CHX_REQUIRE(p->x4 && p->v && p->f && p->mass && p->ctrl, "null pointer");
  CHX_REQUIRE(!p->fl || p->deg, "force list without degrees");
  CHX_REQUIRE(!p->box.periodic || (p->box.lx > 0 && p->box.ly > 0 && p->box.lz > 0),
              "box lengths must be > 0");
  CHX_REQUIRE(!(p->fl) || (p->g >= 1 && p->fl_chunks >= 1), "bad g / fl_chunks");
  return CHX_OK;
}

static int launch_force(const chx_langevin_t* p, int32_t* ctrl, hipStream_t st) {
  if (p->external_force) return CHX_OK;
  if (p->fl) {
    return chx_launch_lj_fl(p->x4, p->fl, p->deg, p->n, p->g, p->fl_chunks, p->box, p->sigma, p->epsilon,
                            p->cutoff, p->f, nullptr, p->ws, ctrl, st);
  }
  // no neighbor list: reference semantics are the no-PBC all-pairs branch when no list was
  // given (box.periodic == 0) and the PBC pair list otherwise
  return chx_launch_lj_allpairs(p->x4, 4, p->n, p->box, p->sigma, p->epsilon, p->cutoff, p->f, nullptr,
                                p->ws, ctrl, st);
}

extern "C" int chx_langevin_chunk(const chx_langevin_t* p, int64_t step0, int32_t n_steps,
                                  int32_t kicks_first, chx_stream_t stream) {
  int rc = validate_langevin(p);
  if (rc != CHX_OK) return rc;
  CHX_REQUIRE(n_steps >= 0, "n_steps < 0");
  CHX_REQUIRE(kicks_first >= 0 && kicks_first <= 2, "kicks_first must be 0, 1 or 2");
  CHX_REQUIRE(step0 >= 0 && step0 + n_steps < (1ll << 31), "step index out of range");
  hipStream_t st = chx_stream(stream);
  BaoabArgs a;
  a.n = p->n;
  a.x4 = reinterpret_cast<float4*>(p->x4);
  a.v = p->v;
  a.f = p->f;
  a.ref4 = reinterpret_cast<const float4*>(p->ref4);
  a.mass = p->mass;
  a.perm = p->perm;
  a.box = make_box3(p->box);
  a.wrap = p->wrap;
  a.T_half_skin = chx_cutoff_threshold_h(p->skin * 0.5f);
  a.hdt = p->half_dt;
  a.a = p->a;
  a.b = p->b;
  a.kT = p->kT;
  a.k0 = p->k0;
  a.k1 = p->k1;
  a.noise = p->noise;
  a.noise_step0 = p->noise_step0;
  a.ctrl = p->ctrl;
  const unsigned blocks = (unsigned)((p->n + 255) / 256);
  for (int s = 0; s < n_steps; ++s) {
    hipLaunchKernelGGL(k_baoab, dim3(blocks), dim3(256), 0, st, a, (long long)(step0 + s),
                       s == 0 ? kicks_first : 2);
    rc = launch_force(p, p->ctrl, st);
    if (rc != CHX_OK) return rc;
  }
  return chx_check_launch("chx_langevin_chunk");
}

// ABI 2: the fused step kernel takes the particle's mass from ref4[p].w (chx_pack_langevin puts it there).  A ref4
// produced any other way (chx_permute_pack pads with 0) would give m = 0 and inf / NaN velocities without any error;
// with CHX_DEBUG_SYNC set the entry points look at the first row before they launch.
static int debug_check_ref4(const chx_langevin_t* p, hipStream_t st, const char* who) {
  static const bool dbg = getenv("CHX_DEBUG_SYNC") != nullptr;
  if (!dbg || !p->fl || !p->ref4) return CHX_OK;
  float row[4] = {0.f, 0.f, 0.f, 0.f};
  chx_aql_join(st);
  if (hipStreamSynchronize(st) != hipSuccess ||
      hipMemcpy(row, p->ref4, sizeof(row), hipMemcpyDeviceToHost) != hipSuccess) {
    chx_set_error("%s: cannot read ref4[0] (debug check)", who);
    return CHX_ERR_LAUNCH;
  }
  if (!(row[3] > 0.0f)) {
    chx_set_error("%s: ref4[0].w = %g is not a mass: ref4 must come from chx_pack_langevin (ABI 2)", who, (double)row[3]);
    return CHX_ERR_INVALID;
  }
  return CHX_OK;
}

extern "C" int chx_langevin_fused_chunk(const chx_langevin_t* p, float* x_even, float* x_odd,
                                        int64_t step0, int32_t n_steps, int32_t kicks_first,
                                        int32_t do_aoa, chx_stream_t stream) {
  CHX_REQUIRE(p, "null params");
  CHX_REQUIRE(p->n >= 1, "n < 1");
  CHX_REQUIRE(x_even && x_odd && x_even != x_odd, "need two distinct position buffers");
  CHX_REQUIRE(p->v && p->mass && p->ctrl, "null pointer");
  CHX_REQUIRE(!p->fl || (p->deg && p->fl_chunks >= 1), "force list without degrees / chunks");
  CHX_REQUIRE(!p->box.periodic || (p->box.lx > 0 && p->box.ly > 0 && p->box.lz > 0),
              "box lengths must be > 0");
  CHX_REQUIRE(n_steps >= 0, "n_steps < 0");
  CHX_REQUIRE(kicks_first >= 0 && kicks_first <= 2, "kicks_first must be 0, 1 or 2");
  CHX_REQUIRE(step0 >= 0 && step0 + n_steps < (1ll << 31) - 1, "step index out of range");
  CHX_REQUIRE(do_aoa || n_steps <= 1, "force-only mode takes a single launch");
  int rc = debug_check_ref4(p, chx_stream_steps(stream), "chx_langevin_fused_chunk");
  if (rc != CHX_OK) return rc;
  rc = chx_launch_lj_steps(p, x_even, x_odd, step0, n_steps, kicks_first, do_aoa,
                               chx_stream_steps(stream));
  if (rc != CHX_OK) return rc;
  return chx_check_launch("chx_langevin_fused_chunk");
}

extern "C" int chx_langevin_fused_run(const chx_langevin_t* p, float* x_even, float* x_odd, int64_t step0,
                                      int32_t n_steps, int32_t kicks_first, int32_t window,
                                      int32_t* n_enqueued_out, chx_stream_t stream) {
  CHX_REQUIRE(p && n_enqueued_out, "null pointer");
  *n_enqueued_out = 0;
  CHX_REQUIRE(p->n >= 1, "n < 1");
  CHX_REQUIRE(x_even && x_odd && x_even != x_odd, "need two distinct position buffers");
  CHX_REQUIRE(p->v && p->mass && p->ctrl, "null pointer");
  CHX_REQUIRE(p->fl && p->deg && p->fl_chunks >= 1, "the launch window needs the force-list path");
  CHX_REQUIRE(p->host_ctrl, "the launch window polls host_ctrl");
  CHX_REQUIRE(window >= 1, "window < 1");
  CHX_REQUIRE(!p->box.periodic || (p->box.lx > 0 && p->box.ly > 0 && p->box.lz > 0),
              "box lengths must be > 0");
  CHX_REQUIRE(n_steps >= 0, "n_steps < 0");
  CHX_REQUIRE(kicks_first >= 0 && kicks_first <= 2, "kicks_first must be 0, 1 or 2");
  CHX_REQUIRE(step0 >= 0 && step0 + n_steps < (1ll << 31) - 1, "step index out of range");
  int rc = debug_check_ref4(p, chx_stream_steps(stream), "chx_langevin_fused_run");
  if (rc != CHX_OK) return rc;
  CHX_REQUIRE(!p->snap4 || (p->ref4 && p->skin > 0.f), "snap4 (decoupled force list) needs ref4 and a skin");
  CHX_REQUIRE(!p->ev_log || p->ev_log_cap > 0, "ev_log without a capacity");
  rc = chx_launch_lj_steps(p, x_even, x_odd, step0, n_steps, kicks_first, 1, chx_stream_steps(stream), window,
                           n_enqueued_out);
  if (rc != CHX_OK) return rc;
  return chx_check_launch("chx_langevin_fused_run");
}

static int validate_batch(chx_batch_item_t* items, int32_t n_items, bool kick, const char* who) {
  CHX_REQUIRE(items && n_items >= 1, "no items");
  for (int r = 0; r < n_items; ++r) {
    const chx_batch_item_t& it = items[r];
    const chx_langevin_t* p = it.p;
    CHX_REQUIRE(p, "null params");
    CHX_REQUIRE(p->n >= 1, "n < 1");
    CHX_REQUIRE(it.x_even && it.x_odd && it.x_even != it.x_odd, "need two distinct position buffers");
    CHX_REQUIRE(p->v && p->mass && p->ctrl, "null pointer");
    CHX_REQUIRE(p->fl && p->deg && p->fl_chunks >= 1, "the batched launch needs the force-list path");
    CHX_REQUIRE(p->host_ctrl, "the batched launch publishes to host_ctrl");
    CHX_REQUIRE(!p->snap4, "the batched launch does not take a decoupled force list (snap4)");
    CHX_REQUIRE(p->g == 1 || p->g == 2 || p->g == 4 || p->g == 8, "lanes per particle must be 1, 2, 4 or 8");
    CHX_REQUIRE(!p->box.periodic || (p->box.lx > 0 && p->box.ly > 0 && p->box.lz > 0), "box lengths must be > 0");
    CHX_REQUIRE(kick || it.n_steps >= 0, "n_steps < 0");
    CHX_REQUIRE(it.kicks_first >= 0 && it.kicks_first <= 2, "kicks_first must be 0, 1 or 2");
    CHX_REQUIRE(it.step0 >= 0 && it.step0 + (kick ? 1 : it.n_steps) < (1ll << 31) - 1, "step index out of range");
    for (int q = 0; q < r; ++q) CHX_REQUIRE(items[q].p->ctrl != p->ctrl, "two items share a control block");
  }
  (void)who;
  return CHX_OK;
}

extern "C" int64_t chx_sizeof_batch_item_t(void) { return (int64_t)sizeof(chx_batch_item_t); }

extern "C" int chx_langevin_batch_run(chx_batch_item_t* items, int32_t n_items, int32_t window, chx_stream_t stream) {
  int rc = validate_batch(items, n_items, false, "chx_langevin_batch_run");
  if (rc != CHX_OK) return rc;
  CHX_REQUIRE(window >= 1, "window < 1");
  rc = chx_launch_lj_steps_batch(items, n_items, 1, window, chx_stream_steps(stream));
  if (rc != CHX_OK) return rc;
  return chx_check_launch("chx_langevin_batch_run");
}

extern "C" int chx_langevin_batch_kick(chx_batch_item_t* items, int32_t n_items, chx_stream_t stream) {
  int rc = validate_batch(items, n_items, true, "chx_langevin_batch_kick");
  if (rc != CHX_OK) return rc;
  for (int r = 0; r < n_items; ++r) items[r].n_steps = 1;
  rc = chx_launch_lj_steps_batch(items, n_items, 0, 0, chx_stream_steps(stream));
  if (rc != CHX_OK) return rc;
  return chx_check_launch("chx_langevin_batch_kick");
}

extern "C" int chx_langevin_resume(const chx_langevin_t* p, chx_stream_t stream) {
  int rc = validate_langevin(p);
  if (rc != CHX_OK) return rc;
  rc = launch_force(p, nullptr, chx_stream(stream));
  if (rc != CHX_OK) return rc;
  return chx_check_launch("chx_langevin_resume");
}

// ---------------------------------------------------------------------------
// Monte Carlo
// ---------------------------------------------------------------------------
__global__ void k_mc_displace(const float* __restrict__ x, float* __restrict__ xo,
                              const float* __restrict__ subset, int n, float sigma_d, Box3 box,
                              uint32_t k0, uint32_t k1, uint32_t ctr,
                              const float* __restrict__ normals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float z0, z1, z2;
  if (normals) {
    z0 = normals[3 * (size_t)i];
    z1 = normals[3 * (size_t)i + 1];
    z2 = normals[3 * (size_t)i + 2];
  } else {
    philox_normals3((uint32_t)i, ctr, CHX_STREAM_MC_DISPLACE, k0, k1, z0, z1, z2);
  }
  // normal * sigma [* mask] then x + d   (chiron/mcmc.py:733-748)
  float dx = fmul(z0, sigma_d), dy = fmul(z1, sigma_d), dz = fmul(z2, sigma_d);
  if (subset) {
    const float mk = subset[i];
    dx = fmul(dx, mk);
    dy = fmul(dy, mk);
    dz = fmul(dz, mk);
  }
  float a = fadd(x[3 * (size_t)i], dx), b = fadd(x[3 * (size_t)i + 1], dy),
        c = fadd(x[3 * (size_t)i + 2], dz);
  if (box.periodic) {  // wrap only when a neighbor list is in play (:751-755)
    a = wrap1(a, box.lx);
    b = wrap1(b, box.ly);
    c = wrap1(c, box.lz);
  }
  xo[3 * (size_t)i] = a;
  xo[3 * (size_t)i + 1] = b;
  xo[3 * (size_t)i + 2] = c;
}

extern "C" int chx_mc_displace(const float* x, float* x_out, const float* subset_mask, int32_t n,
                               float sigma_d, chx_box_t box, uint32_t k0, uint32_t k1, uint32_t ctr,
                               const float* normals, chx_stream_t stream) {
  CHX_REQUIRE(n >= 0, "n < 0");
  if (n == 0) return CHX_OK;
  CHX_REQUIRE(x && x_out, "null pointer");
  CHX_REQUIRE(!box.periodic || (box.lx > 0 && box.ly > 0 && box.lz > 0), "box lengths must be > 0");
  hipLaunchKernelGGL(k_mc_displace, dim3((n + 255) / 256), dim3(256), 0, chx_stream(stream), x, x_out,
                     subset_mask, n, sigma_d, make_box3(box), k0, k1, ctr, normals);
  return chx_check_launch("chx_mc_displace");
}

__global__ void k_metropolis_decide(const float* __restrict__ log_ratio, float* __restrict__ u_cur,
                                    const float* __restrict__ u_prop, uint32_t k0, uint32_t k1,
                                    uint32_t ctr, float uniform_h, int32_t* __restrict__ accepted,
                                    int32_t* __restrict__ counters) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const float lp = log_ratio[0];
  const float up = u_prop ? u_prop[0] : 0.0f;
  float U = uniform_h;
  if (!(U >= 0.0f)) U = u32_to_uniform(philox4x32(make_uint4(0u, ctr, 0u, CHX_STREAM_UNIFORM), k0, k1).x);
  int acc = 0;
  if (!isnan(up) && !isnan(lp)) {             // chiron/mcmc.py:428-429
    acc = (-lp <= 0.0f || U < expf(lp)) ? 1 : 0;  // :544-545
  }
  accepted[0] = acc;
  if (counters) {
    counters[0] += acc;
    counters[1] += 1;
  }
  if (acc && u_cur && u_prop) u_cur[0] = up;
}

__global__ void k_select(float* __restrict__ cur, const float* __restrict__ prop, int64_t count,
                         const int32_t* __restrict__ accepted) {
  if (accepted[0] == 0) return;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) cur[i] = prop[i];
}

extern "C" int chx_metropolis_select(float* x_cur, const float* x_prop, int64_t count,
                                     const float* log_ratio, float* u_cur, const float* u_prop,
                                     uint32_t k0, uint32_t k1, uint32_t ctr, float uniform_h,
                                     int32_t* accepted_out, int32_t* counters, chx_stream_t stream) {
  CHX_REQUIRE(count >= 0, "count < 0");
  CHX_REQUIRE(log_ratio && accepted_out, "null pointer");
  hipStream_t st = chx_stream(stream);
  hipLaunchKernelGGL(k_metropolis_decide, dim3(1), dim3(64), 0, st, log_ratio, u_cur, u_prop, k0, k1,
                     ctr, uniform_h, accepted_out, counters);
  if (count > 0) {
    CHX_REQUIRE(x_cur && x_prop, "null pointer");
    hipLaunchKernelGGL(k_select, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, x_cur, x_prop,
                       count, accepted_out);
  }
  return chx_check_launch("chx_metropolis_select");
}
