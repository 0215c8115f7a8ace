/*
 * CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.
 *
 * Scalar fp32 port of the reference algorithm, used (a) by tests to cross-check
 * the NumPy oracle at sizes NumPy cannot reach and (b) by bench.py as the
 * `cpu_baseline` ("port", 1 thread).  It deliberately keeps the reference's
 * algorithmic shape: O(N^2) neighbor build (chiron/neighbors.py:548-626,
 * :694-729), distances over the whole PADDED (N,M) list every step
 * (chiron/neighbors.py:731-826), masked LJ energy (chiron/potential.py:193-213),
 * closed-form pair forces scattered to i and j (chiron/potential.py:322-332),
 * un-fused BAOAB (chiron/integrators.py:179-195).  Build with
 * -O2 -ffp-contract=off so every operation rounds once, like NumPy/XLA fp32.
 * Parity pinning: see oracle/chiron_oracle.py (this file is checked against it).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int periodic; float lx, ly, lz; } orc_box;

/* identity of the text this binary was compiled from (oracle/build_oracle.py bakes it in and refuses a binary whose
 * hash differs from oracle.c beside it: a stale checker must not pass for a fresh one) */
#ifndef ORC_SOURCE_HASH
#define ORC_SOURCE_HASH "unhashed"
#endif
const char* orc_source_hash(void) { return "ORC_SOURCE_HASH=" ORC_SOURCE_HASH; }

static inline float mod_floor(float t, float L) { /* jnp.mod, neighbors.py:79 */
  float m = fmodf(t, L);
  if (m != 0.0f && ((m < 0.0f) != (L < 0.0f))) m += L;
  return m;
}
static inline float minimg(float d, float L) { return mod_floor(d + L * 0.5f, L) - L * 0.5f; }

static inline float disp(const orc_box* b, const float* a, const float* c, float* r) {
  r[0] = a[0] - c[0]; r[1] = a[1] - c[1]; r[2] = a[2] - c[2];
  if (b->periodic) { r[0] = minimg(r[0], b->lx); r[1] = minimg(r[1], b->ly); r[2] = minimg(r[2], b->lz); }
  return sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
}

/* one fixed-width pass of _build_neighborlist; returns max n_neighbors */
int orc_nl_build_fixed(const float* x, int n, int m, orc_box box, float cs, uint32_t* nl,
                       int32_t* mask, int32_t* nnbr) {
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    int cnt = 0; uint32_t first = 0; float r[3];
    for (int j = i + 1; j < n; ++j) {
      if (disp(&box, x + 3 * i, x + 3 * j, r) < cs) {
        if (cnt == 0) first = (uint32_t)j;
        if (cnt < m) nl[(size_t)i * m + cnt] = (uint32_t)j;
        ++cnt;
      }
    }
    uint32_t fill = cnt ? first : 0u;
    if (fill == (uint32_t)i) fill += 1u;
    for (int k = cnt < m ? cnt : m; k < m; ++k) nl[(size_t)i * m + k] = fill;
    for (int k = 0; k < m; ++k) mask[(size_t)i * m + k] = k < cnt;
    nnbr[i] = cnt;
    if (cnt > mx) mx = cnt;
  }
  return mx;
}

/* padded-list distances + masked energy + scattered pair forces (what calculate + -grad do) */
float orc_lj_nl(const float* x, const uint32_t* nl, const int32_t* mask, int n, int m, orc_box box,
                float sigma, float eps, float rc, float* force) {
  float e_total = 0.0f;
  if (force) memset(force, 0, sizeof(float) * 3 * (size_t)n);
  const float eps4 = (float)(4.0 * (double)eps);
  for (int i = 0; i < n; ++i) {
    float e_row = 0.0f;
    for (int k = 0; k < m; ++k) {
      uint32_t j = nl[(size_t)i * m + k];
      float r[3];
      float d = disp(&box, x + 3 * i, x + 3 * j, r);
      int mk = (d < rc) && mask[(size_t)i * m + k];
      float s = sigma / d, s2 = s * s, s4 = s2 * s2, s8 = s4 * s4;
      float s6 = s2 * s4, s12 = s4 * s8;
      e_row += (float)mk * (eps4 * (s12 - s6));
      if (mk && force) {
        float fs = 24.0f * (eps / (d * d)) * (2.0f * s12 - s6);
        for (int c = 0; c < 3; ++c) { force[3 * i + c] += fs * r[c]; force[3 * j + c] -= fs * r[c]; }
      }
    }
    e_total += e_row;
  }
  return e_total;
}

int orc_check(const float* x, const float* ref, int n, orc_box box, float skin) {
  float r[3];
  for (int i = 0; i < n; ++i)
    if (disp(&box, x + 3 * i, ref + 3 * i, r) >= skin / 2.0f) return 1;
  return 0;
}

/* ---- Philox4x32-10 normals, identical construction to csrc/chx_common.h ---- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static inline float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }
static void normals3(uint32_t idx, uint32_t step, uint32_t stream, uint32_t k0, uint32_t k1, float* z) {
  uint32_t c[4] = {idx, step, 0u, stream};
  philox(c, k0, k1);
  float r01 = sqrtf(-2.0f * logf(u01(c[0]))), r23 = sqrtf(-2.0f * logf(u01(c[2])));
  float t1 = 6.283185307179586f * u01(c[1]), t3 = 6.283185307179586f * u01(c[3]);
  z[0] = r01 * cosf(t1); z[1] = r01 * sinf(t1); z[2] = r23 * cosf(t3);
}

/*
 * LangevinIntegrator.run with a NeighborListNsqrd (chiron/integrators.py:168-195).
 * noise: optional (n_steps, n, 3); else Philox stream 1 keyed (k0,k1).
 * stats[0] = rebuilds, stats[1] = final n_max, stats[2] = seconds spent in list builds (x1e6).
 */
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int build_with_growth(const float* x, int n, int* m, orc_box box, float cs, uint32_t** nl,
                             int32_t** mask, int32_t* nnbr) {
  for (;;) {
    *nl = (uint32_t*)realloc(*nl, sizeof(uint32_t) * (size_t)n * (*m));
    *mask = (int32_t*)realloc(*mask, sizeof(int32_t) * (size_t)n * (*m));
    if (!*nl || !*mask) return -1;
    int mx = orc_nl_build_fixed(x, n, *m, box, cs, *nl, *mask, nnbr);
    int eq = 0;
    for (int i = 0; i < n; ++i) if (nnbr[i] == *m) { eq = 1; break; }
    if (!eq) return 0;          /* neighbors.py:709: grow only on equality */
    *m = mx + 10;               /* :713 */
  }
}

int orc_langevin_run(float* x, float* v, int n, orc_box box, const float* mass, float sigma, float eps,
                     float rc, float skin, int m0, float dt, float a, float b, float kT, int n_steps,
                     const float* noise, uint32_t k0, uint32_t k1, double* stats) {
  const float hdt = (float)((double)dt * 0.5);
  const float cs = (float)((double)rc + (double)skin);
  int m = m0, rebuilds = 0;
  double t_build = 0.0;
  uint32_t* nl = NULL; int32_t* mask = NULL;
  int32_t* nnbr = (int32_t*)malloc(sizeof(int32_t) * n);
  float* ref = (float*)malloc(sizeof(float) * 3 * n);
  float* F = (float*)malloc(sizeof(float) * 3 * n);
  if (!nnbr || !ref || !F) return -1;
  double t0 = now_s();
  if (build_with_growth(x, n, &m, box, cs, &nl, &mask, nnbr)) return -1;
  t_build += now_s() - t0;
  memcpy(ref, x, sizeof(float) * 3 * n);
  orc_lj_nl(x, nl, mask, n, m, box, sigma, eps, rc, F);
  const float L[3] = {box.lx, box.ly, box.lz};
  for (int s = 0; s < n_steps; ++s) {
    for (int i = 0; i < n; ++i) {
      const float mi = mass[i];
      const float sv = sqrtf(kT / mi);
      float z[3];
      if (noise) memcpy(z, noise + ((size_t)s * n + i) * 3, sizeof(z));
      else normals3((uint32_t)i, (uint32_t)s, 1u, k0, k1, z);
      for (int c = 0; c < 3; ++c) {
        float vv = v[3 * i + c] + hdt * F[3 * i + c] / mi;   /* :181 */
        float xx = x[3 * i + c] + hdt * vv;                  /* :183 */
        vv = (a * vv) + (b * sv * z[c]);                     /* :186 */
        xx = xx + hdt * vv;                                  /* :188 */
        if (box.periodic) xx = xx - floorf(xx / L[c]) * L[c];/* :239 */
        v[3 * i + c] = vv; x[3 * i + c] = xx;
      }
    }
    if (orc_check(x, ref, n, box, skin)) {                   /* :241-242 */
      t0 = now_s();
      if (build_with_growth(x, n, &m, box, cs, &nl, &mask, nnbr)) return -1;
      t_build += now_s() - t0;
      memcpy(ref, x, sizeof(float) * 3 * n);
      ++rebuilds;
    }
    orc_lj_nl(x, nl, mask, n, m, box, sigma, eps, rc, F);    /* :193 */
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c) v[3 * i + c] = v[3 * i + c] + hdt * F[3 * i + c] / mass[i]; /* :195 */
  }
  if (stats) { stats[0] = rebuilds; stats[1] = m; stats[2] = t_build; }
  free(nl); free(mask); free(nnbr); free(ref); free(F);
  return 0;
}

/* ==========================================================================
 * Variant (ii) of SURVEY.md 8(d): the SAME physics as orc_langevin_run, but with an O(N) cell-list
 * build and OpenMP over particles -- the CPU baseline a production code would run (the faithful
 * port above is O(N^2) per build and single-threaded, like the reference's vmapped build).
 *
 * Bit-identical trajectories to orc_langevin_run by construction (tested in tests/test_oracle_golden.py):
 *   - a FULL neighbor row per particle (all partners with d_ref < cutoff + skin, ascending id), so the
 *     force on i is a gather -- no scatter, no atomics, any thread count gives the same bits;
 *   - the faithful port adds the pair terms of particle i in ascending partner id as well (first the
 *     reactions `force[j] -= fs r` from the rows of lower ids, then its own row), and every pair is
 *     evaluated in the reference's orientation r = x_lo - x_hi (chiron/neighbors.py:773-779), so the
 *     summation order and every rounding coincide;
 *   - rebuild when any particle has moved >= skin/2 (chiron/neighbors.py:868), checked every step.
 * stats[0] = rebuilds, [1] = row capacity, [2] = seconds in list builds, [3] = threads used.
 * ========================================================================== */
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { int nc[3]; float inv[3]; int* start; int* items; } orc_grid;

static int cell_of(const orc_grid* g, const float* p, const float* L) {
  int c[3];
  for (int d = 0; d < 3; ++d) {
    float w = p[d] - floorf(p[d] / L[d]) * L[d];
    int k = (int)floorf(w * g->inv[d]);
    c[d] = k < 0 ? 0 : (k >= g->nc[d] ? g->nc[d] - 1 : k);
  }
  return (c[2] * g->nc[1] + c[1]) * g->nc[0] + c[0];
}

/* full rows, ascending partner id; returns max degree (> cap means: grow and call again) */
static int build_full_rows(const float* x, int n, orc_box box, float cs, int cap, uint32_t* rows, int32_t* deg) {
  const float L[3] = {box.lx, box.ly, box.lz};
  orc_grid g;
  int cells = 1, use_grid = box.periodic;
  for (int d = 0; d < 3 && use_grid; ++d) {
    g.nc[d] = (int)floor((double)L[d] / ((double)cs * 1.0001 + 1e-4));
    if (g.nc[d] < 3) use_grid = 0;
    else { if (g.nc[d] > 256) g.nc[d] = 256; g.inv[d] = (float)g.nc[d] / L[d]; cells *= g.nc[d]; }
  }
  int* cid = NULL;
  g.start = NULL; g.items = NULL;
  if (use_grid) {
    cid = (int*)malloc(sizeof(int) * n);
    g.start = (int*)calloc((size_t)cells + 1, sizeof(int));
    g.items = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) { cid[i] = cell_of(&g, x + 3 * i, L); g.start[cid[i] + 1]++; }
    for (int c = 0; c < cells; ++c) g.start[c + 1] += g.start[c];
    int* cur = (int*)malloc(sizeof(int) * cells);
    memcpy(cur, g.start, sizeof(int) * cells);
    for (int i = 0; i < n; ++i) g.items[cur[cid[i]]++] = i;   /* ascending id inside a cell */
    free(cur);
  }
  int mx = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(max : mx)
  for (int i = 0; i < n; ++i) {
    uint32_t* row = rows + (size_t)i * cap;
    int cnt = 0;
    float r[3];
    if (use_grid) {
      const int c = cid[i];
      const int cx = c % g.nc[0], cy = (c / g.nc[0]) % g.nc[1], cz = c / (g.nc[0] * g.nc[1]);
      for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        const int ox = (cx + dx + g.nc[0]) % g.nc[0], oy = (cy + dy + g.nc[1]) % g.nc[1], oz = (cz + dz + g.nc[2]) % g.nc[2];
        const int o = (oz * g.nc[1] + oy) * g.nc[0] + ox;
        for (int t = g.start[o]; t < g.start[o + 1]; ++t) {
          const int j = g.items[t];
          if (j == i) continue;
          const int lo = i < j ? i : j, hi = i < j ? j : i;
          if (disp(&box, x + 3 * lo, x + 3 * hi, r) < cs) { if (cnt < cap) row[cnt] = (uint32_t)j; ++cnt; }
        }
      }
      /* the 27 cells are distinct (nc >= 3): no duplicates; order the partners */
      const int m = cnt < cap ? cnt : cap;
      for (int a = 1; a < m; ++a) { uint32_t v = row[a]; int b = a - 1; while (b >= 0 && row[b] > v) { row[b + 1] = row[b]; --b; } row[b + 1] = v; }
    } else {
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        if (disp(&box, x + 3 * lo, x + 3 * hi, r) < cs) { if (cnt < cap) row[cnt] = (uint32_t)j; ++cnt; }
      }
    }
    deg[i] = cnt;
    if (cnt > mx) mx = cnt;
  }
  free(cid); free(g.start); free(g.items);
  return mx;
}

static double forces_full_rows(const float* x, const uint32_t* rows, const int32_t* deg, int n, int cap, orc_box box,
                               float sigma, float eps, float rc, float* force) {
  const float eps4 = (float)(4.0 * (double)eps);
  double e_total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : e_total)
  for (int i = 0; i < n; ++i) {
    float f[3] = {0.f, 0.f, 0.f};
    float e_row = 0.0f;
    const uint32_t* row = rows + (size_t)i * cap;
    for (int k = 0; k < deg[i]; ++k) {
      const int j = (int)row[k];
      const int i_low = i < j;
      float r[3];
      const float d = i_low ? disp(&box, x + 3 * i, x + 3 * j, r) : disp(&box, x + 3 * j, x + 3 * i, r);
      if (!(d < rc)) continue;
      float s = sigma / d, s2 = s * s, s4 = s2 * s2, s8 = s4 * s4;
      float s6 = s2 * s4, s12 = s4 * s8;
      if (i_low) e_row += eps4 * (s12 - s6);   /* each pair's energy once, in its half-list row */
      const float fs = 24.0f * (eps / (d * d)) * (2.0f * s12 - s6);
      for (int c = 0; c < 3; ++c) { if (i_low) f[c] += fs * r[c]; else f[c] -= fs * r[c]; }
    }
    force[3 * i] = f[0]; force[3 * i + 1] = f[1]; force[3 * i + 2] = f[2];
    e_total += (double)e_row;
  }
  return e_total;
}

static int rebuild_full(const float* x, int n, orc_box box, float cs, int* cap, uint32_t** rows, int32_t* deg) {
  for (;;) {
    *rows = (uint32_t*)realloc(*rows, sizeof(uint32_t) * (size_t)n * (size_t)(*cap));
    if (!*rows) return -1;
    const int mx = build_full_rows(x, n, box, cs, *cap, *rows, deg);
    if (mx <= *cap) return 0;
    *cap = mx + 16;
  }
}

int orc_langevin_run_cell_omp(float* x, float* v, int n, orc_box box, const float* mass, float sigma, float eps,
                              float rc, float skin, int cap0, float dt, float a, float b, float kT, int n_steps,
                              const float* noise, uint32_t k0, uint32_t k1, int threads, double* stats) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const float hdt = (float)((double)dt * 0.5);
  const float cs = (float)((double)rc + (double)skin);
  int cap = cap0 > 0 ? cap0 : 96, rebuilds = 0;
  double t_build = 0.0;
  uint32_t* rows = NULL;
  int32_t* deg = (int32_t*)malloc(sizeof(int32_t) * n);
  float* ref = (float*)malloc(sizeof(float) * 3 * n);
  float* F = (float*)malloc(sizeof(float) * 3 * n);
  if (!deg || !ref || !F) return -1;
  double t0 = now_s();
  if (rebuild_full(x, n, box, cs, &cap, &rows, deg)) return -1;
  t_build += now_s() - t0;
  memcpy(ref, x, sizeof(float) * 3 * n);
  forces_full_rows(x, rows, deg, n, cap, box, sigma, eps, rc, F);
  const float L[3] = {box.lx, box.ly, box.lz};
  for (int s = 0; s < n_steps; ++s) {
    int moved = 0;
#pragma omp parallel for schedule(static) reduction(| : moved)
    for (int i = 0; i < n; ++i) {
      const float mi = mass[i];
      const float sv = sqrtf(kT / mi);
      float z[3];
      if (noise) memcpy(z, noise + ((size_t)s * n + i) * 3, sizeof(z));
      else normals3((uint32_t)i, (uint32_t)s, 1u, k0, k1, z);
      for (int c = 0; c < 3; ++c) {
        float vv = v[3 * i + c] + hdt * F[3 * i + c] / mi;
        float xx = x[3 * i + c] + hdt * vv;
        vv = (a * vv) + (b * sv * z[c]);
        xx = xx + hdt * vv;
        if (box.periodic) xx = xx - floorf(xx / L[c]) * L[c];
        v[3 * i + c] = vv; x[3 * i + c] = xx;
      }
      float r[3];
      if (disp(&box, x + 3 * i, ref + 3 * i, r) >= skin / 2.0f) moved |= 1;
    }
    if (moved) {
      t0 = now_s();
      if (rebuild_full(x, n, box, cs, &cap, &rows, deg)) return -1;
      t_build += now_s() - t0;
      memcpy(ref, x, sizeof(float) * 3 * n);
      ++rebuilds;
    }
    forces_full_rows(x, rows, deg, n, cap, box, sigma, eps, rc, F);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c) v[3 * i + c] = v[3 * i + c] + hdt * F[3 * i + c] / mass[i];
  }
  if (stats) {
    stats[0] = rebuilds; stats[1] = cap; stats[2] = t_build;
#ifdef _OPENMP
    stats[3] = omp_get_max_threads();
#else
    stats[3] = 1;
#endif
  }
  free(rows); free(deg); free(ref); free(F);
  return 0;
}

/* ==========================================================================
 * Full-size checkers (tests/test_gpu_fullsize.py): every row of the list and every particle's force at
 * N = 32 768 / 262 144, where the O(N^2) routines above are out of reach.
 * ========================================================================== */

/* Full rows (all partners j != i with d_ref < cs in the reference's fp32 arithmetic and orientation, ascending id)
 * through the cell grid above; returns the largest degree (> cap: grow and call again).  The reference's half row
 * i (chiron/neighbors.py:548-626) is the part of full row i with j > i. */
int orc_full_rows_cell(const float* x, int n, orc_box box, float cs, int cap, uint32_t* rows, int32_t* deg, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  return build_full_rows(x, n, box, cs, cap, rows, deg);
}

/* LJ energy and forces on ALL particles in fp64 (chiron/potential.py:193-213, closed-form gradient :322-332).
 * The interacting SET is the reference's: the fp32 predicate d < rc of chiron/neighbors.py:782 on the fp32 minimum
 * image (a pair one ulp from the cutoff must not flip between oracle and kernel: its force is ~0.03 kJ/mol/nm,
 * far above the tolerance); the VALUES are fp64 on the same coordinates with an fp64 minimum image.
 * force64 (n,3); stats[0] = E, [1] = sum over pairs |e_ij|, [2] = max_i max_c sum_j |f_ij,c|, [3] = pairs.
 * Returns 0, or the required row capacity if `cap` was too small. */
/* geom32 != 0: the pair vectors are the REFERENCE's fp32 ones (x_i - x_j, mod(r + L/2, L) - L/2 in fp32,
 * chiron/neighbors.py:69-79 -- the kernels reproduce them bit for bit), and only the Lennard-Jones arithmetic and the
 * sums are fp64.  At r ~ sigma the fp32 minimum image alone (one rounding at magnitude L/2) moves a pair force by up to
 * 13 ulp(L/2) / (2 r) ~ 2e-5 (L = 23 nm) .. 4e-5 (L = 47 nm) of its size, more than SURVEY 8(c)'s 1e-5: on a clustered
 * fluid the tolerance against pure fp64 geometry measures the reference's formula, not the kernel.
 * fabs_out (n,3) or NULL: per particle and component, sum_j |f_ij,c|.
 * fterm_out (n,3) or NULL: sum_j (|repulsive| + |attractive|) = sum_j 24 eps (2 s^12 + s^6) / d^2 |r_c| -- the scale of
 * the particle's fp32 rounding noise: near d = 2^(1/6) sigma the two terms cancel in f_ij while their rounding errors
 * do not. */
int orc_lj_cell_f64_ex(const float* x, int n, orc_box box, float sigma, float eps, float rc, int cap, double* force64,
                       double* fabs_out, double* fterm_out, double* stats, int threads, int geom32) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  uint32_t* rows = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n * (size_t)cap);
  int32_t* deg = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  if (!rows || !deg) { free(rows); free(deg); return -1; }
  const int mx = build_full_rows(x, n, box, rc, cap, rows, deg);
  if (mx > cap) { free(rows); free(deg); return mx; }
  const double L[3] = {box.lx, box.ly, box.lz};
  double e_tot = 0.0, e_abs = 0.0, f_abs_max = 0.0, pairs = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : e_tot, e_abs, pairs) reduction(max : f_abs_max)
  for (int i = 0; i < n; ++i) {
    double f[3] = {0, 0, 0}, fa[3] = {0, 0, 0}, ft[3] = {0, 0, 0};
    const uint32_t* row = rows + (size_t)i * cap;
    for (int k = 0; k < deg[i]; ++k) {
      const int j = (int)row[k];
      double r[3], d2 = 0.0;
      if (geom32) {
        /* the reference's orientation: lower id minus higher id (chiron/neighbors.py:773-779) */
        float r32[3];
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        (void)disp(&box, x + 3 * lo, x + 3 * hi, r32);
        for (int c = 0; c < 3; ++c) { r[c] = (i < j) ? (double)r32[c] : -(double)r32[c]; d2 += r[c] * r[c]; }
      } else {
        for (int c = 0; c < 3; ++c) {
          r[c] = (double)x[3 * i + c] - (double)x[3 * j + c];
          if (box.periodic) r[c] -= L[c] * nearbyint(r[c] / L[c]);
          d2 += r[c] * r[c];
        }
      }
      const double s2 = (double)sigma * (double)sigma / d2, s6 = s2 * s2 * s2;
      const double e = 4.0 * (double)eps * (s6 * s6 - s6);
      const double fs = 24.0 * (double)eps * (2.0 * s6 * s6 - s6) / d2;
      const double ts = 24.0 * (double)eps * (2.0 * s6 * s6 + s6) / d2;
      for (int c = 0; c < 3; ++c) { f[c] += fs * r[c]; fa[c] += fabs(fs * r[c]); ft[c] += ts * fabs(r[c]); }
      if (i < j) { e_tot += e; e_abs += fabs(e); pairs += 1.0; }
    }
    for (int c = 0; c < 3; ++c) {
      force64[3 * (size_t)i + c] = f[c];
      if (fabs_out) fabs_out[3 * (size_t)i + c] = fa[c];
      if (fterm_out) fterm_out[3 * (size_t)i + c] = ft[c];
      if (fa[c] > f_abs_max) f_abs_max = fa[c];
    }
  }
  if (stats) { stats[0] = e_tot; stats[1] = e_abs; stats[2] = f_abs_max; stats[3] = pairs; }
  free(rows); free(deg);
  return 0;
}

int orc_lj_cell_f64(const float* x, int n, orc_box box, float sigma, float eps, float rc, int cap, double* force64,
                    double* stats, int threads) {
  return orc_lj_cell_f64_ex(x, n, box, sigma, eps, rc, cap, force64, NULL, NULL, stats, threads, 0);
}
