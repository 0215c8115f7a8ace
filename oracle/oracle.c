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
