/*
 * chx.h -- C ABI of libchx.so, the MI355X (gfx950) implementation of chiron's
 * Lennard-Jones / neighbor-list / Langevin / Metropolis hot path.
 *
 * The reference (choderalab/chiron) is pure Python on JAX and has no FFI of its
 * own; each entry point below replaces one jitted function / eager op chain of
 * the reference (cited as file:line relative to the reference checkout) and is
 * what a ctypes binding on the reference side would call.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _h.
 *   - all launches go to `stream` (a hipStream_t; NULL = default stream) and
 *     are asynchronous; no entry point allocates or synchronises unless it says so.
 *   - return value: 0 = CHX_OK, negative = error (chx_last_error() has text).
 *   - arrays are C-contiguous fp32 / int32 / uint32; xyz arrays are (n,3).
 *   - boxes are orthorhombic: chx_box_t{periodic, lx, ly, lz} by value.  In every
 *     signature the box comes BEFORE any scalar float argument unless fewer than five
 *     integer/pointer arguments precede it: some libffi builds (ctypes, cgo-less FFIs)
 *     mis-marshal a mixed int/float struct that takes the last integer register after
 *     float registers are already in use.
 */
#ifndef CHX_H
#define CHX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* chx_stream_t;

#define CHX_OK 0
#define CHX_ERR_INVALID (-1)
#define CHX_ERR_LAUNCH (-2)
#define CHX_ERR_UNSUPPORTED (-3)

/* ABI history (chx_version() returns this; a binding compares it with the header it was written against):
 *   1  round 1.
 *   2  round 2/3 -- INCOMPATIBLE with 1: chx_pack_langevin takes three more pointers (ctrl_reset, publish_src8,
 *      publish_host_dst8); chx_langevin_t grew host_ctrl (and, in round 3, the reference-event fields below it); the
 *      fused step kernel takes the particle's MASS from ref4[p].w (a ref4 packed with chx_permute_pack has w = 0 and
 *      is refused: chx_langevin_fused_chunk / _run check ref4[0].w > 0 when CHX_DEBUG_SYNC is set, and
 *      chx_pack_langevin is the one supported producer); x4[p].w belongs to the kernel (path length); the words of a
 *      force-list chunk are interleaved over the lanes (see "Force list" below); chx_mc_t::state is
 *      CHX_MC_STATE_WORDS = 16 ints (two slots) and chx_mc_t::energy 4 doubles -- the struct's SIZE did not change,
 *      so chx_sizeof_mc_t() cannot tell; this version number does.  Added in round 3 without touching any existing
 *      signature: chx_repack_langevin, chx_validate_snap, chx_set_poll_sleep_ns, chx_poll_stats.
 *   3  round 4 -- INCOMPATIBLE with 2: the deferred-rebuild scheme of round 3 is gone (measured: no gain,
 *      profiles/r04_internal_skin.md) and with it chx_repack_langevin, chx_langevin_t::trigger_frac (ev_log_cap
 *      moves into its slot, so the struct's SIZE does not change: this version number tells), control word [12], the
 *      watch_trigger argument of
 *      chx_langevin_fused_run and the max_blocks argument of chx_nl_build_cell_fl.  Both cell builds now expect
 *      cell_count zeroed on entry and leave it zeroed; chx_nl_build_cell_fl takes a chx_build_riders_t* (copy of the
 *      positions, packed Langevin state, control-block reset, status publish: see there).  chx_langevin_t grew
 *      final_x3 / final_v3 / final_ref3 (the run's last launch writes the state in user order).  chx_mc_t: x_a / x_b
 *      became cur4 / prop4 (four configurations each) and ws doubled -- the Monte Carlo loop is one launch per trial.
 *      Added: chx_philox4x32_h, chx_cell_bin_fl, chx_nl_rows_cell_fl, chx_mc_key_chain_h (chx_mc_t grew k0, k1,
 *      keytab, keytab_h, chain_out_h: the chunk's key chain is computed on the host). */
/*   4  round 5 -- compatible with 3 for every existing signature.  Added: the STEP QUEUE.  chx_langevin_fused_chunk / _run
 *      dispatch their step launches as AQL packets on a private HSA queue paired with `stream` (see chx_steps_join);
 *      "asynchronous on `stream`" then means: behind everything enqueued on `stream` before the call, and every LATER
 *      libchx call on `stream` runs behind them -- but a caller that enqueues work of its OWN on `stream` which reads
 *      what the steps write must call chx_steps_join first.  CHX_AQL=0 in the environment restores plain HIP launches.
 *      New entry points: chx_steps_join, chx_steps_open, chx_step_queue_info, chx_step_queue_status. */
/*   5  round 6 -- compatible with 4 for every existing signature.  Added: chx_queue_debug / chx_queue_debug_spins (the step
 *      queue's delay mode), chx_langevin_batch_run and chx_langevin_batch_kick (one launch per step for SEVERAL replicas).
 *      Memory contract stated (it held before): every device pointer handed to an entry point that may ride on the step
 *      queue must be device-local coarse-grained memory (hipMalloc / a torch CUDA tensor), not fine-grained, host-pinned or
 *      peer memory -- the fused step and Monte Carlo trial packets carry no acquire fence. */
#define CHX_ABI_VERSION 5

/* high bit of a force-list entry: neighbor has the HIGHER original particle id */
#define CHX_FL_HI 0x80000000u

typedef struct {
  int32_t periodic; /* 0: OrthogonalNonPeriodicSpace, else OrthogonalPeriodicSpace */
  float lx, ly, lz; /* diag(box_vectors), chiron/neighbors.py:74-76 */
} chx_box_t;

/* --------------------------------------------------------------------- misc */
int chx_version(void);
const char* chx_last_error(void);
/* Build identity: sha256 (hex) over the library's sources (every .hip and .h under csrc/, include/chx.h) and compiler
 * flags, baked in by chiron_amd/build.py.  The loader refuses a library whose hash differs from the sources
 * beside it, and build() recompiles exactly when it does (a stale binary cannot pass for a fresh one). */
const char* chx_source_hash(void);
/* sizeof(chx_langevin_t) / sizeof(chx_mc_t) as compiled: bindings that mirror the structs by hand
 * (ctypes, cgo, JNI) assert their own layout against these at load time. */
int64_t chx_sizeof_langevin_t(void);
int64_t chx_sizeof_mc_t(void);
int64_t chx_sizeof_build_riders_t(void);

/* Philox4x32-10 on the host (the generator every device stream of this library draws from; Random123 known answers in
 * the tests): out4_h = philox(counter (c0..c3), key (k0, k1)).  Key handling of the bindings (split, fold_in, uniform)
 * is built on it. */
void chx_philox4x32_h(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4_h);

/* Largest fp32 t with sqrtf(t) < c (host helper, exact).  `d < c` on the
 * reference's d = sqrt(r2) is then equivalent to `r2 <= T` (neighbors.py:600-602,
 * :782, :1156); `d >= c` is `r2 > T` (:868). */
float chx_cutoff_threshold_h(float c);

/* ------------------------------------------------------------------- spaces */
/* Space.displacement: r = x1 - x2 [min-image], dist = |r|.
 * chiron/neighbors.py:45-83 (periodic), :116-150 (non-periodic). n rows of 3. */
int chx_displacement(const float* x1, const float* x2, int64_t n, chx_box_t box,
                     float* rij_out, float* dist_out, chx_stream_t stream);

/* OrthogonalPeriodicSpace.wrap: x - floor(x/L)*L (identity when !periodic).
 * chiron/neighbors.py:85-112, :152-175.  xin may equal xout. */
int chx_wrap(const float* xin, float* xout, int64_t n, chx_box_t box, chx_stream_t stream);

/* ---------------------------------------------- NeighborListNsqrd (Verlet) */
/* One pass of vmap(_build_neighborlist), O(N^2): chiron/neighbors.py:548-626,
 * :694-704.  Per row i: ascending j>i with d(ref_i,ref_j) < cutoff+skin, the
 * first m of them, padded with the fill value of :606-609; nl_mask = arange(m)
 * < n_i; n_nbr = full (untruncated) count. */
int chx_nl_build_nsq(const float* xyz, int32_t n, int32_t m, chx_box_t box, float cutoff_plus_skin,
                     uint32_t* nl_out, int32_t* nl_mask_out, int32_t* n_nbr_out,
                     chx_stream_t stream);

/* Cell-list geometry for chx_nl_build_cell (host helper).  Writes the grid
 * dimensions to nc_h[3]; returns the number of cells, or 0 when the cell path
 * does not apply (non-periodic box or fewer than 3 cells along an axis) and
 * the caller must use chx_nl_build_nsq. */
int64_t chx_cell_grid_h(chx_box_t box, float cutoff_plus_skin, int32_t n, int32_t* nc_h);

/* New (the reference has no cell list, chiron/neighbors.py:529-530): O(N)
 * build that produces BIT-IDENTICAL nl/nl_mask/n_nbr to chx_nl_build_nsq.
 * Workspace (caller-allocated): cell_of[n] i32, cell_count[ncell+1] i32 (ALL ZERO on
 * entry, all zero again afterwards: ABI 3), cell_start[ncell+1] i32, sorted_ids[n] u32 (out: particle ids in cell
 * order, ascending id inside a cell), status[4] i32 (out: [0]!=0 -> a row
 * overflowed the on-chip candidate buffer, results invalid, rebuild with
 * chx_nl_build_nsq). */
int chx_nl_build_cell(const float* xyz, int32_t n, int32_t m, chx_box_t box, float cutoff_plus_skin,
                      const int32_t* nc_h, int32_t* cell_of, int32_t* cell_count,
                      int32_t* cell_start, uint32_t* sorted_ids, int32_t* status,
                      uint32_t* nl_out, int32_t* nl_mask_out, int32_t* n_nbr_out,
                      chx_stream_t stream);

/* One-pass build for the hot loop: the same bins and the same predicate as
 * chx_nl_build_cell, but (a) only the REAL entries of each half-list row are written
 * (nl_rows_out[i, :min(n_nbr[i], m)]), with first_out[i] = smallest neighbor id (or
 * 0xFFFFFFFF) so that chx_nl_finish_padding can later produce the reference's padding
 * (chiron/neighbors.py:603-620) and mask (:623) if somebody reads them, and (b) the
 * force list (layout above, internal order = sorted_ids, g lanes, capacity chunks*4*g
 * entries per particle, unused slots = 0) and deg_out[n] are emitted directly, in scan
 * order, without atomics.
 * Workspace contracts (ABI 3): cell_count (ncell + 1 ints) must be ALL ZERO on entry -- the caller zeroes it once, when
 * it allocates it -- and is all zero again when the launches have run (the chain needs no zeroing launch of its own);
 * status8 (out): [0] != 0 -> a half row overflowed the on-chip candidate buffer (nothing valid); [1] max n_nbr;
 * [2] number of rows with n_nbr == m (the reference's growth trigger, :709); [3] max full degree; [4] rows whose full
 * degree exceeded the capacity (force list invalid, retry with more chunks); [5] != 0 -> a cell overflowed its id
 * slots (chx_build_riders_t::cell_slots; nothing valid, retry without); [6] reserved; [7] scratch.
 * xs4_out (n,4): the positions in internal (cell) order, x4[p] = xyz[sorted_ids[p]].
 * nl_rows_out == NULL together with first_out == NULL (both builds below, accepted since round 4): the half rows are
 * COUNTED -- n_nbr_out and status8[1], [2] as above -- but not written or sorted: a list rebuild inside the Langevin loop,
 * whose step kernels read the force list only (the rows are a function of xyz and can be produced by a later call).
 * riders (may be NULL; every member optional): work that rides on the chain's launches instead of costing launches of
 * its own when the build is a list rebuild inside the Langevin loop (chiron/integrators.py:239-242) -- six launches
 * where round 3 used ten. */
typedef struct {
  float* xyz_copy_out;        /* (n,3): a copy of xyz (the list's private reference positions, chiron/neighbors.py:671) */
  /* the Langevin state packed into the new internal order, exactly what chx_pack_langevin(xyz, v3, mass, ref3 or xyz,
   * sorted_ids, ...) would write afterwards; on when x4_out != NULL (then v3, mass, v_out, mass_out are required): */
  const float* v3;            /* (n,3) velocities, user order */
  const float* mass;          /* (n)   masses, user order */
  const float* ref3;          /* (n,3) reference positions of the skin check, user order; NULL = xyz itself */
  float* x4_out;              /* (n,4) (xyz, 0) */
  float* v_out;               /* (n,3) */
  float* mass_out;            /* (n) */
  float* ref4_out;            /* (n,4) (ref xyz, mass), or NULL */
  int32_t* ctrl_reset;        /* CHX_CTRL_WORDS ints: what chx_ctrl_reset writes ... */
  int32_t ctrl_keep;          /* ... except the words chx_pack_langevin's ctrl_keep names */
  int32_t* publish_host_dst8; /* HOST-MAPPED: status[0..7] is stored there (system scope) by the chain's last launch */
  /* Optional workspace that makes the per-particle row kernel ~30 % faster: the binning writes every cell's 27-cell
   * neighbourhood out once as a list of candidate indices (cand_list: ncell * cand_cap words, cand_len: ncell ints),
   * which the ~10 rows of the cell then read instead of each working the neighbourhood's index ranges out and searching
   * them.  A neighbourhood longer than cand_cap keeps the search; same bits either way.  NULL / 0 = off.
   * Round 5: with the lists in place a build that writes NO half rows (nl_rows_out == NULL: the rebuilds inside the
   * Langevin loop) takes a third row kernel at every grid size -- sixteen lanes per row, candidates through a cheap
   * nearest-image test before the reference's predicate sees the survivors; same bits again, the row launch 25 -> 17.5 us
   * at N = 32 768 and 183 -> 93 us at N = 262 144 (where the lists cost the binning 25 us and 98 MB).  It needs
   * cand_cap <= 65536 and ncell * cand_cap < 2^29; CHX_BUILD_SUB=0 in the environment keeps the other kernels. */
  uint32_t* cand_list;
  int32_t* cand_len;
  int32_t cand_cap;
  /* Optional workspace that saves the chain one launch (five instead of six): ncell * cell_slot_cap words in which the
   * counting launch keeps every particle's id under (its cell, its arrival order), so that the per-cell sort finds a
   * cell's ids without a scatter launch in between.  A cell with more particles than cell_slot_cap voids the build:
   * status8[5] != 0, repeat it with cell_slots = NULL (the list written is safe to walk, not correct).  NULL / 0 = off. */
  uint32_t* cell_slots;
  int32_t cell_slot_cap;
} chx_build_riders_t;
int chx_nl_build_cell_fl(const float* xyz, int32_t n, int32_t m, chx_box_t box, float cutoff_plus_skin,
                         const int32_t* nc_h, int32_t* cell_of, int32_t* cell_count,
                         int32_t* cell_start, uint32_t* sorted_ids, int32_t* status8,
                         uint32_t* nl_rows_out, uint32_t* first_out, int32_t* n_nbr_out, int32_t g,
                         int32_t chunks, uint32_t* fl_out, int32_t* deg_out, float* xs4_out,
                         const chx_build_riders_t* riders, chx_stream_t stream);
/* The same build in two calls, for a caller that wants the binning launches on the device before it has allocated the
 * arrays the rows go into (the integrator's run() entry: the device idles until the first launch is made):
 * chx_cell_bin_fl = the binning launches with the riders (publish_host_dst8 is ignored here), chx_nl_rows_cell_fl = the
 * row kernel and the summary, which publishes status8 to publish_host_dst8 when that is given.  Same arguments, same
 * contracts, same bits as the one call. */
int chx_cell_bin_fl(const float* xyz, int32_t n, chx_box_t box, float cutoff_plus_skin, const int32_t* nc_h,
                    int32_t* cell_of, int32_t* cell_count, int32_t* cell_start, uint32_t* sorted_ids,
                    int32_t* status8, float* xs4_out, const chx_build_riders_t* riders, chx_stream_t stream);
int chx_nl_rows_cell_fl(int32_t n, int32_t m, chx_box_t box, float cutoff_plus_skin, const int32_t* nc_h,
                        const int32_t* cell_of, const int32_t* cell_start, const uint32_t* sorted_ids,
                        int32_t* status8, const float* xs4, uint32_t* nl_rows_out, uint32_t* first_out,
                        int32_t* n_nbr_out, int32_t g, int32_t chunks, uint32_t* fl_out, int32_t* deg_out,
                        int32_t* publish_host_dst8, const uint32_t* cand_list, const int32_t* cand_len,
                        int32_t cand_cap, chx_stream_t stream);
int chx_nl_finish_padding(uint32_t* nl_inout, const uint32_t* first, const int32_t* n_nbr, int32_t n,
                          int32_t m, int32_t* nl_mask_out, chx_stream_t stream);

/* NeighborListNsqrd.calculate: chiron/neighbors.py:731-826.  For every slot
 * (i,k): r = disp(x_i, x_nl[i,k]); mask = (d < cutoff) & nl_mask; n_pairs = row sum. */
int chx_nl_calculate(const float* xyz, const uint32_t* nl, const int32_t* nl_mask, int32_t n,
                     int32_t m, chx_box_t box, float cutoff, int32_t* n_pairs_out,
                     int32_t* mask_out, float* dist_out, float* rij_out, chx_stream_t stream);

/* NeighborListNsqrd.check body: flag_out[0] = any_i |disp(x_i, ref_i)| >= skin/2.
 * chiron/neighbors.py:828-907.  (The n-changed test of :890 is host logic.) */
int chx_nl_check(const float* xyz, const float* ref_xyz, int32_t n, chx_box_t box, float skin,
                 int32_t* flag_out, chx_stream_t stream);

/* ------------------------------------------------------------ PairListNsqrd */
/* _pairs_and_mask: all_pairs[i] = all j != i ascending, reduction_mask = i < j.
 * chiron/neighbors.py:1018-1067. */
int chx_pl_build(int32_t n, uint32_t* all_pairs_out, uint8_t* reduction_mask_out,
                 chx_stream_t stream);

/* PairListNsqrd.calculate: chiron/neighbors.py:1106-1270.  cutoff < 0 means
 * cutoff=None (mask = i<j only).  Outputs are (n, n-1[,3]). */
int chx_pl_calculate(const float* xyz, int32_t n, chx_box_t box, float cutoff,
                     int32_t* n_pairs_out, int32_t* mask_out, float* dist_out, float* rij_out,
                     chx_stream_t stream);

/* ----------------------------------------------------- LJ energy and force */
/* Force list: an internal FULL neighbor table derived from the public half
 * list, laid out for the force kernel: particle p (internal order, p =
 * inv[original id], inv may be NULL = identity) owns deg[p] entries; g lanes
 * cooperate on one particle and a row is stored in chunks of W = 4 g entries,
 * interleaved over the lanes inside a chunk: entry k (e = k % W) lives at
 * fl[((k / W) * n + p) * W + 4 (e % g) + e / g], so lane q reads words 4q..4q+3
 * of a chunk with one 16-byte load (fl must be 16-byte aligned) and word u of
 * lane q is row entry u g + q: the u-th gather of a lane group touches g
 * consecutive row entries.  Entry = internal index of the neighbor | CHX_FL_HI when the
 * neighbor's ORIGINAL id is the larger one (the pair is then stored in this
 * particle's row of the reference half list and r_ij = x_p - x_q as in
 * chiron/neighbors.py:777-779; otherwise r_ij = x_q - x_p and the force on p
 * is the reaction).  Exactly the pairs of nl[:, :min(n_nbr, m)] are present. */
int chx_fl_count(const uint32_t* nl, const int32_t* n_nbr, int32_t n, int32_t m,
                 const uint32_t* inv, int32_t* deg_out, int32_t* maxdeg_out, chx_stream_t stream);
int chx_fl_fill(const uint32_t* nl, const int32_t* n_nbr, int32_t n, int32_t m,
                const uint32_t* inv, int32_t g, int32_t* cursor_ws, uint32_t* fl_out,
                chx_stream_t stream);
/* sort each particle's entries by neighbor index (locality + reproducible sums) */
int chx_fl_sort(uint32_t* fl, const int32_t* deg, int32_t n, int32_t g, int32_t maxdeg,
                chx_stream_t stream);

/* dst[p] = src[perm[p]] (rows of 3 floats in, rows of dst_stride floats out, pad = 0),
 * and the inverse dst[perm[p]] = src[p].  perm NULL = identity. */
int chx_permute_pack(const float* src3, const uint32_t* perm, int32_t n, float* dst,
                     int32_t dst_stride, chx_stream_t stream);
int chx_permute_unpack(const float* src, int32_t src_stride, const uint32_t* perm, int32_t n,
                       float* dst3, chx_stream_t stream);
int chx_invert_perm(const uint32_t* perm, int32_t n, uint32_t* inv_out, chx_stream_t stream);
/* one launch, the inverse of chx_pack_langevin: x3_out[perm[p]] = x4[p].xyz, v3_out[perm[p]] = v[p] */
int chx_unpack_langevin(const float* x4, const float* v, const uint32_t* perm, int32_t n,
                        float* x3_out, float* v3_out, chx_stream_t stream);
/* one launch: x4[p] = x3[perm[p]], v[p] = v3[perm[p]], mass_out[p] = mass[perm[p]],
 * ref4[p] = (ref3[perm[p]], mass_out[p]) (ref4_out may be NULL).  Optional riders, so that a list rebuild inside
 * the Langevin loop costs one launch here instead of three: ctrl_reset != NULL -> what chx_ctrl_reset writes, except
 * that ctrl_keep bit 0 leaves the reference-event counters ([8], [9]) and bit 1 the event words ([10], [11]) as they
 * are (carried-over force list: a renewal of the list must not lose a rebuild event that is booked for the very next
 * launch, nor the count so far);
 * publish_src8 / publish_host_dst8 != NULL -> what chx_publish(src, dst, 8) does (the build's status block). */
/* chx_pack_langevin (with its control-block reset) and chx_validate_snap (below) in ONE launch, for a run that starts on a
 * carried-over force list: the halt word of the freshly reset control block is `step` if some particle is farther than
 * skin / 2 from snap4 (the positions the force list was built from, internal order), as is host_ctrl[3] (HOST-MAPPED,
 * may be NULL).  sync2: two ints of device memory, zero on entry (the caller zeroes them once), zero again afterwards.
 * ctrl_reset is required.  Added in round 4. */
int chx_pack_validate_langevin(chx_box_t box, float skin, int32_t step, const float* snap4, int32_t* host_ctrl,
                               int32_t* sync2, const float* x3, const float* v3, const float* mass, const float* ref3,
                               const uint32_t* perm, int32_t n, float* x4_out, float* v_out, float* mass_out,
                               float* ref4_out, int32_t* ctrl_reset, int32_t ctrl_keep, chx_stream_t stream);
int chx_pack_langevin(const float* x3, const float* v3, const float* mass, const float* ref3,
                      const uint32_t* perm, int32_t n, float* x4_out, float* v_out, float* mass_out,
                      float* ref4_out, int32_t* ctrl_reset, int32_t ctrl_keep, const int32_t* publish_src8,
                      int32_t* publish_host_dst8, chx_stream_t stream);

/* List carry-over (LangevinIntegrator: a run() that starts on the force list its predecessor ended on instead of
 * building one): halt launch `step` -- ctrl[3] = min(ctrl[3], step), and the same into host_ctrl[3] when given -- unless
 * every particle of x4 (n,4; internal order of that list) is within skin/2 of the position the list was built from
 * (snap4, same order; minimum image), the guarantee under which a list of radius cutoff + skin holds every pair inside
 * the cutoff (chiron/neighbors.py:872-907 states it for the reference positions).  Enqueued behind chx_pack_langevin
 * (which resets ctrl) and in front of the first step launch. */
int chx_validate_snap(const float* x4, const float* snap4, int32_t n, chx_box_t box, float skin, int32_t step,
                      int32_t* ctrl, int32_t* host_ctrl, chx_stream_t stream);

/* LJPotential.compute_energy / compute_force over the neighbor list:
 * E = sum over listed pairs with d < cutoff of 4 eps ((s/d)^12 - (s/d)^6)
 * (chiron/potential.py:193-213, :260-279), F = -dE/dx (chiron/potential.py:21-24,
 * closed form :322-332).  x4 is (n,4) in internal order; force3_out (n,3) in
 * internal order (NULL = energy only); energy_out[1] (NULL = force only);
 * chunks = allocated chunks of fl (fl holds chunks*n*4*g entries; the slots between a
 * row's end and the end of its last touched chunk -- at least chunk 0 -- must hold a
 * valid index: the kernel loads the first entries of every row before it knows the
 * row's length and gathers whole chunks.  chx_fl_fill and the cell build write the
 * row's own index there, a zero-length pair that keeps the kernel on its fast path);
 * ws: >= chx_lj_ws_count(n) doubles. */
int64_t chx_lj_ws_count(int32_t n);
int chx_lj_fl_energy_force(const float* x4, const uint32_t* fl, const int32_t* deg, int32_t n,
                           int32_t g, int32_t chunks, chx_box_t box, float sigma, float epsilon, float cutoff,
                           float* force3_out, float* energy_out, double* ws, chx_stream_t stream);

/* The same energy / force evaluated directly on the PUBLIC half list (neighbor_list (n,m),
 * n_neighbors (n), user order, xyz (n,3)) -- chiron/potential.py:260-279 on the arrays of
 * chiron/neighbors.py:617-626 -- for callers that do not keep a force list.  One wavefront
 * per row; reaction forces are scattered with float atomics, so the force summation order
 * is not reproducible (the energy is).  force_out (n,3) or NULL, energy_out[1] or NULL. */
int chx_lj_nl_energy_force(const float* xyz, const uint32_t* nl, const int32_t* n_nbr, int32_t n,
                           int32_t m, chx_box_t box, float sigma, float epsilon, float cutoff,
                           float* energy_out, float* force_out, double* ws, chx_stream_t stream);

/* All-pairs kernel: the nbr_list=None branch (no PBC, chiron/potential.py:26-63,
 * :235-258) and the PairListNsqrd branch (PBC, chiron/neighbors.py:1106-1216).
 * cutoff < 0: no cutoff.  xyz/force are (n,3) in user order. */
int chx_lj_allpairs_energy_force(const float* xyz, int32_t n, chx_box_t box, float sigma,
                                 float epsilon, float cutoff, float* force_out,
                                 float* energy_out, double* ws, chx_stream_t stream);

/* ------------------------------------------------------- Langevin (BAOAB) */
/* initialize_velocities: v = sqrt(kT/m) * N(0,1), chiron/utils.py:116-144.
 * normals (n,3) optional injected noise; otherwise Philox stream 0 keyed (k0,k1). */
int chx_init_velocities(float* v_out, const float* mass, int32_t n, float kT, uint32_t k0,
                        uint32_t k1, const float* normals, chx_stream_t stream);

/* Optional JAX-compatible noise source: out[0..count) = jax.random.normal(key, (count,))
 * for a classic threefry key (k0, k1) = the two uint32 words of the jax key -- what the reference draws at
 * chiron/utils.py:142 (velocities), chiron/integrators.py:185 (O step) and chiron/mcmc.py:731 (displacements).
 * Feed it to chx_init_velocities / the `noise` field / chx_mc_displace as injected normals. */
int chx_jax_normal(uint32_t k0, uint32_t k1, int64_t count, float* out, chx_stream_t stream);

/* v += (dt/2) F / m, chiron/integrators.py:181,195 */
int chx_kick(float* v, const float* f, const float* mass, int32_t n, float half_dt,
             chx_stream_t stream);

typedef struct {
  int32_t n;
  int32_t g;             /* force-list lanes per particle */
  float* x4;             /* (n,4) positions, internal order            (in/out) */
  float* v;              /* (n,3) velocities, internal order           (in/out) */
  float* f;              /* (n,3) forces, internal order               (in/out) */
  float* ref4;           /* (n,4) positions at the reference's last list build, or NULL (no list); .w = the particle's
                            MASS (chx_pack_langevin writes it: the fused step kernel takes the mass from here).  Read
                            only, except with snap4 set (decoupled force list, below): the kernel then rewrites the xyz
                            of every row at a reference rebuild event */
  const float* mass;     /* (n)   masses, internal order */
  const uint32_t* perm;  /* (n)   internal -> original id (RNG counter / noise row), NULL = identity */
  const uint32_t* fl;    /* force list (see above), NULL -> all-pairs kernel */
  const int32_t* deg;
  chx_box_t box;
  int32_t wrap;          /* apply Space.wrap + skin check each step (nbr_list given) */
  float skin;            /* Verlet skin (check uses >= skin/2) */
  float half_dt, a, b, kT;
  float sigma, epsilon, cutoff;
  uint32_t k0, k1;       /* Philox key of this run */
  const float* noise;    /* optional injected O-step noise, (n_noise_steps, n, 3) in ORIGINAL order */
  int64_t noise_step0;   /* step index of noise row 0 */
  int32_t* ctrl;         /* CHX_CTRL_WORDS (16) ints.  [0] violation seen by integrate, [1] halt (set by force kernel),
                            [2] step at which the skin check fired, [3] fused path: first step that
                            must not run (INT32_MAX when clear), [4],[5] inner list stale (by step
                            parity; the value is the refresh epoch the word was raised in),
                            [6] launches that ran on the full list, [7] refreshes of the inner list;
                            decoupled force list (snap4 set): [8] reference rebuild events since the block was reset,
                            [9] step of the last one, [10],[11] event words by step parity (value s = "launch s starts
                            with a reference rebuild"; -1 when clear), [12..15] reserved */
  double* ws;            /* energy workspace, >= chx_lj_ws_count(n) */
  int32_t external_force; /* != 0: do not launch the LJ force kernel (the caller fills f
                             between single-step chunks: IdealGasPotential and other potentials) */
  int32_t fl_chunks;      /* allocated chunks of the force list: fl holds fl_chunks * n * 4 * g entries */
  /* Dynamically pruned inner list (fused path only; all NULL / 0 = off).  Every prune_every
   * steps, counted from step index prune_phase, the step launch walks fl itself and copies the
   * entries with |r| < cutoff + inner_skin into fl_inner (same geometry as fl), their counts into
   * deg_inner and restarts every particle's path length (the .w of the position buffers, a running sum of
   * per-step displacements); the following launches walk fl_inner until some particle's path exceeds
   * inner_skin / 2 and fl itself from then to the next refresh (ref_inner4 is unused since round 2).  The interacting pairs are those of fl in either case (csrc/chx_lj.hip: k_lj_step). */
  uint32_t* fl_inner;
  int32_t* deg_inner;
  float* ref_inner4;
  float inner_skin;
  int32_t prune_every;
  int64_t prune_phase;
  /* Optional (fused neighbor-list path): device-visible address of a HOST-MAPPED pinned block of 8 ints.  Every
   * step launch copies the control block there when it starts (system-scope stores by one wavefront), and a thread
   * that raises the halt word also stores it there directly, so the host can read the state of a chunk after
   * waiting for the stream WITHOUT a separate chx_publish launch.  What the host reads is complete up to the
   * chunk's last launch; a halt raised by that very launch is normally there too (its direct store) and otherwise
   * shows up with the next launch -- which is a no-op by then, so nothing is lost but a little time.  Word 0 of
   * the copy is the step index of the newest launch that has begun (not ctrl[0]); words 6 and 7 are also stored
   * whenever they change.  NULL = off. */
  int32_t* host_ctrl;    /* CHX_CTRL_WORDS ints */
  /* Decoupled force list (fused neighbor-list path; snap4 == NULL = off: the ordinary scheme described at
   * chx_langevin_fused_chunk).  The reference rebuilds its list -- and takes new reference positions -- whenever a
   * particle has moved skin/2 from them (chiron/integrators.py:239-242, chiron/neighbors.py:868); what the FORCE needs
   * is only a list that still holds every pair inside the cutoff.  With snap4 set the two are decoupled (what the list
   * carry-over between short run() calls runs on):
   *   - fl / deg were built from the positions snap4 (same cutoff + skin) and stay exhaustive while no particle is
   *     farther than skin/2 from its snap4 row: THAT check raises the halt word ctrl[3] as before;
   *   - the reference's own check against ref4 no longer halts anything: the launch that sees it fire for the
   *     positions of step s+1 stores s+1 in the event word of that parity, and launch s+1 starts by copying its input
   *     positions into ref4 (the reference's new reference positions, from which the caller can build the PUBLIC list
   *     whenever somebody wants to see it), counting the event in ctrl[8] / ctrl[9] and logging its step in
   *     ev_log[(count - 1) % ev_log_cap] (optional).
   * Rebuild events, reference positions and interacting pairs are exactly those of the ordinary scheme; trajectories
   * differ from it only through the summation order of the forces (the internal order follows snap4, not ref4). */
  const float* snap4;     /* (n,4) positions the force list was built from, internal order (.w is not read) */
  int32_t ev_log_cap;
  int32_t* ev_log;
  /* Optional outputs of the FORCE-ONLY launch that ends a run (chx_langevin_fused_chunk with do_aoa == 0; ignored by
   * every other launch): the final state in USER order, written by that launch itself instead of by a
   * chx_unpack_langevin launch behind it -- final_x3[perm[p]] = x[p].xyz, final_v3[perm[p]] = v[p] after the trailing
   * kick (chiron/integrators.py:195, :210-214), final_ref3[perm[p]] = ref4[p].xyz as the launch leaves it (decoupled
   * force list: the reference positions the caller's list is to be rebuilt from).  Each may be NULL.  Void when the
   * launch turned out to be a no-op (halt word <= its step): the caller checks the halt word as for any launch. */
  float* final_x3;
  float* final_v3;
  float* final_ref3;
} chx_langevin_t;

#define CHX_CTRL_WORDS 16

/* One BAOAB update on user-order (n,3) arrays, in place: v += half_dt F/m; x += half_dt v;
 * v = a v + b sqrt(kT/m) xi; x += half_dt v; [wrap]; rebuild_flag_out[0] = any particle moved
 * >= skin/2 from ref_xyz (chiron/integrators.py:181-188, :239-241).  xi = noise (n,3) if
 * given, else Philox stream 1 at counter (particle, step).  The caller evaluates the force
 * and applies the trailing chx_kick (:193-195).  ref_xyz / rebuild_flag_out may be NULL. */
int chx_baoab_step(float* x, float* v, const float* f, const float* mass, int32_t n, chx_box_t box,
                   int32_t wrap, float half_dt, float a, float b, float kT, const float* ref_xyz,
                   float skin, uint32_t k0, uint32_t k1, uint32_t step, const float* noise,
                   int32_t* rebuild_flag_out, chx_stream_t stream);

/* Enqueue n_steps BAOAB steps (chiron/integrators.py:179-195) starting at global
 * step index step0: per step one fused integrate kernel [n_kicks x B, A, O, A,
 * wrap, skin check] and one force kernel.  The first step applies kicks_first
 * half-kicks (1 at the start of a run or after a rebuild-resume that already
 * applied the trailing kick, else 2: the trailing kick of the previous step
 * merged with the leading one).  When the skin check fires at step s the
 * remaining launches of the chunk are no-ops: ctrl[1]=1, ctrl[2]=s, positions
 * and velocities are those after the A-O-A-wrap of step s (force NOT yet
 * recomputed) -- the caller rebuilds the list and resumes with
 * chx_langevin_resume.  Asynchronous; the caller synchronises and reads ctrl. */
int chx_langevin_chunk(const chx_langevin_t* p_h, int64_t step0, int32_t n_steps,
                       int32_t kicks_first, chx_stream_t stream);
/* force evaluation that completes a halted step after the list was rebuilt */
int chx_langevin_resume(const chx_langevin_t* p_h, chx_stream_t stream);

/* Fused neighbor-list path: ONE launch per MD step.  Launch i (global step step0+i)
 * reads positions from (i even ? x_even : x_odd), evaluates the pair force on the
 * force list, applies the half-kicks that consume it (kicks_first for i = 0, else 2:
 * trailing kick of the previous step + leading kick of this one), then A-O-A, wrap
 * and the skin check, and writes the new positions to the other buffer
 * (chiron/integrators.py:179-195 with :193-195 of step s-1 merged into step s).
 * p_h->x4 / p_h->f are ignored.  The .w of the two position buffers belongs to the kernel (path length since
 * the last refresh of the inner list; chx_pack_langevin initialises it to 0).  ctrl[3] must hold INT32_MAX on entry; when the skin
 * check fires while producing the positions of step s+1 the kernel stores s+1 there
 * and every launch with step >= ctrl[3] is a no-op, so the state is exactly the one
 * the reference rebuilds its list from.  do_aoa == 0: force + kicks only (the trailing
 * kick that ends a run).  With p_h->fl == NULL the force is the all-pairs one of the
 * branches without a Verlet list (box.periodic == 0: nbr_list=None, chiron/potential.py:235-258;
 * periodic: PairListNsqrd, chiron/neighbors.py:1106-1216; cutoff < 0: none) in the same single
 * launch per step; those branches have no skin check, ctrl is not touched.  Asynchronous. */
int chx_langevin_fused_chunk(const chx_langevin_t* p_h, float* x_even, float* x_odd, int64_t step0,
                             int32_t n_steps, int32_t kicks_first, int32_t do_aoa,
                             chx_stream_t stream);

/* The same launches as chx_langevin_fused_chunk(do_aoa = 1) on the force-list path, with a bounded launch queue:
 * step s is enqueued only once fewer than `window` earlier step launches are still waiting to begin, and the call
 * stops enqueuing at the first step >= the halt word -- launches the skin check (chiron/integrators.py:239-241) has
 * already turned into no-ops.  *n_enqueued_out = launches made (n_steps unless it stopped early).  The host side polls
 * p_h->host_ctrl, which is required and must be readable by the host at the same address (hipHostMalloc'ed memory):
 * every step launch stores the control block there when it begins, word 0 replaced by its step index and word 3 the
 * halt word (a halting thread also stores word 3 directly).  Before the first call after a chx_ctrl_reset the
 * caller sets host_ctrl[0] = step0 - 1 and host_ctrl[3] = INT32_MAX.  Returns when the launches are enqueued (at
 * most `window` of them not yet begun), not when they have run: a halt raised by the last few launches is for the
 * next call, or for the caller's wait on the stream, to find.  Why: after a halt everything still queued is dead
 * time in front of the list rebuild (about 2 us per no-op launch), so the queue should be as short as the host's
 * launch latency allows rather than a fixed number of chunks.
 * Side effect on the CALLING THREAD: with the sleeping poll policy (the default, chx_set_poll_sleep_ns) the first call
 * lowers the thread's timer slack to 1 us (prctl(PR_SET_TIMERSLACK); a 5 us nap is otherwise rounded up to the
 * kernel's default 50 us) and leaves it there -- the embedding application's thread keeps that slack.
 * CHX_POLL_TIMERSLACK=0 in the environment leaves the slack alone (naps then last ~55 us: use a window >= 24). */
int chx_langevin_fused_run(const chx_langevin_t* p_h, float* x_even, float* x_odd, int64_t step0,
                           int32_t n_steps, int32_t kicks_first, int32_t window,
                           int32_t* n_enqueued_out, chx_stream_t stream);

/* SEVERAL replicas, one launch per step (ABI 5).  The reference advances the replicas of a multistate simulation one after
 * the other (chiron/multistate.py:497-510); a step of a medium-size replica is a latency chain that leaves most of the card
 * idle, so here launch i of a call advances EVERY replica r with i < n_steps[r] by one step -- replica r's step
 * step0[r] + i, reading (i even ? x_even[r] : x_odd[r]) and writing the other buffer, kicks_first[r] half-kicks in its
 * first launch -- in one dispatch (one packet per kind of launch when the replicas' launches differ in kind, side by side).
 * Each replica keeps its own parameter block (box, temperature, key, lists, control block: exactly what
 * chx_langevin_fused_run takes; snap4 must be NULL, host_ctrl is required) and its own halt word: a replica whose skin
 * check has fired turns into a no-op while the others carry on.  Every replica's arithmetic is the one-replica launch's,
 * instruction for instruction: a trajectory does not depend on whether its replica is stepped alone or in a batch.
 * chx_langevin_batch_run keeps at most `window` launches queued ahead of the device (counted on the first participating
 * replica's host_ctrl) and RETURNS at the first halt word it sees among the participants -- n_enqueued[r] = launches in
 * which replica r took part, the same i for all that did --, so that the caller can rebuild that replica's list behind
 * the no-ops and call again (with per-replica step0 / n_steps: replicas drift apart by the launches they sat out).
 * chx_langevin_batch_kick: the force-only launch that ends a run (chx_langevin_fused_chunk with do_aoa = 0), for every item
 * at step0[r], in one dispatch; p->final_x3 / final_v3 as there.  Lanes per particle g in {1, 2, 4, 8}. */
typedef struct {
  const chx_langevin_t* p;
  float* x_even;
  float* x_odd;
  int64_t step0;
  int32_t n_steps;     /* launches wanted for this replica in this call (batch_kick: ignored, one launch) */
  int32_t kicks_first;
  int32_t n_enqueued;  /* out */
  int32_t reserved;
} chx_batch_item_t;
int chx_langevin_batch_run(chx_batch_item_t* items_h, int32_t n_items, int32_t window, chx_stream_t stream);
int chx_langevin_batch_kick(chx_batch_item_t* items_h, int32_t n_items, chx_stream_t stream);
int64_t chx_sizeof_batch_item_t(void);

/* The step queue.  The fused step kernels -- and, while the queue is open, the kernels of the queue-capable entry points
 * (chx_steps_open) -- are dispatched as hand-written AQL kernel-dispatch packets on a user-mode HSA
 * queue of the library's own, one per `stream` value, instead of through hipLaunchKernelGGL: the host side of a dispatch
 * is a copy of the 320-byte argument block into device memory, a 64-byte packet and a doorbell write (the HIP runtime's
 * launch path, and its slow regime, are out of the loop), and the packets of steady-state launches carry no cache
 * fences (their outputs are write-through stores; CHX_AQL_FENCE=agent restores HIP's agent-scope acquire / release on
 * every packet).  Ordering: the first packet after a join waits (barrier-AND packet on a signal that a marker kernel on
 * `stream` sets; no marker when hipStreamQuery finds the stream idle) for everything enqueued on `stream` before it; chx_steps_join waits ON THE HOST until every packet
 * dispatched for `stream` has completed (returns at once when none is outstanding) -- by a barrier packet with a completion
 * signal and a system-scope release, unless the newest packet is a run's LAST launch (chx_langevin_fused_chunk with
 * do_aoa = 0 and p_h->final_x3 set: the launch that also writes the state in user order), which carries both itself.
 * Every libchx entry point that launches on a stream joins first, so libchx calls always see one ordered stream; only work the CALLER enqueues on
 * `stream` by other means (its own kernels, copies, events) needs an explicit chx_steps_join in front when it depends on
 * the steps.  The kernels are loaded from libchx_step.hsaco (required) and libchx_k_*.hsaco (one per translation unit
 * whose kernels may ride on the queue; optional) beside the library (same sources, same flags as the copies inside it).
 * When the queue cannot be set up (file missing, HSA runtime not found, CHX_AQL=0) the launches go through HIP as in
 * ABI 3; chx_step_queue_status() says why. */
int chx_steps_join(chx_stream_t stream);
/* Open the queue for `stream` without dispatching a step: the gate goes up behind everything enqueued on `stream` so far,
 * and from here to the next chx_steps_join the queue-capable entry points -- chx_unpack_langevin, chx_permute_unpack,
 * chx_cell_bin_fl, chx_nl_rows_cell_fl, chx_nl_build_cell_fl, chx_pack_langevin, chx_pack_validate_langevin,
 * chx_validate_snap (chx_mc_displace_chunk opens the queue by itself) -- write their launches as packets behind it instead of launching on the stream (they do so whenever
 * packets are outstanding: that is how a list rebuild inside a run stays behind the steps without a join).  A no-op when
 * the queue is off. */
int chx_steps_open(chx_stream_t stream);
/* out8_h = {queue usable (0 / 1), release scope of a steady-state packet (0 none, 1 agent, -1 off), kernel names in the
 * code objects, packets dispatched for `stream`, joins, gates (stream -> queue hand-overs through a marker launch),
 * packets outstanding (0 / 1), openings on an idle stream (no marker needed: hipStreamQuery said so)}. */
int chx_step_queue_info(int64_t* out8_h, chx_stream_t stream);
const char* chx_step_queue_status(void);
/* Run the start-up check of the fence-free packets AGAIN, now, on the queue paired with `stream` (joined first): 256 rounds
 * of rewrite (write-through stores, one workgroup) / re-read (1024 workgroups, plain vector and scalar loads) with
 * acquire = release = NONE on every packet, about a millisecond.  Returns the number of stale words read (0 = the
 * assumption holds on this card now; > 0: every later packet of the process carries agent-scope fences), -1 when the
 * queue is off, already fenced, or the check's kernels are not loaded.  The first queue of a process runs it once by
 * itself, and every queue repeats it by itself every 2^20 packets, at a join (a millisecond per ~5 s of stepping); a
 * caller may ask at any other time (bench.py does, at its end). */
int chx_step_queue_selfcheck(chx_stream_t stream);
int64_t chx_step_queue_selfcheck_runs(void); /* times the check has run in this process (start-up, periodic, on demand) */
/* Debug mode of the step queue (ABI 5; CHX_QUEUE_DEBUG=<microseconds> in the environment sets the initial value).  With a
 * delay > 0 the queue puts a packet that idles that long (one lane watching the wall clock; at most a second) in front of
 * the first packet behind every gate, of every packet that is not a fused step / Monte Carlo trial launch (unpack, the
 * list build chain, pack, prologues) and of a run's final launch: the launches whose outputs are read from OUTSIDE the
 * queue.  A consumer that has skipped chx_steps_join then reads those outputs before they are written EVERY time; the
 * Python layer completes the detector by filling such outputs with NaN / -1 when it allocates them (chiron_amd.ops.fresh).
 * Results are unchanged (tests/test_gpu_step_queue.py runs the trajectory hashes under it; profiles/r06_queue_debug.md holds the
 * whole GPU suite's log under CHX_QUEUE_DEBUG=250).  delay_us < 0 only reads.  Returns the previous delay.
 * chx_queue_debug_spins: idle packets dispatched so far (a test that sees 0 did not run in debug mode). */
int chx_queue_debug(int32_t delay_us);
int64_t chx_queue_debug_spins(void);

/* How chx_langevin_fused_run waits when its launch queue is full: sleep_ns > 0 -- nanosleep that long between polls
 * (default 5000, or CHX_POLL_SLEEP_NS; the queue refills in bursts of window / 2 launches); 0 -- busy-wait
 * (CHX_POLL=spin).  A PROCESS-GLOBAL host-side setting (every thread that launches through chx_langevin_fused_run
 * sees a change at its next full queue); returns the previous value. */
int64_t chx_set_poll_sleep_ns(int64_t sleep_ns);
/* Diagnostics of the sleeping window since the last reset: out4_h = {waits on a full queue, naps, wake-ups that found
 * two or fewer launches still queued (the nap outlasted the queue), nanoseconds spent in nanosleep}. */
int chx_poll_stats(int64_t* out4_h, int32_t reset);

/* Control-block plumbing of the loop above, as ordinary kernels on the caller's stream:
 * chx_ctrl_reset writes {0, 0, 0, INT32_MAX, -2, -2, 0, 0,  0, -1, -1, -1, INT32_MAX, 0, 0, 0}; chx_publish copies n <= 64 ints into
 * HOST-MAPPED pinned memory (host_mapped_dst is the device-visible address of a
 * hipHostMalloc'ed block) with system-scope visibility, so the host can read the block
 * after waiting for the stream without any copy-engine / blit work in the hot loop. */
int chx_ctrl_reset(int32_t* ctrl, chx_stream_t stream);
int chx_publish(const int32_t* src, int32_t* host_mapped_dst, int32_t n, chx_stream_t stream);

/* ------------------------------------------------------------ Monte Carlo */
/* MonteCarloDisplacementMove proposal: x' = wrap(x + sigma_d * N(0,1) [* mask_i]),
 * chiron/mcmc.py:724-755.  Philox stream 2, counter word 1 = ctr. */
int chx_mc_displace(const float* x, float* x_out, const float* subset_mask, int32_t n,
                    float sigma_d, chx_box_t box, uint32_t k0, uint32_t k1, uint32_t ctr,
                    const float* normals, chx_stream_t stream);

/* MonteCarloBarostatMove geometry: out = in * s, chiron/mcmc.py:977-983 */
int chx_scale(const float* in, float* out, int64_t count, float s, chx_stream_t stream);

/* Metropolis accept/select on the device (no host sync on the accept branch):
 * accept iff !isnan(u_prop) and (-logp <= 0 or U < exp(logp)) with
 * logp = log_ratio[0]; on accept x_cur <- x_prop (count floats), u_cur <- u_prop;
 * counters[0] += accepted, counters[1] += 1.  chiron/mcmc.py:428-436, :531-548.
 * uniform_h < 0: draw U from Philox stream 3 (k0,k1,ctr). */
int chx_metropolis_select(float* x_cur, const float* x_prop, int64_t count,
                          const float* log_ratio, float* u_cur, const float* u_prop,
                          uint32_t k0, uint32_t k1, uint32_t ctr, float uniform_h,
                          int32_t* accepted_out, int32_t* counters, chx_stream_t stream);

/* Device-resident MonteCarloDisplacementMove loop on the force list: n_trials consecutive propose / energy /
 * Metropolis trials (chiron/mcmc.py:243-306, :357-463, :680-787) enqueued back to back, ONE launch per trial (ABI 3;
 * two until round 3) and no host round trip.  Launch t
 *   decides   trial t-1 at its head -- every workgroup, redundantly and identically, from the per-block energy partials
 *             launch t-1 left: U' = fp32(sum partials); accept iff !isnan(beta U') and (beta U' - beta U <= 0 or
 *             uniform < exp(-(beta U' - beta U))) with the uniform drawn from the acceptance subkey of the chain
 *             (:428-436, :531-548); a launch of its own would cost more than the arithmetic (the chunk's last decision
 *             is a one-block launch);
 *   evaluates the energy of the proposal x'_t over the force list (the pair-force kernel's row loop and reduction:
 *             per-block fp64 partials, fixed order) and checks x'_t against the skin (chiron/neighbors.py:868);
 *   proposes  trial t+1: the chain key advances by two splits per trial whatever is decided, so every particle's next
 *             displacement sigma_d N(0,1) [* mask] is known; the launch writes BOTH candidates wrap(x_t + d) and
 *             wrap(x'_t + d) next to the pair (x_t, x'_t), and launch t+1 picks by its decision.  The proposal key is
 *             the subkey of split(chain key) -- what SamplerState.new_PRNG_key would hand out.
 * For the same reason the key chain is worked out on the HOST, by this call, from (k0, k1) = the chain key before trial0
 * (chx_mc_key_chain_h): one row per trial -- proposal key, uniform of the decision -- goes into keytab_h (host-mapped or
 * pinned, 4 (n_trials + 1) uint32) and from there to keytab (device, same size) with an asynchronous copy on `stream`;
 * chain_out_h (2 (n_trials + 1) uint32, may be NULL) receives the chain key BEFORE each trial trial0 + i, i = 0 ..
 * n_trials, so the caller knows the key at the halt trial or at the chunk's end.  The KEY words of the state slots are
 * NOT maintained by the kernels (ABI 3).  keytab_h must not be rewritten before the chunk's launches have run.
 * cur4 and prop4 are (2, 2, n, 4) floats each: [parity][copy][particle]; launch t reads parity t & 1 and writes the
 * other.  On entry copy state[CHX_MC_CUR] of cur4[trial0 & 1] holds the current configuration (internal order); after
 * the chunk copy state[CHX_MC_CUR] (of slot T & 1) of cur4[T & 1] does, T = the halt trial if the chunk halted, else
 * trial0 + n_trials.  state[CHX_MC_REFRESH] of BOTH slots belongs to the kernels (inner list, chx_mc_t::fli: launch t
 * reads the word of slot t & 1 and books the next refresh into the other slot's, like every other state word).
 * When the skin check fires in trial t the launch stores t in state[CHX_MC_HALT] and every launch of trial > t is a
 * no-op (trial t is never decided): the caller runs that trial through the rebuild path and resumes.  state is
 * CHX_MC_STATE_WORDS ints: TWO slots of CHX_MC_SLOT_WORDS words (indices below, slot s at offset
 * s * CHX_MC_SLOT_WORDS) that take turns by trial parity -- a launch reads one and writes the other, so no
 * workgroup reads what another one of the same launch writes.  On entry slot (trial0 & 1) holds the chain state; after
 * the chunk slot (T & 1) does.  The halt word is shared: always state[CHX_MC_HALT] of slot 0.  energy follows the same
 * rule: [2 s] = current potential energy (kJ/mol) of slot s, [2 s + 1] = last proposed.  accept_log / energy_log
 * (n_trials entries each, may be NULL) record the decision and the current energy after every trial, indexed by
 * trial - trial0.  ws: 2 * chx_lj_ws_count(n) doubles (the partials alternate between two halves).  Asynchronous; the
 * launches ride on the step queue (ABI 4: chx_steps_join before the caller's own work on `stream` reads the results). */
#define CHX_MC_CUR 0       /* which copy (0 / 1) of cur4[parity] holds the current configuration */
#define CHX_MC_ACCEPTED 1  /* running counters */
#define CHX_MC_PROPOSED 2
#define CHX_MC_HALT 3      /* first trial whose proposal violated the skin; INT32_MAX when clear */
#define CHX_MC_KEY0 4      /* PRNG chain key */
#define CHX_MC_KEY1 5
#define CHX_MC_REFRESH 6   /* slot t & 1: == t when the launch of trial t renews the inner list (chx_mc_t::fli) */
#define CHX_MC_REFRESHES 7 /* slot 0 only: launches that have renewed it so far (statistics) */
#define CHX_MC_SLOT_WORDS 8
#define CHX_MC_STATE_WORDS 16
typedef struct chx_mc_t {
  int32_t n;
  int32_t g;               /* lanes per particle of the force list */
  int32_t fl_chunks;
  int32_t reserved;
  float* cur4;             /* (2,2,n,4) configurations, internal order (see above) */
  float* prop4;            /* (2,2,n,4) proposal candidates */
  const float* ref4;       /* (n,4) positions the list was built from, internal order */
  const uint32_t* perm;    /* internal -> original id */
  const float* subset_mask;/* per ORIGINAL id 0/1 factors, or NULL */
  const uint32_t* fl;
  const int32_t* deg;
  chx_box_t box;
  float skin;
  float sigma_d;           /* displacement sigma (nm) */
  float sigma, epsilon, cutoff;
  float inner_skin;        /* nm; > 0 with fli != NULL: the trial launches walk an inner list (below) */
  double beta;             /* mol/kJ */
  int32_t* state;          /* CHX_MC_STATE_WORDS ints (two slots, see above) */
  double* energy;          /* 4 doubles: [2 s] current, [2 s + 1] proposed, of slot s */
  int32_t* accept_log;
  float* energy_log;
  double* ws;              /* 2 * chx_lj_ws_count(n) doubles */
  uint32_t k0, k1;         /* chain key before trial0 */
  uint32_t* keytab;        /* device, 4 (n_trials + 1) words */
  uint32_t* keytab_h;      /* pinned host staging of the same size, filled by the call */
  uint32_t* chain_out_h;   /* host, 2 (n_trials + 1) words, or NULL */
  /* Inner list (optional workspace, contents private to the kernels; NULL = off).  Most entries of the force list lie
   * between cutoff and cutoff + skin and add exact zeros to a trial's energy: the chunk's first trial launch (and later
   * ones on demand) copies the entries within cutoff + inner_skin of its proposal into fli, and the launches after it walk
   * those until a particle has strayed inner_skin / 2 from where it stood then (every launch checks the candidates it
   * writes for the next trial, which then renews the list).  Compacted per lane and per accumulator, so the energies --
   * and with them every decision -- are the force list's, bit for bit. */
  uint32_t* fli;           /* as many words as fl */
  int32_t* cnti;           /* n * g ints */
  float* refi4;            /* (n,4) */
} chx_mc_t;
int chx_mc_displace_chunk(const chx_mc_t* p_h, int64_t trial0, int32_t n_trials,
                          chx_stream_t stream);
/* The key chain of n_trials consecutive trials from the chain key (k0, k1), on the host: per trial two splits
 * (chiron/mcmc.py:724 proposal, :531 acceptance).  keytab_h row i (4 words) = {proposal key of trial i (2), fp32 bits of
 * the uniform of its decision, fp32 bits of that uniform's logarithm}, rows 0 .. n_trials; chain_out_h entry i (2 words) = chain key before trial i, entries
 * 0 .. n_trials.  Either may be NULL. */
int chx_mc_key_chain_h(uint32_t k0, uint32_t k1, int32_t n_trials, uint32_t* keytab_h, uint32_t* chain_out_h);

#ifdef __cplusplus
}
#endif
#endif /* CHX_H */
