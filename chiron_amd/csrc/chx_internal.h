// Internal (non-exported) launch helpers shared between translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chx.h"

struct CellGrid {
  int ncx, ncy, ncz;
  float sx, sy, sz;  // nc / L
};

// Riders of the binning pass (all optional; the integrator's rebuild chain: one launch sequence instead of
// copy / zero / pack launches of their own):
//   xyz_copy  (n,3): a copy of the input positions (the list's private reference positions);
//   v3 .. ref4: the Langevin state packed into the cell order the binning produces (what chx_pack_langevin does):
//               x4[p] = (xyz[id], 0), v[p] = v3[id], mass_out[p] = mass[id], ref4[p] = ((ref3 ? ref3 : xyz)[id], mass[id])
//               with id = sorted_ids[p]; on when x4 != NULL;
//   ctrl:      what chx_ctrl_reset writes (ctrl_keep as in chx_pack_langevin).
struct BinRiders {
  float* xyz_copy;
  const float* v3;
  const float* mass;
  const float* ref3;
  float4* x4;
  float* v;
  float* mass_out;
  float4* ref4;
  int32_t* ctrl;
  int ctrl_keep;
  // per-cell id slots (include/chx.h: chx_build_riders_t::cell_slots): the counting launch stores every particle's id in
  // slot (cell, arrival order) and the scatter launch is not needed
  uint32_t* cell_slots;
  int cell_slot_cap;
};
// Bin particles into cells (chx_neighbors.hip): cell_of, cell_start, sorted_ids [, xs4_out]; zeroes status.
// cell_count must be all zero on entry (ncell + 1 words) and is all zero again when the chain has run.
int chx_cell_prepare(const float* xyz, int n, chx_box_t box, float cutoff_plus_skin,
                     const int32_t* nc_h, int32_t* cell_of, int32_t* cell_count, int32_t* cell_start,
                     uint32_t* sorted_ids, int32_t* status, int status_len, CellGrid* g_out,
                     hipStream_t st, float* xs4_out = nullptr,  // xs4_out: positions in cell order, .w = cell index
                     const BinRiders* riders = nullptr,
                     // optional per-cell candidate index lists (chx_cell.h): ncell * cand_cap words and ncell lengths
                     uint32_t* cand_list = nullptr, int32_t* cand_len = nullptr, int cand_cap = 0);

// ctrl != NULL: the kernel is part of a Langevin chunk and turns into a no-op
// (setting ctrl[1]) when ctrl[0] says the skin check fired.
int chx_launch_lj_fl(const float* x4, const uint32_t* fl, const int32_t* deg, int n, int g, int chunks,
                     chx_box_t box, float sigma, float epsilon, float cutoff, float* f3,
                     float* energy, double* ws, int32_t* ctrl, hipStream_t st);
// Force-list addressing: a row is stored in chunks of W = CHX_FL_QUAD * g entries, fl[chunk][particle][W], and lane q
// of the g lanes that own a particle reads the four words 4q..4q+3 of a chunk with one 16-byte load.  Inside a chunk the
// row is INTERLEAVED over the lanes: row entry e (0 <= e < W, in the row's order) sits in word 4 (e % g) + e / g, so
// word u of lane q is row entry u g + q and the u-th gather instruction of a group fetches g CONSECUTIVE row entries --
// neighbors that are adjacent in the (cell-ordered) position array and mostly share 128-byte lines -- instead of
// entries 4 apart from each other: at liquid density the gathers are bound by the L1's line-lookup rate.
#define CHX_FL_QUAD 4
__host__ __device__ __forceinline__ int chx_fl_word(int e, int g) {  // row entry within a chunk -> word within the chunk
  const int sh = 31 - __builtin_clz((unsigned)g);                    // g is a power of two
  return CHX_FL_QUAD * (e & (g - 1)) + (e >> sh);
}
__device__ __forceinline__ size_t chx_fl_addr(int p, int k, int n, int g) {
  const int sh = 31 - __clz(CHX_FL_QUAD * g);
  return ((((size_t)(k >> sh)) * (size_t)n + (size_t)p) << sh) + (size_t)chx_fl_word(k & ((1 << sh) - 1), g);
}
// number of per-block partials the list kernel leaves in ws for (n, g)
static inline int chx_lj_fl_blocks(int n, int g) { return (int)(((int64_t)n * g + 255) / 256); }
// n_steps fused [force + BAOAB] launches on the force list (k_lj_step); launch i reads
// (i even ? x_even : x_odd) and writes the other buffer.  do_aoa == 0: force + kicks only.
int chx_launch_lj_steps(const chx_langevin_t* p, float* x_even, float* x_odd, int64_t step0,
                        int n_steps, int kicks_first, int do_aoa, hipStream_t st, int window = 0,
                        int32_t* n_enqueued = nullptr);
// the same for several replicas at once, one dispatch per step (chx_launch_window.hip)
int chx_launch_lj_steps_batch(chx_batch_item_t* items, int n_items, int do_aoa, int window, hipStream_t st);
int chx_launch_lj_allpairs(const float* x, int stride, int n, chx_box_t box, float sigma,
                           float epsilon, float cutoff, float* f3, float* energy, double* ws,
                           int32_t* ctrl, hipStream_t st);
