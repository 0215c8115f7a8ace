// Host-side memory-safety check of libchx's C ABI: this driver is linked against a
// host-only AddressSanitizer + UBSan build of the library's own sources (scripts/build_host_asan.sh:
// `hipcc --offload-host-only -fsanitize=address,undefined`) and exercises everything that runs on the host
// before a kernel is launched: the pure host helpers over wide input ranges, and the argument validation of
// every entry point (which must return CHX_ERR_INVALID, set chx_last_error() and touch nothing).  No GPU is
// needed or used: no call below reaches a kernel launch.  Exit code 0 = all checks passed and no sanitizer report.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/chx.h"

static int g_fail = 0;
#define EXPECT(cond)                                                       \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
      ++g_fail;                                                            \
    }                                                                      \
  } while (0)
#define EXPECT_INVALID(call)                                               \
  do {                                                                     \
    const int rc_ = (call);                                                \
    if (rc_ != CHX_ERR_INVALID || strlen(chx_last_error()) == 0) {         \
      fprintf(stderr, "FAIL %s:%d: %s -> %d\n", __FILE__, __LINE__, #call, rc_); \
      ++g_fail;                                                            \
    }                                                                      \
  } while (0)

int main() {
  EXPECT(chx_version() == CHX_ABI_VERSION);
  EXPECT(strlen(chx_source_hash()) > 0);
  EXPECT(chx_sizeof_langevin_t() == (int64_t)sizeof(chx_langevin_t));
  EXPECT(chx_sizeof_mc_t() == (int64_t)sizeof(chx_mc_t));

  // ---- chx_cutoff_threshold_h: T = max{t : sqrtf(t) < c}
  const float cs[] = {1e-30f, 1e-6f, 0.25f, 0.5f, 1.0f, 1.02f, 1.2f, 1.52f, 3.0f, 46.8805f, 1e4f, 1e18f};
  for (float c : cs) {
    const float t = chx_cutoff_threshold_h(c);
    EXPECT(sqrtf(t) < c);
    EXPECT(!(sqrtf(nextafterf(t, INFINITY)) < c));
  }
  EXPECT(chx_cutoff_threshold_h(0.0f) == -1.0f);
  EXPECT(chx_cutoff_threshold_h(-3.0f) == -1.0f);
  EXPECT(chx_cutoff_threshold_h(NAN) == -1.0f);
  EXPECT(isinf(chx_cutoff_threshold_h(INFINITY)));
  for (int i = 0; i < 20000; ++i) {  // dense sweep around typical cutoffs
    const float c = 0.05f + 1e-4f * (float)i;
    const float t = chx_cutoff_threshold_h(c);
    EXPECT(sqrtf(t) < c && !(sqrtf(nextafterf(t, INFINITY)) < c));
  }

  // ---- the step queue without a GPU: nothing to join, nothing to open, the queue reports itself off with a reason
  EXPECT(chx_steps_join(nullptr) == CHX_OK);
  EXPECT(chx_steps_open(nullptr) == CHX_OK);
  {
    int64_t info[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    EXPECT(chx_step_queue_info(info, nullptr) == CHX_OK);
    EXPECT(info[0] == 0 && info[1] == -1 && info[3] == 0 && info[6] == 0);
    EXPECT(chx_step_queue_info(nullptr, nullptr) == CHX_OK);
    EXPECT(chx_step_queue_status() != nullptr && strlen(chx_step_queue_status()) > 0);
    EXPECT(chx_steps_join(nullptr) == CHX_OK);
  }

  // ---- chx_cell_grid_h
  int32_t nc[3] = {-7, -7, -7};
  chx_box_t open_box = {0, 0.f, 0.f, 0.f};
  EXPECT(chx_cell_grid_h(open_box, 1.52f, 1000, nc) == 0 && nc[0] == 0);
  EXPECT(chx_cell_grid_h(open_box, 1.52f, 1000, nullptr) == 0);
  const float Ls[] = {1e-3f, 3.0f, 4.56f, 4.5602f, 5.86f, 23.4402f, 46.8805f, 1000.f, 3e5f};
  for (float L : Ls) {
    for (int n : {1, 8, 512, 32768, 262144}) {
      chx_box_t b = {1, L, L * 1.3f, L * 0.9f};
      const int64_t cells = chx_cell_grid_h(b, 1.52f, n, nc);
      if (cells) {
        EXPECT(nc[0] >= 3 && nc[1] >= 3 && nc[2] >= 3 && nc[0] <= 1024 && nc[1] <= 1024 && nc[2] <= 1024);
        EXPECT(cells == (int64_t)nc[0] * nc[1] * nc[2]);
        EXPECT(b.lx / nc[0] >= 1.52f && b.ly / nc[1] >= 1.52f && b.lz / nc[2] >= 1.52f);  // edge >= cutoff + skin
      } else {
        EXPECT(nc[0] == 0 || nc[0] >= 0);
      }
    }
  }
  chx_box_t bad_box = {1, 10.f, -1.f, 10.f};
  EXPECT(chx_cell_grid_h(bad_box, 1.52f, 100, nc) == 0);
  chx_box_t nan_box = {1, NAN, 10.f, 10.f};
  EXPECT(chx_cell_grid_h(nan_box, 1.52f, 100, nc) == 0);
  chx_box_t ok_box = {1, 10.f, 10.f, 10.f};
  EXPECT(chx_cell_grid_h(ok_box, 0.f, 100, nc) == 0);
  EXPECT(chx_cell_grid_h(ok_box, NAN, 100, nc) == 0);
  EXPECT(chx_cell_grid_h(ok_box, 1.52f, 0, nc) == 0);

  // ---- workspace sizing
  EXPECT(chx_lj_ws_count(0) >= 1 && chx_lj_ws_count(-5) >= 1);
  EXPECT(chx_lj_ws_count(262144) >= 262144 / 4);
  EXPECT(chx_lj_ws_count(INT32_MAX) > 0);

  // ---- argument validation of every entry point: invalid sizes / null pointers never reach a launch
  std::vector<float> f(64, 0.f);
  std::vector<uint32_t> u(64, 0u);
  std::vector<int32_t> i32(64, 0);
  std::vector<double> d(64, 0.0);
  std::vector<uint8_t> u8(64, 0);
  float* F = f.data();
  uint32_t* U = u.data();
  int32_t* I = i32.data();
  EXPECT_INVALID(chx_displacement(F, F, -1, ok_box, F, F, nullptr));
  EXPECT(chx_displacement(nullptr, nullptr, 0, ok_box, nullptr, nullptr, nullptr) == CHX_OK);
  EXPECT_INVALID(chx_wrap(F, F, -1, ok_box, nullptr));
  EXPECT(chx_wrap(nullptr, nullptr, 0, ok_box, nullptr) == CHX_OK);
  EXPECT_INVALID(chx_nl_build_nsq(F, -1, 4, ok_box, 1.5f, U, I, I, nullptr));
  EXPECT_INVALID(chx_nl_build_nsq(F, 4, 0, ok_box, 1.5f, U, I, I, nullptr));
  EXPECT_INVALID(chx_nl_build_nsq(nullptr, 4, 4, ok_box, 1.5f, U, I, I, nullptr));
  EXPECT_INVALID(chx_nl_build_nsq(F, 4, 4, bad_box, 1.5f, U, I, I, nullptr));
  EXPECT(chx_nl_build_nsq(nullptr, 0, 4, ok_box, 1.5f, nullptr, nullptr, nullptr, nullptr) == CHX_OK);
  int32_t grid[3] = {3, 3, 3};
  EXPECT_INVALID(chx_nl_build_cell(F, -1, 4, ok_box, 1.5f, grid, I, I, I, U, I, U, I, I, nullptr));
  EXPECT_INVALID(chx_nl_build_cell(nullptr, 4, 4, ok_box, 1.5f, grid, I, I, I, U, I, U, I, I, nullptr));
  EXPECT_INVALID(chx_nl_build_cell_fl(F, -1, 4, ok_box, 1.5f, grid, I, I, I, U, I, U, U, I, 8, 2, U, I, F, nullptr, nullptr));
  EXPECT_INVALID(chx_nl_build_cell_fl(F, 4, 4, ok_box, 1.5f, grid, I, I, I, U, I, U, U, I, 3, 2, U, I, F, nullptr, nullptr));  // g not 2^k
  EXPECT_INVALID(chx_nl_finish_padding(U, U, I, -1, 4, I, nullptr));
  EXPECT_INVALID(chx_nl_calculate(F, U, I, -1, 4, ok_box, 1.0f, I, I, F, F, nullptr));
  EXPECT_INVALID(chx_nl_check(F, F, -1, ok_box, 0.5f, I, nullptr));
  EXPECT_INVALID(chx_pl_build(-1, U, u8.data(), nullptr));
  EXPECT_INVALID(chx_pl_calculate(F, -1, ok_box, 1.0f, I, I, F, F, nullptr));
  EXPECT_INVALID(chx_fl_count(U, I, -1, 4, nullptr, I, I, nullptr));
  EXPECT_INVALID(chx_fl_fill(U, I, -1, 4, nullptr, 8, I, U, nullptr));
  EXPECT_INVALID(chx_fl_sort(U, I, -1, 8, 4, nullptr));
  EXPECT_INVALID(chx_permute_pack(F, nullptr, -1, F, 4, nullptr));
  EXPECT_INVALID(chx_permute_unpack(F, 4, nullptr, -1, F, nullptr));
  EXPECT_INVALID(chx_invert_perm(U, -1, U, nullptr));
  EXPECT_INVALID(chx_pack_langevin(F, F, F, F, U, -1, F, F, F, F, nullptr, 0, nullptr, nullptr, nullptr));
  EXPECT_INVALID(chx_pack_langevin(F, F, F, F, U, 4, F, F, F, F, nullptr, 0, I, nullptr, nullptr));  // publish without a destination
  // round 4: the pack launch that also checks a carried-over list, the build's optional workspaces, the key table
  EXPECT_INVALID(chx_pack_validate_langevin(ok_box, 0.4f, 0, F, nullptr, I, F, F, F, F, U, 0, F, F, F, F, I, 0, nullptr));      // n < 1
  EXPECT_INVALID(chx_pack_validate_langevin(ok_box, 0.4f, 0, nullptr, nullptr, I, F, F, F, F, U, 4, F, F, F, F, I, 0, nullptr)); // no snap4
  EXPECT_INVALID(chx_pack_validate_langevin(ok_box, 0.4f, 0, F, nullptr, nullptr, F, F, F, F, U, 4, F, F, F, F, I, 0, nullptr)); // no sync2
  EXPECT_INVALID(chx_pack_validate_langevin(ok_box, 0.4f, 0, F, nullptr, I, F, F, F, F, U, 4, F, F, F, F, nullptr, 0, nullptr)); // no ctrl
  EXPECT_INVALID(chx_pack_validate_langevin(ok_box, 0.0f, 0, F, nullptr, I, F, F, F, F, U, 4, F, F, F, F, I, 0, nullptr));       // skin
  EXPECT_INVALID(chx_pack_validate_langevin(ok_box, 0.4f, -1, F, nullptr, I, F, F, F, F, U, 4, F, F, F, F, I, 0, nullptr));      // step
  EXPECT_INVALID(chx_pack_validate_langevin(ok_box, 0.4f, 0, F, nullptr, I, F, F, F, nullptr, U, 4, F, F, F, F, I, 0, nullptr)); // ref4 without ref3
  EXPECT(chx_sizeof_build_riders_t() == (int64_t)sizeof(chx_build_riders_t));
  {
    chx_build_riders_t rd;
    memset(&rd, 0, sizeof(rd));
    rd.cell_slot_cap = 64;  // a capacity without the slots
    EXPECT_INVALID(chx_cell_bin_fl(F, 4, ok_box, 1.5f, grid, I, I, I, U, I, F, &rd, nullptr));
    rd.cell_slot_cap = -1;
    rd.cell_slots = U;
    EXPECT_INVALID(chx_cell_bin_fl(F, 4, ok_box, 1.5f, grid, I, I, I, U, I, F, &rd, nullptr));
    // half rows: nl_rows_out and first_out go together (both NULL = counted only)
    EXPECT_INVALID(chx_nl_rows_cell_fl(4, 4, ok_box, 1.5f, grid, I, I, U, I, F, nullptr, U, I, 4, 2, U, I, nullptr, nullptr, nullptr, 0, nullptr));
    EXPECT_INVALID(chx_nl_rows_cell_fl(4, 4, ok_box, 1.5f, grid, I, I, U, I, F, U, nullptr, I, 4, 2, U, I, nullptr, nullptr, nullptr, 0, nullptr));
    EXPECT_INVALID(chx_nl_build_cell_fl(F, 4, 4, ok_box, 1.5f, grid, I, I, I, U, I, nullptr, U, I, 4, 2, U, I, F, nullptr, nullptr));
  }
  {
    std::vector<uint32_t> tab(4 * 33), chain(2 * 33);
    EXPECT(chx_mc_key_chain_h(2024u, 7u, 32, tab.data(), chain.data()) == CHX_OK);
    for (int r = 0; r <= 32; ++r) {
      float uf, lf;
      memcpy(&uf, &tab[4 * r + 2], 4);
      memcpy(&lf, &tab[4 * r + 3], 4);
      EXPECT(uf > 0.f && uf < 1.f && lf < 0.f && fabs((double)lf - log((double)uf)) <= 1.2e-7 * fabs(log((double)uf)) + 1e-30);
    }
    EXPECT(chx_mc_key_chain_h(1u, 2u, 0, nullptr, nullptr) == CHX_OK);
    EXPECT_INVALID(chx_mc_key_chain_h(1u, 2u, -1, tab.data(), chain.data()));
  }
  EXPECT_INVALID(chx_unpack_langevin(F, F, U, -1, F, F, nullptr));
  EXPECT_INVALID(chx_unpack_langevin(F, nullptr, U, 4, F, F, nullptr));
  EXPECT(chx_unpack_langevin(nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr) == CHX_OK);
  EXPECT_INVALID(chx_lj_fl_energy_force(F, U, I, -1, 8, 1, ok_box, 0.34f, 1.f, 1.02f, F, F, d.data(), nullptr));
  EXPECT_INVALID(chx_lj_fl_energy_force(F, U, I, 4, 8, 1, ok_box, 0.34f, 1.f, 1.02f, nullptr, nullptr, d.data(), nullptr));
  EXPECT_INVALID(chx_lj_fl_energy_force(F, U, I, 4, 8, 1, ok_box, 0.34f, 1.f, 1.02f, nullptr, F, nullptr, nullptr));
  EXPECT_INVALID(chx_lj_nl_energy_force(F, U, I, -1, 4, ok_box, 0.34f, 1.f, 1.02f, F, F, d.data(), nullptr));
  EXPECT_INVALID(chx_lj_allpairs_energy_force(F, -1, ok_box, 0.34f, 1.f, 1.02f, F, F, d.data(), nullptr));
  EXPECT_INVALID(chx_init_velocities(F, F, -1, 2.5f, 1u, 2u, nullptr, nullptr));
  EXPECT_INVALID(chx_kick(F, F, F, -1, 5e-4f, nullptr));
  EXPECT_INVALID(chx_jax_normal(1u, 2u, -1, F, nullptr));
  EXPECT_INVALID(chx_jax_normal(1u, 2u, (int64_t)1 << 33, F, nullptr));
  EXPECT_INVALID(chx_jax_normal(1u, 2u, 8, nullptr, nullptr));
  EXPECT(chx_jax_normal(1u, 2u, 0, nullptr, nullptr) == CHX_OK);
  EXPECT_INVALID(chx_baoab_step(F, F, F, F, -1, ok_box, 1, 5e-4f, 0.99f, 0.04f, 2.5f, F, 0.5f, 1u, 2u, 0u, nullptr, I, nullptr));
  EXPECT_INVALID(chx_mc_displace(F, F, nullptr, -1, 0.1f, ok_box, 1u, 2u, 0u, nullptr, nullptr));
  EXPECT_INVALID(chx_scale(F, F, -1, 1.1f, nullptr));
  EXPECT_INVALID(chx_metropolis_select(F, F, -1, F, F, F, 1u, 2u, 0u, -1.f, I, I, nullptr));
  EXPECT_INVALID(chx_ctrl_reset(nullptr, nullptr));
  EXPECT_INVALID(chx_publish(I, I, 65, nullptr));
  EXPECT_INVALID(chx_publish(nullptr, I, 4, nullptr));
  // struct-taking entry points: null struct, zeroed struct, bad sizes, unsupported lane count
  EXPECT_INVALID(chx_langevin_chunk(nullptr, 0, 1, 1, nullptr));
  EXPECT_INVALID(chx_langevin_resume(nullptr, nullptr));
  EXPECT_INVALID(chx_langevin_fused_chunk(nullptr, F, F, 0, 1, 1, 1, nullptr));
  chx_langevin_t lp;
  memset(&lp, 0, sizeof(lp));
  lp.n = -3;
  EXPECT_INVALID(chx_langevin_chunk(&lp, 0, 1, 1, nullptr));
  EXPECT_INVALID(chx_langevin_fused_chunk(&lp, F, F, 0, 1, 1, 1, nullptr));
  lp.n = 4;
  EXPECT_INVALID(chx_langevin_fused_chunk(&lp, nullptr, F, 0, 1, 1, 1, nullptr));
  EXPECT_INVALID(chx_langevin_fused_chunk(&lp, F, F, 0, -1, 1, 1, nullptr));
  lp.v = F, lp.mass = F, lp.ctrl = I, lp.fl = U, lp.deg = I, lp.ref4 = F, lp.g = 3, lp.fl_chunks = 1;
  EXPECT_INVALID(chx_langevin_fused_chunk(&lp, F, F, 0, 1, 1, 1, nullptr));  // g = 3 is not a supported lane count
  // the bounded launch queue: needs the force-list path, a host-visible control block, a window and an out pointer
  int32_t sent = -1;
  EXPECT_INVALID(chx_langevin_fused_run(nullptr, F, F, 0, 1, 1, 4, &sent, nullptr));
  EXPECT_INVALID(chx_langevin_fused_run(&lp, F, F, 0, 1, 1, 4, nullptr, nullptr));
  EXPECT_INVALID(chx_langevin_fused_run(&lp, F, F, 0, 1, 1, 4, &sent, nullptr));  // host_ctrl == NULL
  lp.host_ctrl = I;
  EXPECT_INVALID(chx_langevin_fused_run(&lp, F, F, 0, 1, 1, 0, &sent, nullptr));  // window < 1
  EXPECT_INVALID(chx_langevin_fused_run(&lp, F, F, 0, -1, 1, 4, &sent, nullptr));
  {
    // the halt word already rules out the first step: nothing is launched (so the bad lane count is never reached)
    int32_t hc[CHX_CTRL_WORDS] = {6, 0, 0, 7, -2, -2, 0, 0, 0, -1, -1, -1, 0x7FFFFFFF, 0, 0, 0};
    lp.host_ctrl = hc;
    sent = -1;
    int rc = chx_langevin_fused_run(&lp, F, F, 7, 5, 2, 4, &sent, nullptr);
    (void)rc;  // (without a device the trailing hipGetLastError may complain; what matters is that nothing was launched)
    EXPECT(sent == 0);
    // a free window and no halt: the first launch is attempted and rejected for its lane count
    hc[3] = 0x7fffffff;
    EXPECT_INVALID(chx_langevin_fused_run(&lp, F, F, 7, 5, 2, 4, &sent, nullptr));
    lp.host_ctrl = I;
  }
  // several replicas per launch (ABI 5): the argument checks run before anything touches a device
  {
    chx_batch_item_t items[2];
    memset(items, 0, sizeof(items));
    EXPECT_INVALID(chx_langevin_batch_run(nullptr, 1, 4, nullptr));
    EXPECT_INVALID(chx_langevin_batch_run(items, 0, 4, nullptr));
    EXPECT_INVALID(chx_langevin_batch_run(items, 1, 4, nullptr));  // null parameter block
    lp.g = 4;
    items[0].p = &lp, items[0].x_even = F, items[0].x_odd = F, items[0].n_steps = 1, items[0].kicks_first = 1;
    EXPECT_INVALID(chx_langevin_batch_run(items, 1, 4, nullptr));  // one position buffer for both parities
    items[0].x_odd = F + 16;
    items[1] = items[0];
    EXPECT_INVALID(chx_langevin_batch_run(items, 2, 4, nullptr));  // two items on one control block
    EXPECT_INVALID(chx_langevin_batch_run(items, 1, 0, nullptr));  // window < 1
    items[0].kicks_first = 3;
    EXPECT_INVALID(chx_langevin_batch_kick(items, 1, nullptr));
    items[0].kicks_first = 1;
    lp.g = 16;
    EXPECT_INVALID(chx_langevin_batch_run(items, 1, 4, nullptr));  // lane counts above 8 are not instantiated
    lp.g = 3;
    EXPECT(chx_sizeof_batch_item_t() == (int64_t)sizeof(chx_batch_item_t));
  }
  // the queue's debug mode and the on-demand fence check without a GPU: a setting, and "not applicable"
  {
    const int old = chx_queue_debug(-1);
    EXPECT(chx_queue_debug(125) == old && chx_queue_debug(old) == 125);
    EXPECT(chx_queue_debug_spins() == 0);
    EXPECT(chx_step_queue_selfcheck(nullptr) == -1);
    EXPECT(chx_step_queue_selfcheck_runs() == 0);
  }
  EXPECT_INVALID(chx_mc_displace_chunk(nullptr, 0, 1, nullptr));
  chx_mc_t mp;
  memset(&mp, 0, sizeof(mp));
  mp.n = -1;
  EXPECT_INVALID(chx_mc_displace_chunk(&mp, 0, 1, nullptr));
  mp.n = 4;
  EXPECT_INVALID(chx_mc_displace_chunk(&mp, 0, 1, nullptr));  // null buffers
  if (g_fail) {
    fprintf(stderr, "%d host checks failed\n", g_fail);
    return 1;
  }
  printf("host ABI checks passed under ASan/UBSan\n");
  return 0;
}
