// Can the step launches be replayed from hipGraphs whose kernel arguments are UPDATED between replays
// (hipGraphExecKernelNodeSetParams), so that the kernels keep their by-value argument block and nothing about them changes?
// A ring of R executable graphs of L step launches each; before every replay the L nodes get their new step index and buffer
// pair; a graph is reused only after a launch enqueued behind it has begun (progress word in host-mapped memory).  Every
// launch checks on the device that its step index is the one the chain expects -- an update that reached a replay still
// in flight, or a stale argument block, shows up in the mismatch counter.
// Prints per mode: launches per second, host microseconds per replay (updates / launch call), context switches.
//   hipcc --offload-arch=gfx950 -O3 -o graph_update_probe graph_update_probe.hip && ./graph_update_probe [launches] [L] [R]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/resource.h>
#include <time.h>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

struct Big {  // ~ the step kernel's by-value argument block
  const float4* in;
  float4* out;
  int32_t* host_ctrl;
  const void* pad[24];
};

// ctrl[0]: the step the chain expects next; ctrl[1]: launches whose step index was not that
__global__ __launch_bounds__(256) void k_step(int32_t* ctrl, long long step, int n, Big b) {
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&b.host_ctrl[0], (int)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float4 v = b.in[t];
  const int j = ((int)(v.x * 1e3f) & 0x7fffffff) % n;
  const float4 w = b.in[j];
#pragma unroll 1
  for (int k = 0; k < 200; ++k) v.y = v.y * 0.999f + w.x * 1e-3f;
  v.x = w.x * 0.5f + v.x * 0.5f;
  b.out[t] = v;
  if (t == 0) {
    if (ctrl[0] != (int)step) ctrl[1] += 1;
    ctrl[0] = (int)step + 1;
  }
}

static double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char** argv) {
  const int n = 32768 * 4, launches = argc > 1 ? atoi(argv[1]) : 200000, window = 12;
  const int L = argc > 2 ? atoi(argv[2]) : 8, R = argc > 3 ? atoi(argv[3]) : 4;
  float4 *a, *b;
  int32_t *ctrl, *host;
  CK(hipMalloc(&a, 16 * (size_t)n));
  CK(hipMalloc(&b, 16 * (size_t)n));
  CK(hipMalloc(&ctrl, 64));
  CK(hipHostMalloc(&host, 64, hipHostMallocMapped));
  CK(hipMemset(a, 0, 16 * (size_t)n));
  CK(hipMemset(ctrl, 0, 64));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Big big = {};
  big.host_ctrl = host;
  volatile int32_t* hc = host;
  host[0] = -1;
  setvbuf(stdout, nullptr, _IOLBF, 0);

  // ---- eager ----
  auto run_eager = [&](int count, long long s0) {
    for (int i = 0; i < count; ++i) {
      const long long s = s0 + i;
      while (s - hc[0] > window) __builtin_ia32_pause();
      big.in = (s & 1) ? b : a;
      big.out = (s & 1) ? a : b;
      hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, st, ctrl, s, n, big);
    }
  };

  // ---- ring of updated graphs ----
  struct NodeArgs {
    int32_t* ctrl;
    long long step;
    int n;
    Big b;
    void* ptrs[4];
  };
  std::vector<hipGraphExec_t> execs(R);
  std::vector<std::vector<hipGraphNode_t>> nodes(R, std::vector<hipGraphNode_t>(L));
  std::vector<std::vector<NodeArgs>> args(R, std::vector<NodeArgs>(L));
  for (int r = 0; r < R; ++r) {
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    for (int i = 0; i < L; ++i) {
      NodeArgs& na = args[r][i];
      na.ctrl = ctrl;
      na.step = i;
      na.n = n;
      na.b = big;
      na.b.in = (i & 1) ? b : a;
      na.b.out = (i & 1) ? a : b;
      na.ptrs[0] = &na.ctrl, na.ptrs[1] = &na.step, na.ptrs[2] = &na.n, na.ptrs[3] = &na.b;
      hipKernelNodeParams kp = {};
      kp.func = (void*)k_step;
      kp.gridDim = dim3(n / 256);
      kp.blockDim = dim3(256);
      kp.kernelParams = na.ptrs;
      CK(hipGraphAddKernelNode(&nodes[r][i], g, i ? &nodes[r][i - 1] : nullptr, i ? 1 : 0, &kp));
    }
    CK(hipGraphInstantiate(&execs[r], g, nullptr, nullptr, 0));
  }
  hipEvent_t evs[4];
  for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  double t_upd = 0, t_launch = 0;
  long replays_done = 0;
  auto run_graphs = [&](int count, long long s0, bool update) {  // L must be even: a graph starts on an even step
    const int reps = count / L;
    for (int rep = 0; rep < reps; ++rep) {
      const long long s = s0 + (long long)rep * L;
      const int r = (int)((s / L) % R);
      if (update) {
        while (s - hc[0] > window) __builtin_ia32_pause();  // (window < (R - 1) L: the graph's previous replay is over)
      } else if (rep >= 2) {
        CK(hipEventSynchronize(evs[(rep - 2) % 4]));  // stale step indices say nothing about progress: two replays in flight
      }
      const double t0 = now();
      if (update) {
        for (int i = 0; i < L; ++i) {
          NodeArgs& na = args[r][i];
          na.step = s + i;
          hipKernelNodeParams kp = {};
          kp.func = (void*)k_step;
          kp.gridDim = dim3(n / 256);
          kp.blockDim = dim3(256);
          kp.kernelParams = na.ptrs;
          CK(hipGraphExecKernelNodeSetParams(execs[r], nodes[r][i], &kp));
        }
      }
      const double t1 = now();
      CK(hipGraphLaunch(execs[r], st));
      const double t2 = now();
      if (!update) CK(hipEventRecord(evs[rep % 4], st));
      t_upd += t1 - t0, t_launch += t2 - t1, replays_done += 1;
    }
  };

  long long s = 0;
  auto reset_chain = [&]() {
    CK(hipStreamSynchronize(st));
    int32_t z[2] = {(int32_t)s, 0};
    CK(hipMemcpy(ctrl, z, 8, hipMemcpyHostToDevice));
  };
  double t0 = now();
  while (now() - t0 < 3.0) {  // clock warm-up
    run_eager(2000, s);
    s += 2000;
  }
  CK(hipStreamSynchronize(st));
  const int count = launches / L * L;
  for (int rep = 0; rep < 3; ++rep) {
    for (int mode = 0; mode < 3; ++mode) {  // 0 eager, 1 graphs with updated arguments, 2 graphs replayed as they are
      reset_chain();
      t_upd = t_launch = 0, replays_done = 0;
      rusage r0, r1;
      getrusage(RUSAGE_THREAD, &r0);
      const double t1 = now();
      if (mode == 0) run_eager(count, s);
      else run_graphs(count, s, mode == 1);
      CK(hipStreamSynchronize(st));
      const double dt = now() - t1;
      getrusage(RUSAGE_THREAD, &r1);
      int32_t c[2];
      CK(hipMemcpy(c, ctrl, 8, hipMemcpyDeviceToHost));
      if (mode != 2) s += count;
      printf("%-28s %d launches: %.0f per second (%.2f us each), host per replay: update %.2f us + launch %.2f us, mismatched steps %d%s, voluntary context switches %ld, involuntary %ld\n",
             mode == 0 ? "eager" : mode == 1 ? "graph ring, updated args" : "graph ring, stale args", count, count / dt,
             1e6 * dt / count, replays_done ? 1e6 * t_upd / replays_done : 0.0, replays_done ? 1e6 * t_launch / replays_done : 0.0, c[1],
             mode == 2 ? " (expected: every one)" : "", r1.ru_nvcsw - r0.ru_nvcsw, r1.ru_nivcsw - r0.ru_nivcsw);
    }
  }
  return 0;
}
