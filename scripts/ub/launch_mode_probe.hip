// Does a BARE HIP program (no Python, no torch) show the "slow process" regime of the Langevin loop -- thousands of
// voluntary context switches in the launching thread, a busy runtime helper thread, 3-12 % fewer launches per second?
// A ~4.5 us dependent kernel chain launched eagerly behind a 12-deep window that polls a host-mapped progress word,
// exactly like chx_langevin_fused_run.  Prints launches/s and the thread's context switches; run it a dozen times.
//   hipcc --offload-arch=gfx950 -O3 -o launch_mode_probe launch_mode_probe.hip && ./launch_mode_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/resource.h>
#include <time.h>

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

__global__ __launch_bounds__(256) void k_step(int32_t* ctrl, long long step, int n, Big b) {
  if (b.host_ctrl && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&b.host_ctrl[0], (int)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float4 v = b.in[t];
  // two dependent round trips + some arithmetic, like the real step
  const int j = ((int)(v.x * 1e3f) & 0x7fffffff) % n;
  const float4 w = b.in[j];
#pragma unroll 1
  for (int k = 0; k < 200; ++k) v.y = v.y * 0.999f + w.x * 1e-3f;
  v.x = w.x * 0.5f + v.x * 0.5f;
  b.out[t] = v;
  (void)ctrl;
}

// last node of a captured batch: counts finished batches where the host can see it
__global__ void k_tick(int32_t* ctr, int32_t* host) {
  const int v = ctr[0] + 1;
  ctr[0] = v;
  __hip_atomic_store(&host[1], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the same step with its argument block in device memory (kernarg = a few scalars + one pointer)
__global__ __launch_bounds__(256) void k_step_ptr(int32_t* ctrl, long long step, int n, const Big* bp, int flip) {
  Big b = *bp;
  if (flip) {
    const float4* t = b.in;
    b.in = b.out;
    b.out = const_cast<float4*>(t);
  }
  if (b.host_ctrl && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&b.host_ctrl[0], (int)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float4 v = b.in[t];
  const int j = ((int)(v.x * 1e3f) & 0x7fffffff) % n;
  const float4 w = b.in[j];
#pragma unroll 1
  for (int k = 0; k < 200; ++k) v.y = v.y * 0.999f + w.x * 1e-3f;
  v.x = w.x * 0.5f + v.x * 0.5f;
  b.out[t] = v;
  (void)ctrl;
}

__global__ __launch_bounds__(256) void k_step_ptr_mid(int32_t* ctrl, long long step, int n, const Big* bp, int flip, long long p0,
                                                      long long p1, long long p2, long long p3, long long p4, long long p5,
                                                      long long p6, long long p7) {
  Big b = *bp;
  if (flip) {
    const float4* t = b.in;
    b.in = b.out;
    b.out = const_cast<float4*>(t);
  }
  if (b.host_ctrl && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&b.host_ctrl[0], (int)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float4 v = b.in[t];
  const int j = ((int)(v.x * 1e3f) & 0x7fffffff) % n;
  const float4 w = b.in[j];
#pragma unroll 1
  for (int k = 0; k < 200; ++k) v.y = v.y * 0.999f + w.x * 1e-3f;
  v.x = w.x * 0.5f + v.x * 0.5f + (float)((p0 ^ p1 ^ p2 ^ p3 ^ p4 ^ p5 ^ p6 ^ p7) & 0) ;
  b.out[t] = v;
  (void)ctrl;
}

static double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char** argv) {
  const int n = 32768 * 4, launches = argc > 1 ? atoi(argv[1]) : 100000, window = 12;
  float4 *a, *b;
  int32_t *ctrl, *host;
  CK(hipMalloc(&a, 16 * (size_t)n));
  CK(hipMalloc(&b, 16 * (size_t)n));
  CK(hipMalloc(&ctrl, 64));
  CK(hipHostMalloc(&host, 64, hipHostMallocMapped));
  CK(hipMemset(a, 0, 16 * (size_t)n));
  Big big = {};
  big.host_ctrl = host;
  hipStream_t st = nullptr;  // argv[2]: 0 = null stream (what torch's default stream is), 1 = blocking stream, 2 = non-blocking stream
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const int throttle = argc > 3 ? atoi(argv[3]) : 1;  // 0: no progress word, no window (launch as fast as the host can)
  if (!throttle) big.host_ctrl = nullptr;
  const int query_every = argc > 4 ? atoi(argv[4]) : 0;  // > 0: hipStreamQuery from the launching thread every so many launches
  if (mode == 1) CK(hipStreamCreate(&st));
  if (mode == 2) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int small_kernarg = argc > 6 ? atoi(argv[6]) : 0;  // 1: argument block in device memory
  Big* big_d = nullptr;
  if (small_kernarg) {
    Big tmp = big;
    tmp.in = a;
    tmp.out = b;
    CK(hipMalloc(&big_d, sizeof(Big)));
    CK(hipMemcpy(big_d, &tmp, sizeof(Big), hipMemcpyHostToDevice));
  }
  auto run = [&](int count, long long s0) {
    volatile int32_t* hc = host;
    for (int i = 0; i < count; ++i) {
      const long long s = s0 + i;
      while (throttle && s - hc[0] > window) __builtin_ia32_pause();
      big.in = (i & 1) ? b : a;
      big.out = (i & 1) ? a : b;
      if (small_kernarg == 2) hipLaunchKernelGGL(k_step_ptr_mid, dim3(n / 256), dim3(256), 0, st, ctrl, s, n, big_d, i & 1, 1ll, 2ll, 3ll, 4ll, 5ll, 6ll, 7ll, 8ll);
      else if (small_kernarg) hipLaunchKernelGGL(k_step_ptr, dim3(n / 256), dim3(256), 0, st, ctrl, s, n, big_d, i & 1);
      else hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, st, ctrl, s, n, big);
      if (query_every > 0 && (i % query_every) == query_every - 1) (void)hipStreamQuery(st);
    }
    CK(hipDeviceSynchronize());
  };
  // argv[5] = G > 0: the same chain replayed from a hipGraph of G step launches (+ the tick), at most two replays in flight
  const int graph_len = argc > 5 ? atoi(argv[5]) : 0;
  hipGraphExec_t gexec = nullptr;
  hipStream_t gs = st;
  if (graph_len > 0) {
    if (!gs) CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
    Big gb = big;
    gb.host_ctrl = nullptr;
    hipGraph_t g;
    CK(hipStreamBeginCapture(gs, hipStreamCaptureModeGlobal));
    for (int i = 0; i < graph_len; ++i) {
      gb.in = (i & 1) ? b : a;
      gb.out = (i & 1) ? a : b;
      hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, gs, ctrl, (long long)i, n, gb);
    }
    hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, gs, ctrl, host);
    CK(hipStreamEndCapture(gs, &g));
    CK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
    CK(hipMemset(ctrl, 0, 64));
    host[1] = 0;
  }
  auto run_graph = [&](int count) {  // count launches = count / graph_len replays
    volatile int32_t* hc = host;
    const int replays = count / graph_len;
    const int base = hc[1];
    for (int r = 0; r < replays; ++r) {
      while ((base + r) - hc[1] >= 2) __builtin_ia32_pause();
      CK(hipGraphLaunch(gexec, gs));
    }
    CK(hipStreamSynchronize(gs));
  };
  host[0] = -1;
  double t0 = now();
  long long s = 0;
  while (now() - t0 < 3.0) {  // clock warm-up
    if (graph_len > 0) run_graph(2048);
    else run(2000, s);
    s += 2000;
  }
  for (int rep = 0; rep < 5; ++rep) {
    rusage r0, r1;
    getrusage(RUSAGE_THREAD, &r0);
    const double t1 = now();
    if (graph_len > 0) run_graph(launches);
    else run(launches, s);
    s += launches;
    const double dt = now() - t1;
    getrusage(RUSAGE_THREAD, &r1);
    printf("stream mode %d, window %s, query every %d, graph of %d, small kernarg %d: %d launches: %.0f per second (%.2f us each), voluntary context switches %ld, involuntary %ld\n", mode, throttle ? "on" : "off", query_every, graph_len, small_kernarg, launches,
           launches / dt, 1e6 * dt / launches, r1.ru_nvcsw - r0.ru_nvcsw, r1.ru_nivcsw - r0.ru_nivcsw);
  }
  return 0;
}
