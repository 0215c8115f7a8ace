// What sets the launch-to-launch floor of a tiny kernel of the step kernel's shape?  (GPU box only, diagnostic.)
// Every variant: thread t loads in[t] (float4) and stores it to out[t]; 32768*8 threads; 512 launches back to back.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o floor_probe floor_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

struct Big {  // ~ the step kernel's by-value argument block
  int n;
  const float4* in;
  float4* out;
  const void* p[20];
  float f[16];
  int tail;
};

template <int NT>
__global__ __launch_bounds__(NT) void k_small(const float4* __restrict__ in, float4* __restrict__ out, int n) {
  const int t = blockIdx.x * NT + threadIdx.x;
  if (t < n) out[t] = in[t];
}

__global__ __launch_bounds__(256) void k_big(Big b) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < b.n + b.tail) b.out[t] = b.in[t];
}

// argument block read from a persistent device buffer instead of the kernarg segment
__global__ __launch_bounds__(256) void k_bigptr(const Big* __restrict__ bp) {
  const Big b = *bp;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < b.n + b.tail) b.out[t] = b.in[t];
}

__global__ __launch_bounds__(256) void k_lds(const float4* __restrict__ in, float4* __restrict__ out, int n) {
  __shared__ float s[96];
  const int t = blockIdx.x * 256 + threadIdx.x;
  float4 v = t < n ? in[t] : make_float4(0, 0, 0, 0);
  if (threadIdx.x < 96) s[threadIdx.x] = v.x;
  __syncthreads();
  if (t < n) {
    v.w += s[threadIdx.x % 96] * 0.f;
    out[t] = v;
  }
}

__global__ __launch_bounds__(256) void k_vgpr(const float4* __restrict__ in, float4* __restrict__ out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  asm volatile("" ::: "v95");  // forces an allocation of 96 VGPRs
  if (t < n) out[t] = in[t];
}

__global__ __launch_bounds__(256) void k_ctrl(const int* __restrict__ ctrl, long long step, const float4* __restrict__ in,
                                              float4* __restrict__ out, int n) {
  if ((long long)ctrl[3] <= step) return;
  const int w = ctrl[4 + (int)(step & 1)];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n + (w == 12345)) out[t] = in[t];
}

template <typename F>
static void bench(const char* name, F launch, int reps = 512) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) {
    for (int i = 0; i < 20; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.3f us/launch\n", name, 1e3 * ms / reps);
  }
}

int main() {
  const int n = 32768 * 8;
  float4 *a, *b;
  int* ctrl;
  Big* bigd;
  CK(hipMalloc(&a, 16 * (size_t)n));
  CK(hipMalloc(&b, 16 * (size_t)n));
  CK(hipMalloc(&ctrl, 64));
  CK(hipMalloc(&bigd, sizeof(Big)));
  CK(hipMemset(a, 0, 16 * (size_t)n));
  int h[8] = {0, 0, 0, 0x7fffffff, -2, -2, 0, 0};
  CK(hipMemcpy(ctrl, h, 32, hipMemcpyHostToDevice));
  Big big = {};
  big.n = n;
  big.in = a;
  big.out = b;
  CK(hipMemcpy(bigd, &big, sizeof(Big), hipMemcpyHostToDevice));
  printf("sizeof(Big) = %zu\n", sizeof(Big));
  bench("E0 1024 x 256, 3 scalar args", [&](int i) { hipLaunchKernelGGL(k_small<256>, dim3(n / 256), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E1 1024 x 256, 256-B by-value struct", [&](int i) { Big g = big; g.in = (i & 1) ? b : a; g.out = (i & 1) ? a : b; hipLaunchKernelGGL(k_big, dim3(n / 256), dim3(256), 0, 0, g); });
  bench("E1b 1024 x 256, struct in a device buffer", [&](int i) { hipLaunchKernelGGL(k_bigptr, dim3(n / 256), dim3(256), 0, 0, bigd); });
  bench("E2 1024 x 256, LDS + barrier", [&](int i) { hipLaunchKernelGGL(k_lds, dim3(n / 256), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E3 1024 x 256, 96 VGPRs", [&](int i) { hipLaunchKernelGGL(k_vgpr, dim3(n / 256), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E4 256 x 1024", [&](int i) { hipLaunchKernelGGL(k_small<1024>, dim3(n / 1024), dim3(1024), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E5 512 x 512", [&](int i) { hipLaunchKernelGGL(k_small<512>, dim3(n / 512), dim3(512), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E6 2048 x 128", [&](int i) { hipLaunchKernelGGL(k_small<128>, dim3(n / 128), dim3(128), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E6b 4096 x 64", [&](int i) { hipLaunchKernelGGL(k_small<64>, dim3(n / 64), dim3(64), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E7 1024 x 256, halt + stale control words", [&](int i) { hipLaunchKernelGGL(k_ctrl, dim3(n / 256), dim3(256), 0, 0, ctrl, (long long)i, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  bench("E8 256 x 256 (a quarter of the threads)", [&](int i) { hipLaunchKernelGGL(k_small<256>, dim3(256), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, 65536); });
  bench("E9 1 x 64 (one wave)", [&](int i) { hipLaunchKernelGGL(k_small<64>, dim3(1), dim3(64), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, 64); });
  // the same tiny kernel replayed from a hipGraph (64 dependent launches per graph): is the eager figure the
  // host's launch rate or the device's dependent-dispatch rate?
  {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 64; ++i)
      hipLaunchKernelGGL(k_small<256>, dim3(n / 256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) {
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < 16; ++k) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-44s %7.3f us/launch\n", "G0 E0 from a 64-node hipGraph", 1e3 * ms / (16 * 64));
    }
  }
  return 0;
}
