// Round 5 probe: does dispatching the step launches as hand-written AQL packets on a private HSA queue beat
// hipLaunchKernelGGL -- (a) on the host (cost of one dispatch), (b) on the device, by lowering the packet's acquire /
// release fence scopes?  (GPU box only, diagnostic; nothing here is product code.)
//
// HIP gives every kernel of a stream the barrier bit and agent-scope acquire + release fences: on a part with eight
// non-coherent L2s the acquire drops every non-local line of every L2 at each kernel start, so the step kernel
// re-fetches its (unchanged) index list from Infinity Cache in every launch (profiles/r04_pmc_hot_kernels.md).  A packet
// with acquire = release = NONE skips both cache operations; what the launches hand to each other must then travel by
// write-through (sc1) stores and sc1 loads, as inside a persistent kernel (MI355X_MICROARCH.md, "Valid forms").
//
// The probe kernel has the step kernel's dependent chain at C3 geometry: N = 32768 particles x 4 lanes, one 16-byte
// index quad per lane (read-only), own position + four gathered positions (ping-pong buffers), a 4-lane reduction,
// lane 0 integrates (reads / writes v, writes the new position), a halt word read by every launch.
//
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 aql_probe.hip -o aql_probe -lhsa-runtime64
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 --cuda-device-only aql_probe.hip -o aql_probe.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <fcntl.h>
#include <unistd.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <string>
#include <vector>

struct PArgs {
  int n;
  int pad;
  const float4* x_cur;
  float4* x_next;
  float* v;
  const uint4* fl;
  int* ctrl;
};

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f3v __attribute__((ext_vector_type(3)));

// MODE 0: plain loads of everything; stores write-through (like the product kernel)
// MODE 1: every load of data another launch wrote is an sc1 load (positions, velocities, halt word)
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(PArgs a, long long step) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int p = tid >> 2, q = tid & 3;
  if (p >= a.n) return;
  const uint4 e = a.fl[tid];  // read-only between "rebuilds": plain load, may stay in L2
  int halt;
  if (MODE == 1) halt = __hip_atomic_load(a.ctrl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else halt = a.ctrl[3];
  if ((long long)halt <= step) return;
  f4v xi, x0, x1, x2, x3;
  if (MODE == 1) {
    asm volatile(
        "global_load_dwordx4 %0, %5, off sc1\n\t"
        "global_load_dwordx4 %1, %6, off sc1\n\t"
        "global_load_dwordx4 %2, %7, off sc1\n\t"
        "global_load_dwordx4 %3, %8, off sc1\n\t"
        "global_load_dwordx4 %4, %9, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(xi), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
        : "v"(a.x_cur + p), "v"(a.x_cur + e.x), "v"(a.x_cur + e.y), "v"(a.x_cur + e.z), "v"(a.x_cur + e.w)
        : "memory");
  } else {
    const float4 t = a.x_cur[p], t0 = a.x_cur[e.x], t1 = a.x_cur[e.y], t2 = a.x_cur[e.z], t3 = a.x_cur[e.w];
    xi = (f4v){t.x, t.y, t.z, t.w};
    x0 = (f4v){t0.x, t0.y, t0.z, t0.w};
    x1 = (f4v){t1.x, t1.y, t1.z, t1.w};
    x2 = (f4v){t2.x, t2.y, t2.z, t2.w};
    x3 = (f4v){t3.x, t3.y, t3.z, t3.w};
  }
  float fx = 0.f, fy = 0.f, fz = 0.f;
  const f4v xs[4] = {x0, x1, x2, x3};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = xi.x - xs[k].x, dy = xi.y - xs[k].y, dz = xi.z - xs[k].z;
    const float r2 = dx * dx + dy * dy + dz * dz + 0.25f;
    const float s = 1.0f / (r2 * r2);
    fx += s * dx, fy += s * dy, fz += s * dz;
  }
  fx += __shfl_xor(fx, 1), fy += __shfl_xor(fy, 1), fz += __shfl_xor(fz, 1);
  fx += __shfl_xor(fx, 2), fy += __shfl_xor(fy, 2), fz += __shfl_xor(fz, 2);
  if (q == 0) {
    float v0, v1, v2;
    if (MODE == 1) {
      f3v vv;
      asm volatile("global_load_dwordx3 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(vv) : "v"(a.v + 3 * (size_t)p) : "memory");
      v0 = vv.x, v1 = vv.y, v2 = vv.z;
    } else {
      v0 = a.v[3 * (size_t)p], v1 = a.v[3 * (size_t)p + 1], v2 = a.v[3 * (size_t)p + 2];
    }
    v0 = 0.999f * v0 + 1e-3f * fx, v1 = 0.999f * v1 + 1e-3f * fy, v2 = 0.999f * v2 + 1e-3f * fz;
    const f3v vo = {v0, v1, v2};
    const f4v xo = {xi.x + 1e-3f * v0, xi.y + 1e-3f * v1, xi.z + 1e-3f * v2, xi.w + 1.0f};
    asm volatile("global_store_dwordx3 %0, %1, off sc1\n\ts_nop 1" ::"v"(a.v + 3 * (size_t)p), "v"(vo) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(a.x_next + p), "v"(xo) : "memory");
  }
}

// ---- staleness test (mode "stale"): does a launch with acquire = NONE read stale L1 / scalar-cache / L2 lines? ----
// k_fill (ONE workgroup) rewrites a 4 KB buffer; k_check (1024 workgroups: every CU of every XCD, several per CU) reads
// all of it with plain vector loads and one scalar load and counts words that are not the expected value.  The check
// of iteration i leaves the buffer (value i) in every L1, scalar cache and L2; the fill writes i + 1; the next check
// must see i + 1 everywhere.
struct SArgs {
  int* buf;
  int* bad;  // [0] vector mismatches, [1] scalar mismatches
  int val;
  int wt;    // k_fill: 1 = write-through (sc1) stores, 0 = plain stores
};
__global__ __launch_bounds__(256) void k_fill(SArgs a) {
  int* p = a.buf + threadIdx.x * 4;
  const int v = a.val;
  if (a.wt) {
    typedef int i4v __attribute__((ext_vector_type(4)));
    const i4v t = {v, v, v, v};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  } else {
    p[0] = v, p[1] = v, p[2] = v, p[3] = v;
  }
}
__global__ __launch_bounds__(256) void k_check(SArgs a) {
  const int4 t = reinterpret_cast<const int4*>(a.buf)[threadIdx.x];
  const int s0 = a.buf[(blockIdx.x & 7) * 32];  // uniform address: a scalar load
  int bad = (t.x != a.val) + (t.y != a.val) + (t.z != a.val) + (t.w != a.val);
  if (bad) atomicAdd(a.bad, bad);
  if (threadIdx.x == 0 && s0 != a.val) atomicAdd(a.bad + 1, 1);
}

// ---- queue -> HIP stream hand-over without the host (mode "wait") ----
__global__ void k_flag_set(int* flag, int value) { __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ __launch_bounds__(256) void k_copy4(const float4* in, float4* out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n) out[t] = in[t];
}

// ---- which XCD does block b of consecutive dispatches land on? (mode "xcc") ----
__global__ __launch_bounds__(256) void k_xcc(int* out, int launch, int nblk) {
  if (threadIdx.x == 0) {
    const int id = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF;  // HW_REG_XCC_ID, bits 3:0
    out[(size_t)launch * nblk + blockIdx.x] = id;
  }
}

template __global__ void k_probe<0>(PArgs, long long);
template __global__ void k_probe<1>(PArgs, long long);

#ifndef __HIP_DEVICE_COMPILE__
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)
#define HK(x)                                                            \
  do {                                                                   \
    hsa_status_t s_ = (x);                                               \
    if (s_ != HSA_STATUS_SUCCESS && s_ != HSA_STATUS_INFO_BREAK) {       \
      const char* m_ = nullptr;                                          \
      hsa_status_string(s_, &m_);                                        \
      fprintf(stderr, "%s:%d hsa error %d %s\n", __FILE__, __LINE__, (int)s_, m_ ? m_ : ""); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

static double now_us() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

struct Sym {
  std::string name;
  uint64_t object;
  uint32_t kernarg, group, priv;
};
static std::vector<Sym> g_syms;
static hsa_status_t sym_cb(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t s, void*) {
  hsa_symbol_kind_t kind;
  hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind);
  if (kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
  uint32_t len = 0;
  hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
  std::string name(len, '\0');
  hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME, &name[0]);
  Sym y;
  y.name = name;
  hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &y.object);
  hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &y.kernarg);
  hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &y.group);
  hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &y.priv);
  g_syms.push_back(y);
  return HSA_STATUS_SUCCESS;
}

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) g_gpu = a, g_have_gpu = true;
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) g_cpu = a, g_have_cpu = true;
  return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_kernarg_pool, g_dev_pool;
static bool g_have_kernarg = false, g_have_dev = false;
static hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_kernarg) g_kernarg_pool = pool, g_have_kernarg = true;
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t dev_pool_cb(hsa_amd_memory_pool_t pool, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  bool alloc = false;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev) g_dev_pool = pool, g_have_dev = true;
  return HSA_STATUS_SUCCESS;
}

static void queue_err(hsa_status_t s, hsa_queue_t*, void*) {
  const char* m = nullptr;
  hsa_status_string(s, &m);
  fprintf(stderr, "HSA QUEUE ERROR %d %s\n", (int)s, m ? m : "");
  fflush(stderr);
  _exit(3);
}

struct State {
  int n;
  float4 *xa, *xb;
  float* v;
  uint4* fl;
  int* ctrl;
  std::vector<float4> x0;
};

static void reset(State& s) {
  CK(hipMemcpy(s.xa, s.x0.data(), sizeof(float4) * s.n, hipMemcpyHostToDevice));
  CK(hipMemset(s.xb, 0, sizeof(float4) * s.n));
  CK(hipMemset(s.v, 0, sizeof(float) * 3 * s.n));
  int c[16] = {0};
  c[3] = INT_MAX;
  CK(hipMemcpy(s.ctrl, c, sizeof(c), hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
}
static uint64_t hash_state(State& s, int steps) {
  std::vector<float4> h(s.n);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), (steps & 1) ? s.xb : s.xa, sizeof(float4) * s.n, hipMemcpyDeviceToHost));
  uint64_t x = 1469598103934665603ull;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(h.data());
  for (size_t i = 0; i < (size_t)s.n * 4; ++i) x = (x ^ w[i]) * 1099511628211ull;
  return x;
}

int main(int argc, char** argv) {
  const int n = 32768, K = argc > 1 ? atoi(argv[1]) : 4000, reps = 3;
  const char* hsaco = argc > 2 ? argv[2] : "aql_probe.hsaco";
  State s;
  s.n = n;
  CK(hipSetDevice(0));
  CK(hipMalloc(&s.xa, sizeof(float4) * n));
  CK(hipMalloc(&s.xb, sizeof(float4) * n));
  CK(hipMalloc(&s.v, sizeof(float) * 3 * n));
  CK(hipMalloc(&s.fl, sizeof(uint4) * n * 4));
  CK(hipMalloc(&s.ctrl, 64));
  s.x0.resize(n);
  uint32_t r = 12345u;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return r >> 8; };
  for (int i = 0; i < n; ++i) s.x0[i] = make_float4((i % 32) + (rnd() & 255) / 512.f, ((i / 32) % 32) + (rnd() & 255) / 512.f, (i / 1024) + (rnd() & 255) / 512.f, 0.f);
  std::vector<uint4> fl((size_t)n * 4);
  for (int p = 0; p < n; ++p)
    for (int q = 0; q < 4; ++q) {
      uint32_t e[4];
      for (int k = 0; k < 4; ++k) {
        int d = (int)(rnd() % 2049) - 1024;  // neighbours a few thousand slots away at most, like cell order
        int j = p + d;
        if (j < 0) j += n;
        if (j >= n) j -= n;
        if (j == p) j = (p + 1) % n;
        e[k] = (uint32_t)j;
      }
      fl[(size_t)p * 4 + q] = make_uint4(e[0], e[1], e[2], e[3]);
    }
  CK(hipMemcpy(s.fl, fl.data(), sizeof(uint4) * fl.size(), hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)((n * 4 + 255) / 256);

  // ---------------- baseline: hipLaunchKernelGGL ----------------
  uint64_t want = 0;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < reps; ++rep) {
      reset(s);
      PArgs a = {n, 0, s.xa, s.xb, s.v, s.fl, s.ctrl};
      const double t0 = now_us();
      double t_enq = 0;
      for (int i = 0; i < K; ++i) {
        a.x_cur = (i & 1) ? s.xb : s.xa;
        a.x_next = (i & 1) ? s.xa : s.xb;
        const double e0 = now_us();
        if (mode) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 0, 0, a, (long long)i);
        else hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 0, 0, a, (long long)i);
        t_enq += now_us() - e0;
      }
      CK(hipDeviceSynchronize());
      const double t1 = now_us();
      const uint64_t h = hash_state(s, K);
      if (mode == 0 && rep == 0) want = h;
      printf("HIP eager  mode %d : %7.3f us/launch  host %5.2f us/launch  hash %016llx %s\n", mode, (t1 - t0) / K, t_enq / K,
             (unsigned long long)h, h == want ? "" : "MISMATCH");
    }
  }
  fflush(stdout);

  // ---------------- HSA: queue, code object, kernarg ring ----------------
  HK(hsa_init());
  HK(hsa_iterate_agents(agent_cb, nullptr));
  if (!g_have_gpu || !g_have_cpu) return fprintf(stderr, "no agents\n"), 1;
  HK(hsa_amd_agent_iterate_memory_pools(g_cpu, pool_cb, nullptr));
  HK(hsa_amd_agent_iterate_memory_pools(g_gpu, dev_pool_cb, nullptr));
  if (!g_have_kernarg) return fprintf(stderr, "no kernarg pool\n"), 1;
  hsa_queue_t* q = nullptr;
  const uint32_t QSIZE = 1024;
  HK(hsa_queue_create(g_gpu, QSIZE, HSA_QUEUE_TYPE_SINGLE, queue_err, nullptr, UINT32_MAX, UINT32_MAX, &q));
  hsa_file_t fd = open(hsaco, 0);
  if (fd < 0) return fprintf(stderr, "cannot open %s\n", hsaco), 1;
  hsa_code_object_reader_t reader;
  HK(hsa_code_object_reader_create_from_file(fd, &reader));
  hsa_executable_t exe;
  HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  HK(hsa_executable_freeze(exe, nullptr));
  HK(hsa_executable_iterate_agent_symbols(exe, g_gpu, sym_cb, nullptr));
  const Sym* ks[2] = {nullptr, nullptr};
  for (const Sym& y : g_syms) {
    printf("symbol %s kernarg %u group %u private %u\n", y.name.c_str(), y.kernarg, y.group, y.priv);
    if (y.name.find("k_probeILi0E") != std::string::npos) ks[0] = &y;
    if (y.name.find("k_probeILi1E") != std::string::npos) ks[1] = &y;
  }
  if (!ks[0] || !ks[1]) return fprintf(stderr, "kernels not found\n"), 1;
  const size_t SLOT = 512;
  if (argc > 3 && !strcmp(argv[3], "wait")) {
    const Sym* kfl = nullptr;
    for (const Sym& y : g_syms)
      if (y.name.find("k_flag_set") != std::string::npos) kfl = &y;
    if (!kfl || !g_have_dev) return fprintf(stderr, "wait: kernel or device pool missing\n"), 1;
    char* ring = nullptr;
    HK(hsa_amd_memory_pool_allocate(g_dev_pool, SLOT * QSIZE, 0, (void**)&ring));
    HK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, ring));
    hsa_signal_t done;
    HK(hsa_signal_create(1, 0, nullptr, &done));
    float4* y;
    CK(hipMalloc(&y, sizeof(float4) * n));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto push = [&](const Sym& k, unsigned nblk, const void* a, size_t sz, int acq, int rel, bool last) {
      uint64_t widx = hsa_queue_add_write_index_relaxed(q, 1);
      while (widx - hsa_queue_load_read_index_scacquire(q) >= QSIZE - 1) {
      }
      char* ka = ring + SLOT * (widx % QSIZE);
      memcpy(ka, a, sz);
      hsa_kernel_dispatch_packet_t* pk = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (widx % QSIZE);
      pk->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
      pk->workgroup_size_x = nblk == 0 ? 1 : 256, pk->workgroup_size_y = 1, pk->workgroup_size_z = 1;
      pk->reserved0 = 0;
      pk->grid_size_x = nblk == 0 ? 1 : nblk * 256, pk->grid_size_y = 1, pk->grid_size_z = 1;
      pk->private_segment_size = k.priv;
      pk->group_segment_size = k.group;
      pk->kernel_object = k.object;
      pk->kernarg_address = ka;
      pk->reserved2 = 0;
      pk->completion_signal = last ? done : hsa_signal_t{0};
      const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
      __atomic_store_n(reinterpret_cast<uint16_t*>(pk), header, __ATOMIC_RELEASE);
      hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)widx);
    };
    for (int kind = 0; kind < 3; ++kind) {
      int* flag = nullptr;
      hipError_t e = hipSuccess;
      const char* kname = kind == 0 ? "hipMalloc word" : (kind == 1 ? "hipMallocSignalMemory" : "host join (reference)");
      if (kind == 0) e = hipMalloc(&flag, 64);
      else if (kind == 1) e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
      else e = hipMalloc(&flag, 64);
      if (e != hipSuccess) {
        printf("wait test, %s: allocation failed (%s)\n", kname, hipGetErrorString(e));
        (void)hipGetLastError();
        continue;
      }
      CK(hipMemset(flag, 0, 4));
      const int steps = 200;
      for (int rep = 0; rep < 5; ++rep) {
        reset(s);
        CK(hipMemset(y, 0, sizeof(float4) * n));
        CK(hipDeviceSynchronize());
        hsa_signal_store_relaxed(done, 1);
        const int gen = rep + 1;
        const double t0 = now_us();
        for (int i = 0; i < steps; ++i) {
          struct { PArgs a; long long step; } ka = {{n, 0, (i & 1) ? s.xb : s.xa, (i & 1) ? s.xa : s.xb, s.v, s.fl, s.ctrl}, i};
          push(*ks[0], blocks, &ka, sizeof(ka), i == 0 ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, false);
        }
        struct { int* f; int v; } fa = {flag, gen};
        push(*kfl, 0, &fa, sizeof(fa), HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_SYSTEM, true);
        const double t_enq = now_us();
        hipError_t we = hipSuccess;
        if (kind < 2) {
          we = hipStreamWaitValue32(st, flag, (uint32_t)gen, hipStreamWaitValueGte, 0xFFFFFFFFu);
          if (we != hipSuccess) {
            printf("wait test, %s: hipStreamWaitValue32 failed (%s)\n", kname, hipGetErrorString(we));
            (void)hipGetLastError();
          }
        } else {
          while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) {
          }
        }
        const float4* xfin = (steps & 1) ? s.xb : s.xa;
        hipLaunchKernelGGL(k_copy4, dim3((n + 255) / 256), dim3(256), 0, st, xfin, y, n);
        CK(hipStreamSynchronize(st));
        const double t1 = now_us();
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) {
        }
        std::vector<float4> hy(n), hx(n);
        CK(hipMemcpy(hy.data(), y, sizeof(float4) * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx.data(), xfin, sizeof(float4) * n, hipMemcpyDeviceToHost));
        const bool same = memcmp(hy.data(), hx.data(), sizeof(float4) * n) == 0;
        printf("wait test, %-22s rep %d: %d steps + hand-over + copy kernel: %.1f us (enqueue %.1f us)  copy saw the final state: %s\n", kname, rep,
               steps, t1 - t0, t_enq - t0, same ? "yes" : "NO");
        fflush(stdout);
        if (we != hipSuccess) break;
      }
      CK(hipFree(flag));
    }
    printf("rc 0\n");
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "xcc")) {
    const Sym* kx = nullptr;
    for (const Sym& y : g_syms)
      if (y.name.find("k_xcc") != std::string::npos) kx = &y;
    if (!kx || !g_have_dev) return fprintf(stderr, "xcc: kernel or device pool missing\n"), 1;
    char* ring = nullptr;
    HK(hsa_amd_memory_pool_allocate(g_dev_pool, SLOT * QSIZE, 0, (void**)&ring));
    HK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, ring));
    hsa_signal_t done;
    HK(hsa_signal_create(1, 0, nullptr, &done));
    const int L = 64, NB = 512;
    int* out;
    CK(hipMalloc(&out, sizeof(int) * L * NB));
    CK(hipMemset(out, 0xFF, sizeof(int) * L * NB));
    CK(hipDeviceSynchronize());
    for (int i = 0; i < L; ++i) {
      uint64_t widx = hsa_queue_add_write_index_relaxed(q, 1);
      char* ka = ring + SLOT * (widx % QSIZE);
      struct { int* o; int l; int nb; } a = {out, i, NB};
      memcpy(ka, &a, sizeof(a));
      hsa_kernel_dispatch_packet_t* pk = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (widx % QSIZE);
      pk->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
      pk->workgroup_size_x = 256, pk->workgroup_size_y = 1, pk->workgroup_size_z = 1;
      pk->reserved0 = 0;
      pk->grid_size_x = NB * 256, pk->grid_size_y = 1, pk->grid_size_z = 1;
      pk->private_segment_size = kx->priv;
      pk->group_segment_size = kx->group;
      pk->kernel_object = kx->object;
      pk->kernarg_address = ka;
      pk->reserved2 = 0;
      pk->completion_signal = i == L - 1 ? done : hsa_signal_t{0};
      const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
      __atomic_store_n(reinterpret_cast<uint16_t*>(pk), header, __ATOMIC_RELEASE);
      hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)widx);
    }
    while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) {
    }
    std::vector<int> h((size_t)L * NB);
    CK(hipMemcpy(h.data(), out, sizeof(int) * L * NB, hipMemcpyDeviceToHost));
    int exact = 0;
    printf("XCD of block 0 in %d consecutive dispatches of %d blocks:", L, NB);
    for (int i = 0; i < L; ++i) {
      printf(" %d", h[(size_t)i * NB]);
      bool rr = true;
      for (int b = 0; b < NB; ++b) rr &= h[(size_t)i * NB + b] == (h[(size_t)i * NB] + b) % 8;
      exact += rr;
    }
    printf("\nexact round-robin (block b on XCD (first + b) %% 8) in %d of %d dispatches\n", exact, L);
    printf("first dispatch, blocks 0..31:");
    for (int b = 0; b < 32; ++b) printf(" %d", h[b]);
    printf("\nrc 0\n");
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "stale")) {
    const Sym *kf = nullptr, *kc = nullptr;
    for (const Sym& y : g_syms) {
      if (y.name.find("k_fill") != std::string::npos) kf = &y;
      if (y.name.find("k_check") != std::string::npos) kc = &y;
    }
    if (!kf || !kc || !g_have_dev) return fprintf(stderr, "stale: kernels or device pool missing\n"), 1;
    char* ring = nullptr;
    HK(hsa_amd_memory_pool_allocate(g_dev_pool, SLOT * QSIZE, 0, (void**)&ring));
    HK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, ring));
    int *buf, *bad;
    CK(hipMalloc(&buf, 4096));
    CK(hipMalloc(&bad, 64));
    hsa_signal_t done;
    HK(hsa_signal_create(1, 0, nullptr, &done));
    auto push = [&](const Sym& k, unsigned nblk, const SArgs& a, int acq, int rel, bool last) {
      uint64_t widx = hsa_queue_add_write_index_relaxed(q, 1);
      while (widx - hsa_queue_load_read_index_scacquire(q) >= QSIZE - 1) {
      }
      char* ka = ring + SLOT * (widx % QSIZE);
      memcpy(ka, &a, sizeof(a));
      hsa_kernel_dispatch_packet_t* pk = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (widx % QSIZE);
      pk->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
      pk->workgroup_size_x = 256, pk->workgroup_size_y = 1, pk->workgroup_size_z = 1;
      pk->reserved0 = 0;
      pk->grid_size_x = nblk * 256, pk->grid_size_y = 1, pk->grid_size_z = 1;
      pk->private_segment_size = k.priv;
      pk->group_segment_size = k.group;
      pk->kernel_object = k.object;
      pk->kernarg_address = ka;
      pk->reserved2 = 0;
      pk->completion_signal = last ? done : hsa_signal_t{0};
      const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
      __atomic_store_n(reinterpret_cast<uint16_t*>(pk), header, __ATOMIC_RELEASE);
      hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)widx);
    };
    struct V { const char* name; int acq, rel, wt; };
    const V vs[] = {{"acq agent  rel agent  write-through fill", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 1},
                    {"acq agent  rel agent  plain fill        ", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 0},
                    {"acq NONE   rel NONE   write-through fill", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 1},
                    {"acq NONE   rel agent  plain fill        ", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT, 0},
                    {"acq NONE   rel NONE   plain fill        ", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0},
                    {"acq agent  rel NONE   plain fill        ", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE, 0}};
    const int iters = K;
    for (const V& v : vs) {
      CK(hipMemset(buf, 0, 4096));
      CK(hipMemset(bad, 0, 64));
      CK(hipDeviceSynchronize());
      hsa_signal_store_relaxed(done, 1);
      const double t0 = now_us();
      for (int i = 0; i < iters; ++i) {
        SArgs a = {buf, bad, i, v.wt};
        const bool first = i == 0, last = i == iters - 1;
        push(*kf, 1, a, first ? HSA_FENCE_SCOPE_SYSTEM : v.acq, v.rel, false);
        push(*kc, 1024, a, v.acq, last ? HSA_FENCE_SCOPE_SYSTEM : v.rel, last);
      }
      while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) {
      }
      const double t1 = now_us();
      int h[2];
      CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
      printf("stale test, %s: %d iterations, %d stale vector words, %d stale scalar loads (%.2f us per fill+check)\n", v.name, iters, h[0],
             h[1], (t1 - t0) / iters);
      fflush(stdout);
    }
    printf("rc 0\n");
    return 0;
  }
  for (int kernarg_dev = 0; kernarg_dev < 2; ++kernarg_dev) {
    char* ring = nullptr;
    if (!kernarg_dev) {
      HK(hsa_amd_memory_pool_allocate(g_kernarg_pool, SLOT * QSIZE, 0, (void**)&ring));
      HK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ring));
    } else {
      if (!g_have_dev) break;
      HK(hsa_amd_memory_pool_allocate(g_dev_pool, SLOT * QSIZE, 0, (void**)&ring));
      if (hsa_amd_agents_allow_access(1, &g_cpu, nullptr, ring) != HSA_STATUS_SUCCESS) {
        printf("device-memory kernargs: the host may not map them; skipped\n");
        break;
      }
    }
    hsa_signal_t done;
    HK(hsa_signal_create(1, 0, nullptr, &done));
    struct Variant { const char* name; int acq, rel, mode; };
    const Variant vs[] = {
        {"sys/sys  plain", HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, 0},
        {"agt/agt  plain", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 0},
        {"agt/agt  sc1  ", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 1},
        {"agt/none sc1  ", HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE, 1},
        {"none/agt sc1  ", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT, 1},
        {"none/none sc1 ", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 1},
        {"none/none plain (expected stale)", HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0},
    };
    for (const Variant& vr : vs) {
      for (int rep = 0; rep < reps; ++rep) {
        reset(s);
        hsa_signal_store_relaxed(done, 1);
        const Sym& k = *ks[vr.mode];
        double t_enq = 0;
        const double t0 = now_us();
        for (int i = 0; i < K; ++i) {
          // a slot of the ring is free once the packet that used it has been consumed
          uint64_t widx = hsa_queue_add_write_index_relaxed(q, 1);
          while (widx - hsa_queue_load_read_index_scacquire(q) >= QSIZE - 1) {
          }
          const double e0 = now_us();
          char* ka = ring + SLOT * (widx % QSIZE);
          PArgs a = {n, 0, (i & 1) ? s.xb : s.xa, (i & 1) ? s.xa : s.xb, s.v, s.fl, s.ctrl};
          memcpy(ka, &a, sizeof(a));
          const long long step = i;
          memcpy(ka + sizeof(PArgs), &step, 8);
          // implicit arguments (code object v5 layout) behind the explicit ones
          char* im = ka + ((sizeof(PArgs) + 8 + 7) & ~(size_t)7);
          memset(im, 0, 256);
          const uint32_t bc[3] = {blocks, 1, 1};
          memcpy(im, bc, 12);
          const uint16_t gs[3] = {256, 1, 1};
          memcpy(im + 12, gs, 6);
          const uint16_t dims = 1;
          memcpy(im + 64, &dims, 2);
          hsa_kernel_dispatch_packet_t* pk = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (widx % QSIZE);
          pk->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
          pk->workgroup_size_x = 256, pk->workgroup_size_y = 1, pk->workgroup_size_z = 1;
          pk->reserved0 = 0;
          pk->grid_size_x = blocks * 256, pk->grid_size_y = 1, pk->grid_size_z = 1;
          pk->private_segment_size = k.priv;
          pk->group_segment_size = k.group;
          pk->kernel_object = k.object;
          pk->kernarg_address = ka;
          pk->reserved2 = 0;
          const bool last = i == K - 1;
          pk->completion_signal = last ? done : hsa_signal_t{0};
          const int acq = (i == 0) ? HSA_FENCE_SCOPE_SYSTEM : vr.acq;
          const int rel = last ? HSA_FENCE_SCOPE_SYSTEM : vr.rel;
          const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                  (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
          __atomic_store_n(reinterpret_cast<uint16_t*>(pk), header, __ATOMIC_RELEASE);
          hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)widx);
          t_enq += now_us() - e0;
        }
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) {
          fprintf(stderr, "still waiting for the queue ...\n");
        }
        const double t1 = now_us();
        const uint64_t h = hash_state(s, K);
        printf("AQL %s kernargs %-6s: %7.3f us/launch  host %5.2f us/dispatch  hash %016llx %s\n", vr.name, kernarg_dev ? "device" : "host",
               (t1 - t0) / K, t_enq / K, (unsigned long long)h, h == want ? "" : "MISMATCH");
        fflush(stdout);
      }
    }
    hsa_signal_destroy(done);
  }
  hsa_queue_destroy(q);
  printf("rc 0\n");
  return 0;
}
#endif
