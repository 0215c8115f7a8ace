// libchx: the step queue.  Inside LangevinIntegrator.run() and inside a chunk of the device-resident Monte Carlo loop the
// library's kernels are dispatched as hand-written AQL kernel-dispatch packets on a private user-mode HSA queue instead
// of through hipLaunchKernelGGL: the fused step kernels always (chx_launch_step), unpack / the list build chain / pack /
// the Monte Carlo trial while the queue is open (chx_launch.h).
//
// Why (round 5, scripts/ub/aql_probe.hip, profiles/r05_step_queue.md):
//   * the host side of a dispatch is a copy of the argument block into device memory, a 64-byte packet and a doorbell
//     write -- about a microsecond against the 2.5 us of a HIP launch (5+ us in the HIP runtime's slow launch regime,
//     profiles/r03_launch_regime.md, which starves a 5 us step kernel for the life of one fresh process in ten): the
//     device's rate no longer depends on the launching thread;
//   * the packet header carries the fence scopes.  HIP gives every kernel agent-scope acquire + release fences; a
//     steady-state step launch needs neither: its outputs leave as write-through (sc1) stores or atomics, and the hardware
//     invalidates the vector L1 and the scalar cache at every dispatch whatever the acquire scope says (measured: 3000
//     rewrite / re-read rounds over every CU, not one stale word with acquire = release = NONE and write-through
//     stores; with plain stores and no release every word is stale -- the test has teeth; the first queue of a process
//     repeats it on the card at hand, selfcheck_fences).  Without the two cache operations a dependent launch boundary is
//     0.25 us shorter.  Every other packet ends with an agent-scope release.  CHX_AQL_FENCE=agent keeps HIP's scopes;
//   * a list rebuild inside a run is a sequence of packets behind the drained no-op launches: no join, no HIP launch.
//
// Ordering with the HIP stream the rest of libchx (and torch) launches on -- the Python layer thinks in ONE stream:
//   stream -> queue: opening the queue (chx_aql_open; the first dispatch does it by itself).  Nothing pending on the
//                 stream (hipStreamQuery): nothing to wait behind.  Otherwise a barrier-AND packet on a gate signal,
//                 and a one-thread marker kernel launched on the stream opens the gate: the packets run behind
//                 everything that was enqueued on the stream before them;
//   queue -> stream: chx_aql_join() -- a barrier packet with a completion signal, and the host waits for it.  Every libchx
//                 entry point that is not queue-capable joins before it launches on a stream (chx_stream), a
//                 queue-capable one rides on the open queue (chx_launch.h), and the Python layer joins where it used to
//                 wait for the stream.
// The kernels come from libchx_step.hsaco (the device half of chx_lj.hip) and libchx_k_<unit>.hsaco, compiled by
// chiron_amd/build.py from the same sources with the same flags as the copies inside libchx.so.  The HSA runtime is the
// image the process's HIP runtime already runs on (the one that is initialised; bound with dlsym: no second runtime, no
// link-time dependency).  At most 128 packets are outstanding (a completion signal per 32-packet segment).
#include <cxxabi.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <hsa/amd_hsa_signal.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <link.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "chx_aql.h"
#include "chx_common.h"

// write-combined stores through the BAR must have left the core before the read-back / the doorbell
static inline void store_fence() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_sfence();
#else
  __sync_synchronize();
#endif
}

int chx_aql_open_count = 0;
// the start-up check's kernels (chx_core.hip)
const void* chx_selfcheck_fill_fn();
const void* chx_selfcheck_read_fn();
const void* chx_debug_spin_fn();

namespace {

// ---- the HSA entry points, bound at run time ----
#define CHX_HSA_FUNCS(X)                                                                                           \
  X(hsa_init) X(hsa_shut_down) X(hsa_iterate_agents) X(hsa_agent_get_info) X(hsa_queue_create) X(hsa_queue_destroy)  \
  X(hsa_queue_add_write_index_relaxed) X(hsa_queue_load_read_index_scacquire) X(hsa_signal_create)                   \
  X(hsa_signal_store_relaxed) X(hsa_signal_store_screlease) X(hsa_signal_load_scacquire)                             \
  X(hsa_signal_wait_scacquire) X(hsa_code_object_reader_create_from_file) X(hsa_executable_create_alt)              \
  X(hsa_executable_load_agent_code_object) X(hsa_executable_freeze) X(hsa_executable_iterate_agent_symbols)         \
  X(hsa_executable_symbol_get_info) X(hsa_amd_agent_iterate_memory_pools) X(hsa_amd_memory_pool_get_info)           \
  X(hsa_amd_memory_pool_allocate) X(hsa_amd_agents_allow_access) X(hsa_amd_signal_create) X(hsa_status_string)
struct Hsa {
#define X(name) decltype(&::name) name = nullptr;
  CHX_HSA_FUNCS(X)
#undef X
};
Hsa H;

// every HSA runtime image resident in the process (a python process under rocprofv3 holds two: the system's, which the
// profiler's preloaded tool brought along and which the HIP runtime then binds to, and the copy bundled with torch)
int phdr_cb(struct dl_phdr_info* info, size_t, void* data) {
  std::vector<std::string>* found = static_cast<std::vector<std::string>*>(data);
  if (!info->dlpi_name) return 0;
  const std::string name = info->dlpi_name;
  const size_t slash = name.rfind('/');
  const std::string base = slash == std::string::npos ? name : name.substr(slash + 1);
  if (base.rfind("libhsa-runtime64.so", 0) == 0) found->push_back(name);
  return 0;
}

// The HSA runtime the process's HIP runtime is running on: the resident image that is INITIALISED.  Which image HIP's
// calls bind to depends on the load order (the one next to libamdhip64 in a plain python process, the preloaded system
// copy under a profiler); asking each image for the system timestamp tells without side effects -- an image nobody has
// initialised answers HSA_STATUS_ERROR_NOT_INITIALIZED.  None initialised: the process has not started HIP, and the step
// queue is not the one to start a runtime.
void* find_hsa() {
  std::vector<std::string> found;
  dl_iterate_phdr(phdr_cb, &found);
  void* live = nullptr;
  for (const std::string& name : found) {
    void* h = dlopen(name.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (!h) continue;
    auto get = reinterpret_cast<decltype(&::hsa_system_get_info)>(dlsym(h, "hsa_system_get_info"));
    uint64_t t = 0;
    const bool up = get != nullptr && get(HSA_SYSTEM_INFO_TIMESTAMP, &t) == HSA_STATUS_SUCCESS;
    if (getenv("CHX_AQL_VERBOSE")) fprintf(stderr, "[chx] step queue: HSA runtime image %s: %s\n", name.c_str(), up ? "initialised" : "idle");
    if (up && !live) live = h;
    else dlclose(h);
  }
  return live;
}

struct Gate {
  hsa_signal_t sig;
};

struct Queue {
  hsa_queue_t* q = nullptr;
  char* ring = nullptr;  // kernarg slots in device memory, written by the host through the BAR
  uint32_t size = 0;     // packets in the ring: what the runtime made of the request (an intercepted queue may differ)
  bool open = false;     // packets outstanding since the last join
  bool fresh = false;    // the gate is up and no kernel packet has followed it yet
  bool overlap_ok = false;    // the packet just written may run beside the next STEP packet (chx_aql_allow_overlap)
  bool fresh_system = false;  // ... and the queue was opened on an idle stream, without a barrier packet: acquire at system scope
  hsa_signal_t done{};   // completion signal of the join barrier
  hsa_signal_t fin{};    // ... of a packet dispatched as the LAST of a run (chx_aql_dispatch: final_packet)
  bool final_sent = false;  // the newest packet carries `fin` and a system-scope release: a join only has to wait for it
  bool fin_unobserved = false;  // a packet carrying `fin` was dispatched and nobody has waited for it since
  Gate gates[4];
  unsigned gate_next = 0;
  // Run-ahead bound.  The ring is cut into segments of SEG packets; the last packet of a segment carries that segment's
  // completion signal, and the first packet of segment k waits until segment k - AHEAD has completed: at most
  // AHEAD * SEG packets are ever outstanding.  Two reasons: (1) the kernarg slot of packet w is reused by packet
  // w + size, which is safe only once packet w has COMPLETED (its waves read the block while they run) -- the queue's
  // read index does not say that: it says "consumed", and on an intercepted queue (rocprofv3) "copied out", long before
  // the packet runs; (2) a profiler's interceptor expands every packet into several on the real queue (counter
  // collection: about five), and its overflow path did not survive a caller that runs hundreds of packets ahead
  // (round 5: segmentation faults inside the runtime's doorbell handler with --pmc, and with --kernel-trace on 47 us
  // kernels).  The integrator's launch window keeps a dozen packets outstanding: it never waits here.
  static constexpr unsigned SEG = 32, AHEAD = 4, MAXSEG = 128;
  hsa_signal_t seg_sig[MAXSEG];
  bool seg_used[MAXSEG];
  uint64_t n_dispatched = 0, n_joins = 0, n_gates = 0, n_idle_opens = 0;
  uint64_t checked_at = 0;  // n_dispatched when the fence-free packets' assumption was last checked on this queue
};

constexpr uint32_t QSIZE = 1024;  // packets asked for; the launch window keeps ~a dozen outstanding
constexpr size_t SLOT = 4096;     // bytes per kernarg slot (the step kernel's block: 320; a batched step's: 8 x 304; any kernel's + 256 implicit)

std::mutex g_mu;
int g_state = 0;  // 0 untried, 1 on, -1 off
bool g_fence_none = true;
int g_device = -1;  // the HIP device the queue's agent is (the one current when the queue was set up)
hsa_agent_t g_gpu{}, g_cpu{};
hsa_amd_memory_pool_t g_dev_pool{};
bool g_have_gpu = false, g_have_cpu = false, g_have_pool = false;
std::map<std::string, AqlKernel> g_kernels;  // demangled "void k_lj_step<4, true, false, false>" -> descriptor
std::map<hipStream_t, Queue*> g_queues;
char g_err[512] = "";
volatile int g_queue_error = 0;

void fail(const char* what, hsa_status_t s = HSA_STATUS_SUCCESS) {
  const char* m = nullptr;
  if (s != HSA_STATUS_SUCCESS && H.hsa_status_string) H.hsa_status_string(s, &m);
  snprintf(g_err, sizeof(g_err), "step queue: %s%s%s", what, m ? ": " : "", m ? m : "");
  if (getenv("CHX_AQL_VERBOSE")) fprintf(stderr, "[chx] %s\n", g_err);
}

struct AgentPick {
  uint32_t want_bdf;
  uint32_t want_domain;
  bool have_want;
  int n_gpu;
  hsa_agent_t first_gpu;
  hsa_agent_t match;
  bool matched;
  int n_bdf;  // agents whose bus / device / function match (whatever their domain query said)
  hsa_agent_t bdf_match;
  char want_uuid[17];  // the HIP device's UUID (16 characters), "" when unknown
  int n_uuid;
  hsa_agent_t uuid_match;
};
hsa_status_t agent_cb(hsa_agent_t a, void* data) {
  AgentPick* p = static_cast<AgentPick*>(data);
  hsa_device_type_t t;
  if (H.hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) g_cpu = a, g_have_cpu = true;
  if (t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
  if (p->n_gpu++ == 0) p->first_gpu = a;
  uint32_t bdf = 0, domain = 0;
  H.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
  H.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
  if (p->have_want && !p->matched && bdf == p->want_bdf && domain == p->want_domain) p->match = a, p->matched = true;
  if (p->have_want && bdf == p->want_bdf && p->n_bdf++ == 0) p->bdf_match = a;
  if (p->want_uuid[0]) {
    char uuid[32] = "";  // "GPU-" + 16 hex characters
    if (H.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_UUID, uuid) == HSA_STATUS_SUCCESS &&
        strncmp(uuid, "GPU-", 4) == 0 && strncmp(uuid + 4, p->want_uuid, 16) == 0 && p->n_uuid++ == 0)
      p->uuid_match = a;
  }
  return HSA_STATUS_SUCCESS;
}
hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void*) {
  hsa_amd_segment_t seg;
  H.hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  bool alloc = false;
  H.hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  H.hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_pool) g_dev_pool = pool, g_have_pool = true;
  return HSA_STATUS_SUCCESS;
}
hsa_status_t sym_cb(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t s, void*) {
  hsa_symbol_kind_t kind;
  H.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind);
  if (kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
  uint32_t len = 0;
  H.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
  std::string name(len, '\0');
  H.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME, &name[0]);
  if (name.size() > 3 && name.compare(name.size() - 3, 3, ".kd") == 0) name.resize(name.size() - 3);
  AqlKernel k{};
  H.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object);
  H.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size);
  H.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_size);
  H.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_size);
  int status = 0;
  char* dem = abi::__cxa_demangle(name.c_str(), nullptr, nullptr, &status);
  std::string key = (status == 0 && dem) ? dem : name;
  free(dem);
  const size_t paren = key.find('(');
  if (paren != std::string::npos) key.resize(paren);
  g_kernels[key] = k;
  g_kernels[name] = k;  // (the mangled symbol too: what hipKernelNameRefByPtr hands out)
  return HSA_STATUS_SUCCESS;
}
void queue_error_cb(hsa_status_t s, hsa_queue_t*, void*) {
  fail("the queue reported an error", s);
  fprintf(stderr, "[chx] %s\n", g_err);
  g_queue_error = 1;
}

std::string own_dir() {
  Dl_info info;
  if (!dladdr(reinterpret_cast<void*>(&chx_aql_on), &info) || !info.dli_fname) return ".";
  const std::string f = info.dli_fname;
  const size_t slash = f.rfind('/');
  return slash == std::string::npos ? "." : f.substr(0, slash);
}

bool init_locked() {
  const char* e = getenv("CHX_AQL");
  if (e && e[0] == '0') return fail("off (CHX_AQL=0)"), false;
  const char* fe = getenv("CHX_AQL_FENCE");
  g_fence_none = !(fe && (fe[0] == 'a' || fe[0] == 'A'));  // "agent": HIP's scopes on every packet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail("no HIP device"), false;
  g_device = dev;
  void* lib = find_hsa();
  if (!lib) return fail("the HSA runtime of this process's HIP runtime was not found"), false;
#define X(name)                                                      \
  H.name = reinterpret_cast<decltype(H.name)>(dlsym(lib, #name));    \
  if (!H.name) return fail("missing HSA entry point " #name), false;
  CHX_HSA_FUNCS(X)
#undef X
  hsa_status_t s = H.hsa_init();  // (reference-counted: the runtime is up already)
  if (s != HSA_STATUS_SUCCESS) return fail("hsa_init", s), false;
  AgentPick pick{};
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), dev) == hipSuccess) {
    unsigned dom = 0, b = 0, d = 0, f = 0;
    if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4) {
      pick.have_want = true;
      pick.want_domain = dom;
      pick.want_bdf = (b << 8) | (d << 3) | f;
    }
  }
  {
    hipUUID uu;
    memset(&uu, 0, sizeof(uu));
    if (hipDeviceGetUuid(&uu, dev) == hipSuccess && uu.bytes[0] != 0) memcpy(pick.want_uuid, uu.bytes, 16);
    else (void)hipGetLastError();
  }
  H.hsa_iterate_agents(agent_cb, &pick);
  // The queue must sit on the very card the HIP device is: by UUID when both runtimes report one; else by PCI address; a
  // UUID that names one agent and a PCI address that names another switch the queue off rather than guess.
  if (pick.n_uuid == 1 && pick.matched && pick.uuid_match.handle != pick.match.handle)
    return fail("the HIP device's UUID and PCI address name different HSA agents"), false;
  if (pick.n_uuid == 1) g_gpu = pick.uuid_match, g_have_gpu = true;
  else if (pick.matched) g_gpu = pick.match, g_have_gpu = true;
  else if (pick.n_bdf == 1) g_gpu = pick.bdf_match, g_have_gpu = true;  // (a runtime that does not report PCI domains)
  else if (pick.n_gpu == 1) g_gpu = pick.first_gpu, g_have_gpu = true;
  if (!g_have_gpu || !g_have_cpu) return fail("the HIP device was not found among the HSA agents"), false;
  if (getenv("CHX_AQL_VERBOSE"))
    fprintf(stderr, "[chx] step queue: HIP device %d (%s, UUID %.16s) is the HSA agent matched by %s (%d GPU agents)\n", dev, bus,
            pick.want_uuid[0] ? pick.want_uuid : "?", pick.n_uuid == 1 ? "UUID" : (pick.matched ? "PCI address" : (pick.n_bdf == 1 ? "bus/device/function" : "being the only one")),
            pick.n_gpu);
  H.hsa_amd_agent_iterate_memory_pools(g_gpu, pool_cb, nullptr);
  if (!g_have_pool) return fail("no device memory pool"), false;
  // the kernels' code objects: libchx_step.hsaco (the step kernels: required) and libchx_k_*.hsaco (the other
  // translation units' device halves: optional, each one makes its kernels dispatchable through the queue)
  hsa_executable_t exe;
  s = H.hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe);
  if (s != HSA_STATUS_SUCCESS) return fail("hsa_executable_create_alt", s), false;
  const std::string dir = own_dir();
  std::vector<std::string> files = {dir + "/libchx_step.hsaco"};
  for (const char* unit : {"core", "neighbors", "cell_fl", "mc"}) files.push_back(dir + "/libchx_k_" + unit + ".hsaco");
  int loaded = 0;
  for (size_t i = 0; i < files.size(); ++i) {
    const int fd = open(files[i].c_str(), O_RDONLY);
    if (fd < 0) {
      if (i == 0) return fail("libchx_step.hsaco not found beside libchx.so"), false;
      continue;
    }
    hsa_code_object_reader_t reader;
    s = H.hsa_code_object_reader_create_from_file(fd, &reader);
    if (s == HSA_STATUS_SUCCESS) s = H.hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr);
    close(fd);
    if (s != HSA_STATUS_SUCCESS) {
      if (i == 0) return fail("loading libchx_step.hsaco", s), false;
      if (getenv("CHX_AQL_VERBOSE")) fprintf(stderr, "[chx] step queue: %s not loaded\n", files[i].c_str());
      continue;
    }
    ++loaded;
  }
  s = H.hsa_executable_freeze(exe, nullptr);
  if (s != HSA_STATUS_SUCCESS) return fail("hsa_executable_freeze", s), false;
  H.hsa_executable_iterate_agent_symbols(exe, g_gpu, sym_cb, nullptr);
  if (getenv("CHX_AQL_VERBOSE")) fprintf(stderr, "[chx] step queue: %d code objects, %zu kernel names\n", loaded, g_kernels.size());
  if (g_kernels.empty()) return fail("libchx_step.hsaco holds no kernels"), false;
  return true;
}

bool g_selfchecked = false;
int g_selfcheck_result = -1;  // -1 not run, 0 passed, > 0 stale words seen (fences switched back on)
int64_t g_selfcheck_runs = 0;  // times the check has run in this process

// forward: the packet helpers below
uint64_t reserve(Queue* Q, hsa_signal_t* completion);
inline void publish(Queue* Q, uint64_t widx, uint16_t header);
bool wait_zero(hsa_signal_t sig, const char* what, Queue* Q = nullptr);
inline void* slot_of(Queue* Q, uint64_t widx);

// One raw kernel packet with explicit fence scopes (the self-check only).
void raw_packet(Queue* Q, const AqlKernel* k, unsigned blocks, const void* args, size_t nbytes, int acquire, int release,
                hsa_signal_t completion) {
  hsa_signal_t seg{0};
  const uint64_t widx = reserve(Q, &seg);
  char* ka = Q->ring + SLOT * (widx % Q->size);
  memcpy(ka, args, nbytes);
  store_fence();
  (void)*reinterpret_cast<volatile uint32_t*>(ka + ((nbytes - 4) & ~(size_t)3));
  hsa_kernel_dispatch_packet_t* pk = static_cast<hsa_kernel_dispatch_packet_t*>(slot_of(Q, widx));
  pk->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  pk->workgroup_size_x = 256, pk->workgroup_size_y = 1, pk->workgroup_size_z = 1;
  pk->reserved0 = 0;
  pk->grid_size_x = blocks * 256u, pk->grid_size_y = 1, pk->grid_size_z = 1;
  pk->private_segment_size = k->private_size;
  pk->group_segment_size = k->group_size;
  pk->kernel_object = k->object;
  pk->kernarg_address = ka;
  pk->reserved2 = 0;
  pk->completion_signal = completion.handle ? completion : seg;
  if (completion.handle && seg.handle) H.hsa_signal_store_relaxed(seg, 0);  // (the caller waits for everything anyway)
  publish(Q, widx, (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE)));
}

// Start-up check of the assumption the fence-free packets rest on (chx_core.hip: k_selfcheck_*): 256 rounds of
// rewrite (one workgroup, write-through) / re-read (1024 workgroups, plain vector and scalar loads) with acquire =
// release = NONE on every packet.  One stale word -> every packet gets HIP's agent-scope fences for the life of the
// process (and the status string says so).  About a millisecond, once per process.  CHX_AQL_SELFCHECK=0 skips it.
void selfcheck_fences(Queue* Q, hipStream_t st, bool again = false) {
  if (!again) {
    if (g_selfchecked || !g_fence_none) return;
    g_selfchecked = true;
    const char* e = getenv("CHX_AQL_SELFCHECK");
    if (e && e[0] == '0') return;
  }
  auto name_of = [&](const void* fn) -> const AqlKernel* {
    const char* nm = hipKernelNameRefByPtr(fn, st);
    if (!nm) return nullptr;
    auto f = g_kernels.find(nm);
    return f == g_kernels.end() ? nullptr : &f->second;
  };
  const AqlKernel* kf = name_of(chx_selfcheck_fill_fn());
  const AqlKernel* kr = name_of(chx_selfcheck_read_fn());
  if (!kf || !kr) return;  // (libchx_k_core.hsaco absent: nothing to run the check with)
  // (the check's 8 KB and its signal are made once and kept: the check comes back -- chx_step_queue_selfcheck, and every
  // RECHECK_EVERY packets of a queue)
  static int32_t* mem = nullptr;
  static hsa_signal_t done{0};
  if (mem == nullptr) {
    int32_t* m = nullptr;
    if (H.hsa_amd_memory_pool_allocate(g_dev_pool, 8192, 0, reinterpret_cast<void**>(&m)) != HSA_STATUS_SUCCESS) return;
    if (H.hsa_amd_agents_allow_access(1, &g_cpu, nullptr, m) != HSA_STATUS_SUCCESS) return;
    if (H.hsa_signal_create(1, 0, nullptr, &done) != HSA_STATUS_SUCCESS) return;
    mem = m;
  }
  int32_t* buf = mem;
  int32_t* bad = mem + 1536;  // (its own 128-byte lines, away from the 4 KB under test)
  bad[0] = bad[1] = 0;
  store_fence();
  (void)*reinterpret_cast<volatile int32_t*>(bad + 1);
  H.hsa_signal_store_relaxed(done, 1);
  const int rounds = 256;
  for (int i = 0; i < rounds; ++i) {
    struct { int32_t* b; int v; } fa = {buf, i + 1};
    struct { const int32_t* b; int v; int32_t* bad; } ra = {buf, i + 1, bad};
    raw_packet(Q, kf, 1, &fa, sizeof(fa), i == 0 ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, hsa_signal_t{0});
    const bool last = i == rounds - 1;
    raw_packet(Q, kr, 1024, &ra, sizeof(ra), HSA_FENCE_SCOPE_NONE, last ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE,
               last ? done : hsa_signal_t{0});
  }
  if (!wait_zero(done, "the start-up check of the step queue did not finish")) return;
  const int stale = reinterpret_cast<volatile int32_t*>(bad)[0] + reinterpret_cast<volatile int32_t*>(bad)[1];
  g_selfcheck_result = stale;
  g_selfcheck_runs++;
  if (stale != 0) {
    g_fence_none = false;
    snprintf(g_err, sizeof(g_err), "step queue: the start-up check read %d stale words behind acquire-less dispatches: every packet "
                                   "carries agent-scope fences in this process", stale);
    fprintf(stderr, "[chx] %s\n", g_err);
  } else if (getenv("CHX_AQL_VERBOSE")) {
    fprintf(stderr, "[chx] step queue: start-up check passed (%d rewrite / re-read rounds without fences, no stale word)\n", rounds);
  }
}

Queue* queue_for(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_queues.find(st);
  if (it != g_queues.end()) return it->second;
  Queue* Q = new Queue();
  hsa_status_t s = H.hsa_queue_create(g_gpu, QSIZE, HSA_QUEUE_TYPE_SINGLE, queue_error_cb, nullptr, UINT32_MAX, UINT32_MAX, &Q->q);
  if (s != HSA_STATUS_SUCCESS) return fail("hsa_queue_create", s), delete Q, nullptr;
  Q->size = Q->q->size;
  if (Q->size < 16 || (Q->size & (Q->size - 1)) != 0) return fail("the queue's size is not a power of two >= 16"), delete Q, nullptr;
  if (getenv("CHX_AQL_VERBOSE")) fprintf(stderr, "[chx] step queue: %u packets (asked for %u), ring at %p\n", Q->size, QSIZE, Q->q->base_address);
  s = H.hsa_amd_memory_pool_allocate(g_dev_pool, SLOT * (size_t)Q->size, 0, reinterpret_cast<void**>(&Q->ring));
  if (s == HSA_STATUS_SUCCESS) s = H.hsa_amd_agents_allow_access(1, &g_cpu, nullptr, Q->ring);
  if (s != HSA_STATUS_SUCCESS) return fail("kernarg ring in device memory", s), delete Q, nullptr;
  s = H.hsa_signal_create(0, 0, nullptr, &Q->done);
  if (s == HSA_STATUS_SUCCESS) s = H.hsa_signal_create(0, 0, nullptr, &Q->fin);
  for (Gate& g : Q->gates)
    if (s == HSA_STATUS_SUCCESS) s = H.hsa_amd_signal_create(1, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &g.sig);
  const unsigned nseg = Q->size / Queue::SEG;
  if (nseg > Queue::MAXSEG || nseg < 2 * Queue::AHEAD) return fail("unexpected queue size"), delete Q, nullptr;
  for (unsigned i = 0; i < nseg; ++i) {
    Q->seg_used[i] = false;
    if (s == HSA_STATUS_SUCCESS) s = H.hsa_signal_create(0, 0, nullptr, &Q->seg_sig[i]);
  }
  if (s != HSA_STATUS_SUCCESS) return fail("signals", s), delete Q, nullptr;
  g_queues[st] = Q;
  selfcheck_fences(Q, st);  // (the process's first queue: once)
  return Q;
}

inline void* slot_of(Queue* Q, uint64_t widx) { return reinterpret_cast<char*>(Q->q->base_address) + 64 * (widx % Q->size); }

// Host wait for a completion signal.  Gives up -- and marks the queue broken for the process -- when the queue has made
// NO PROGRESS for two minutes (its read index stands still: a gate that waits behind minutes of stream work is not a
// hang) or the runtime has reported an error; the failure is also left for the calling entry point's chx_check_launch.
bool wait_zero(hsa_signal_t sig, const char* what, Queue* Q) {
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  uint64_t seen = Q ? H.hsa_queue_load_read_index_scacquire(Q->q) : 0;
  while (H.hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, 1000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) {
    if (g_queue_error) {
      chx_note_launch_failure(CHX_ERR_LAUNCH);
      return false;
    }
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (Q) {
      const uint64_t now = H.hsa_queue_load_read_index_scacquire(Q->q);
      if (now != seen) seen = now, t0 = t1;
    }
    if (t1.tv_sec - t0.tv_sec > 120) {
      fail(what);
      fprintf(stderr, "[chx] %s\n", g_err);
      g_queue_error = 1;
      chx_set_error("%s", g_err);
      chx_note_launch_failure(CHX_ERR_LAUNCH);
      return false;
    }
  }
  return true;
}

// A slot of the packet ring (and with it a kernarg slot).  Returns the write index; *completion = the signal the packet
// must carry (its segment's, when the slot is the last of a segment), else a null signal.
uint64_t reserve(Queue* Q, hsa_signal_t* completion) {
  const uint64_t widx = H.hsa_queue_add_write_index_relaxed(Q->q, 1);
  const unsigned QS = Q->size, nseg = QS / Queue::SEG;
  const unsigned slot = (unsigned)(widx % QS), seg = slot / Queue::SEG;
  if (slot % Queue::SEG == 0) {
    const unsigned back = (seg + nseg - Queue::AHEAD) % nseg;
    if (Q->seg_used[back]) {
      (void)wait_zero(Q->seg_sig[back], "the step queue has made no progress for 120 s", Q);
      Q->seg_used[back] = false;
    }
  }
  while (widx - H.hsa_queue_load_read_index_scacquire(Q->q) >= QS - 1) {
    if (g_queue_error) break;
  }
  *completion = hsa_signal_t{0};
  if (slot % Queue::SEG == Queue::SEG - 1) {
    H.hsa_signal_store_relaxed(Q->seg_sig[seg], 1);
    Q->seg_used[seg] = true;
    *completion = Q->seg_sig[seg];
  }
  return widx;
}
inline void publish(Queue* Q, uint64_t widx, uint16_t header) {
  __atomic_store_n(reinterpret_cast<uint16_t*>(slot_of(Q, widx)), header, __ATOMIC_RELEASE);
  H.hsa_signal_store_screlease(Q->q->doorbell_signal, (hsa_signal_value_t)widx);
}
void push_barrier(Queue* Q, hsa_signal_t dep, hsa_signal_t completion, int acquire, int release) {
  Q->final_sent = false;
  hsa_signal_t seg_sig{0};
  const uint64_t widx = reserve(Q, &seg_sig);
  if (seg_sig.handle != 0) {
    // the last slot of a segment went to a barrier packet: it carries the segment's signal unless it has one of its own
    // (the join's: the join then waits for everything, and the segment counts as completed)
    if (completion.handle == 0) completion = seg_sig;
    else H.hsa_signal_store_relaxed(seg_sig, 0);
  }
  hsa_barrier_and_packet_t* pk = static_cast<hsa_barrier_and_packet_t*>(slot_of(Q, widx));
  pk->reserved0 = 0;
  pk->reserved1 = 0;
  for (int i = 0; i < 5; ++i) pk->dep_signal[i] = hsa_signal_t{0};
  pk->dep_signal[0] = dep;
  pk->reserved2 = 0;
  pk->completion_signal = completion;
  publish(Q, widx, (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE)));
}

}  // namespace

// the marker that opens a gate from the HIP stream: everything enqueued on the stream before it has completed
__global__ void k_aql_gate_open(long long* value) {
  __hip_atomic_store(value, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

bool chx_aql_on() {
  int st = __atomic_load_n(&g_state, __ATOMIC_ACQUIRE);
  if (st == 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_state == 0) __atomic_store_n(&g_state, init_locked() ? 1 : -1, __ATOMIC_RELEASE);
    st = g_state;
  }
  if (st != 1 || g_queue_error) return false;
  // (one process, one GPU is the deployment -- a rank per card; a thread that has made ANOTHER device current launches
  // through HIP as before: the queue belongs to the card it was opened on)
  int dev = -1;
  return hipGetDevice(&dev) == hipSuccess && dev == g_device;
}

const AqlKernel* chx_aql_kernel(const char* prefix) {
  auto it = g_kernels.find(prefix);  // (filled once under the lock in init_locked, read-only afterwards)
  return it == g_kernels.end() ? nullptr : &it->second;
}

int chx_aql_hot_release() { return g_fence_none ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT; }

namespace {

// the HIP -> queue hand-over: behind everything the stream holds
void open_gate(Queue* Q, hipStream_t st) {
  // Nothing pending on the stream (the usual case at the start of a run(): the caller has synchronised, or the previous
  // run() ended with a join): there is nothing to wait behind, so no marker launch and no barrier packet -- the first
  // packet goes out with a system-scope acquire instead.  (The stream belongs to the calling thread: nothing can be
  // enqueued on it between this query and the packets that follow.)
  if (hipStreamQuery(st) == hipSuccess) {
    Q->open = true;
    Q->fresh = true;
    Q->fresh_system = true;
    Q->n_idle_opens++;
    __atomic_fetch_add(&chx_aql_open_count, 1, __ATOMIC_RELAXED);
    return;
  }
  (void)hipGetLastError();  // (hipErrorNotReady is the expected answer, not an error to keep)
  Gate& g = Q->gates[Q->gate_next++ % 4];
  H.hsa_signal_store_relaxed(g.sig, 1);
  long long* value = reinterpret_cast<long long*>(const_cast<int64_t*>(&reinterpret_cast<amd_signal_t*>(g.sig.handle)->value));
  hipLaunchKernelGGL(k_aql_gate_open, dim3(1), dim3(1), 0, st, value);
  push_barrier(Q, g.sig, hsa_signal_t{0}, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_NONE);
  Q->open = true;
  Q->fresh = true;
  Q->n_gates++;
  __atomic_fetch_add(&chx_aql_open_count, 1, __ATOMIC_RELAXED);
}

// Debug mode (chx_queue_debug): a spin packet of g_debug_delay_us microseconds goes in front of (a) the first packet behind
// a gate, (b) every packet that is not a hot one -- unpack, the list build chain, pack, prologues: the launches whose
// outputs somebody outside the queue reads -- and (c) a run's final launch.  Whoever reads such an output without having
// joined then reads it EARLY every time, not once in a thousand runs; together with the poison the Python layer puts
// into those arrays when it allocates them (chiron_amd/ops.py: fresh) a missing join is a deterministic failure.
int g_debug_delay_us = [] {
  const char* e = getenv("CHX_QUEUE_DEBUG");
  const int v = e ? atoi(e) : 0;
  return v < 0 ? 0 : v;
}();
const AqlKernel* g_spin_kernel = nullptr;
bool g_spin_looked = false;
uint64_t g_debug_spins = 0;

void debug_spin_packet(Queue* Q, hipStream_t st) {
  if (!g_spin_looked) {
    g_spin_looked = true;
    const char* nm = hipKernelNameRefByPtr(chx_debug_spin_fn(), st);
    if (nm) {
      auto f = g_kernels.find(nm);
      if (f != g_kernels.end()) g_spin_kernel = &f->second;
    }
    if (!g_spin_kernel) fprintf(stderr, "[chx] step queue debug mode: no code object holds k_debug_spin (no delays)\n");
  }
  if (!g_spin_kernel) return;
  struct { long long ticks; } a = {(long long)g_debug_delay_us * 100ll};
  raw_packet(Q, g_spin_kernel, 1, &a, sizeof(a), HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, hsa_signal_t{0});
  g_debug_spins++;
}

int dispatch_core(hipStream_t st, const AqlKernel* k, dim3 grid, dim3 block, const void* kernarg, size_t nbytes, int release,
                  bool is_step = false, bool final_packet = false, bool hot_acquire = false, bool beside_previous = false,
                  bool leaves_queue = false) {
  // (the segment may be shorter than the caller's block -- trailing padding of a by-value struct -- never longer)
  const size_t seg = k->kernarg_size;
  if (nbytes + 256 > SLOT || nbytes > seg + 8) {
    // more explicit bytes than the kernel declares (+ alignment slack): the packed block does not belong to this kernel
    chx_set_error("step queue: argument block of %zu bytes does not fit the kernel's segment of %zu", nbytes, seg);
    return CHX_ERR_INVALID;
  }
  Queue* Q = queue_for(st);
  if (!Q || g_queue_error) {
    chx_set_error("%s", g_err);
    return CHX_ERR_LAUNCH;
  }
  // Acquire scope: NONE only for the launches that are on a microsecond budget and whose inputs are known to live in
  // device-local memory written through or released by their predecessors (the fused steps, the Monte Carlo trial); every
  // other packet -- the list build chain, pack / unpack, prologues -- takes an agent-scope acquire like a HIP launch.
  int acquire = (g_fence_none && (is_step || hot_acquire)) ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT;
  if (!Q->open) open_gate(Q, st);
  if (g_debug_delay_us > 0 && (Q->fresh || !(is_step || hot_acquire) || final_packet || leaves_queue)) debug_spin_packet(Q, st);
  // (a step packet behind the list build's summary launch starts beside it: chx_aql_allow_overlap)
  const bool barrier = !(is_step && (Q->overlap_ok || (beside_previous && !Q->fresh)));
  Q->overlap_ok = false;
  if (Q->fresh) {  // first packet since the last join
    acquire = Q->fresh_system ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
    Q->fresh = Q->fresh_system = false;
  }
  hsa_signal_t seg_sig{0};
  const uint64_t widx = reserve(Q, &seg_sig);
  // The last launch of a run(): what the join would put behind it -- a barrier packet with a system-scope release and a
  // completion signal -- rides on the packet itself, and the join only waits (one packet less between the last wave and
  // the host).  Not on a segment's last slot, whose signal the run-ahead bound needs; any packet that follows takes the
  // shortcut away again (the join then goes the ordinary way; `fin` fires unobserved and is armed anew next time).
  Q->final_sent = false;
  if (final_packet && seg_sig.handle == 0 && Q->fin_unobserved) {
    // an earlier packet still owns `fin` (it was followed by other packets, so no join waited for it): re-arming the signal
    // under it would let its completion pass for this packet's.  Done already (value <= 0): the signal is free again;
    // otherwise this launch goes without the shortcut and the join takes the barrier path.
    if (H.hsa_signal_load_scacquire(Q->fin) <= 0) Q->fin_unobserved = false;
  }
  if (final_packet && seg_sig.handle == 0 && !Q->fin_unobserved) {
    Q->fin_unobserved = true;
    H.hsa_signal_store_relaxed(Q->fin, 1);
    seg_sig = Q->fin;
    release = HSA_FENCE_SCOPE_SYSTEM;
    Q->final_sent = true;
  }
  // the argument block goes to device memory through the BAR; the read-back of its last word completes the posted
  // writes before the doorbell can make the packet processor fetch them
  char* ka = Q->ring + SLOT * (widx % Q->size);
  memcpy(ka, kernarg, nbytes);
  size_t end = nbytes;
  if (seg > ((nbytes + 7) & ~(size_t)7)) {
    // implicit arguments (code object v5): block counts, group sizes, remainders, global offsets, grid dimensions
    const size_t ib = (nbytes + 7) & ~(size_t)7;
    char im[256];
    memset(im, 0, sizeof(im));
    const uint32_t bc[3] = {grid.x, grid.y, grid.z};
    memcpy(im, bc, 12);
    const uint16_t gs[3] = {(uint16_t)block.x, (uint16_t)block.y, (uint16_t)block.z};
    memcpy(im + 12, gs, 6);
    const uint16_t dims = (uint16_t)(grid.z * block.z > 1 ? 3 : (grid.y * block.y > 1 ? 2 : 1));
    memcpy(im + 64, &dims, 2);
    size_t want = seg - ib;
    if (want > sizeof(im)) want = sizeof(im);
    if (ib + want > SLOT) want = SLOT - ib;
    memcpy(ka + ib, im, want);
    end = ib + want;
  }
  store_fence();
  (void)*reinterpret_cast<volatile uint32_t*>(ka + ((end - 4) & ~(size_t)3));
  hsa_kernel_dispatch_packet_t* pk = static_cast<hsa_kernel_dispatch_packet_t*>(slot_of(Q, widx));
  const int ndim = grid.z * block.z > 1 ? 3 : (grid.y * block.y > 1 ? 2 : 1);
  pk->setup = (uint16_t)(ndim << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
  pk->workgroup_size_x = (uint16_t)block.x, pk->workgroup_size_y = (uint16_t)block.y, pk->workgroup_size_z = (uint16_t)block.z;
  pk->reserved0 = 0;
  pk->grid_size_x = grid.x * block.x, pk->grid_size_y = grid.y * block.y, pk->grid_size_z = grid.z * block.z;
  pk->private_segment_size = k->private_size;
  pk->group_segment_size = k->group_size;
  pk->kernel_object = k->object;
  pk->kernarg_address = ka;
  pk->reserved2 = 0;
  pk->completion_signal = seg_sig;
  publish(Q, widx, (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                              (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE)));
  Q->n_dispatched++;
  return CHX_OK;
}

std::mutex g_name_mu;
std::map<const void*, const AqlKernel*> g_by_host_fn;

}  // namespace

int chx_aql_dispatch(hipStream_t st, const AqlKernel* k, unsigned blocks, const void* kernarg, size_t nbytes, bool release_agent,
                     bool final_packet) {
  if (nbytes < k->kernarg_size) {
    chx_set_error("step queue: argument block of %zu bytes is shorter than the kernel's %u", nbytes, k->kernarg_size);
    return CHX_ERR_INVALID;
  }
  return dispatch_core(st, k, dim3(blocks), dim3(256), kernarg, k->kernarg_size, release_agent ? HSA_FENCE_SCOPE_AGENT : chx_aql_hot_release(),
                       true, final_packet);
}

int chx_aql_dispatch_grid(hipStream_t st, const AqlKernel* k, dim3 grid, const void* kernarg, size_t nbytes, bool release_agent,
                          bool beside_previous, bool leaves_queue) {
  if (nbytes < k->kernarg_size) {
    chx_set_error("step queue: argument block of %zu bytes is shorter than the kernel's %u", nbytes, k->kernarg_size);
    return CHX_ERR_INVALID;
  }
  return dispatch_core(st, k, grid, dim3(256), kernarg, k->kernarg_size, release_agent ? HSA_FENCE_SCOPE_AGENT : chx_aql_hot_release(),
                       true, false, false, beside_previous, leaves_queue);
}

void chx_aql_allow_overlap(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_queues.find(st);
  if (it != g_queues.end() && it->second->open) it->second->overlap_ok = true;
}

int chx_aql_dispatch_any(hipStream_t st, const AqlKernel* k, dim3 grid, dim3 block, const void* kernarg, size_t explicit_bytes,
                         bool hot_acquire) {
  return dispatch_core(st, k, grid, block, kernarg, explicit_bytes, HSA_FENCE_SCOPE_AGENT, false, false, hot_acquire);
}

bool chx_aql_active(hipStream_t st) {
  if (__atomic_load_n(&chx_aql_open_count, __ATOMIC_RELAXED) <= 0 || g_queue_error) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_queues.find(st);
  return it != g_queues.end() && it->second->open;
}

const AqlKernel* chx_aql_kernel_of(const void* host_fn, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_name_mu);
  auto it = g_by_host_fn.find(host_fn);
  if (it != g_by_host_fn.end()) return it->second;
  const char* name = hipKernelNameRefByPtr(host_fn, st);
  const AqlKernel* k = nullptr;
  if (name != nullptr) {
    auto f = g_kernels.find(name);
    if (f != g_kernels.end()) k = &f->second;
  }
  if (k == nullptr && getenv("CHX_AQL_VERBOSE")) fprintf(stderr, "[chx] step queue: no code object holds %s\n", name ? name : "(unnamed kernel)");
  g_by_host_fn[host_fn] = k;
  return k;
}

int chx_aql_open(hipStream_t st) {
  if (!chx_aql_on()) return CHX_OK;
  Queue* Q = queue_for(st);
  if (!Q || g_queue_error) {
    chx_set_error("%s", g_err);
    return CHX_ERR_LAUNCH;
  }
  if (!Q->open) open_gate(Q, st);
  return CHX_OK;
}

void chx_aql_join(hipStream_t st) {
  Queue* Q = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_queues.find(st);
    if (it != g_queues.end()) Q = it->second;
  }
  if (!Q || !Q->open) return;
  Q->overlap_ok = false;
  if (Q->final_sent) {  // (the newest packet is a run's last launch: it releases at system scope and signals by itself)
    (void)wait_zero(Q->fin, "the queue has not drained for 120 s", Q);
    Q->final_sent = false;
    Q->fin_unobserved = false;
  } else {
    H.hsa_signal_store_relaxed(Q->done, 1);
    push_barrier(Q, hsa_signal_t{0}, Q->done, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_SYSTEM);
    (void)wait_zero(Q->done, "the queue has not drained for 120 s", Q);
  }
  Q->open = false;
  Q->n_joins++;
  __atomic_fetch_sub(&chx_aql_open_count, 1, __ATOMIC_RELAXED);
  // The assumption the fence-free packets rest on -- checked when the process opened its first queue -- is checked AGAIN
  // every RECHECK_EVERY packets of a queue, here, with the queue drained (a millisecond per ~5 s of stepping): a card that
  // is shared with other processes, a clock or firmware state that changed under a long run.  One stale word switches
  // every later packet of the process to HIP's agent-scope fences (chx_step_queue_status says so).
  static const uint64_t RECHECK_EVERY = [] {
    const char* e = getenv("CHX_AQL_RECHECK_EVERY");  // (tests: a short period)
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? (uint64_t)v : (1ull << 20);
  }();
  if (g_fence_none && g_selfcheck_result == 0 && Q->n_dispatched - Q->checked_at >= RECHECK_EVERY) {
    Q->checked_at = Q->n_dispatched;
    std::lock_guard<std::mutex> lk(g_mu);  // (the check's buffer and signal are shared by the process's queues)
    selfcheck_fences(Q, st, true);
  }
}

// ---- C ABI (include/chx.h) ----
extern "C" int chx_steps_join(chx_stream_t stream) {
  if (__atomic_load_n(&chx_aql_open_count, __ATOMIC_RELAXED) > 0) chx_aql_join(reinterpret_cast<hipStream_t>(stream));
  return g_queue_error ? CHX_ERR_LAUNCH : CHX_OK;
}
extern "C" int chx_steps_open(chx_stream_t stream) { return chx_aql_open(reinterpret_cast<hipStream_t>(stream)); }
extern "C" int chx_step_queue_info(int64_t* out8_h, chx_stream_t stream) {
  const bool on = chx_aql_on();
  if (out8_h) {
    for (int i = 0; i < 8; ++i) out8_h[i] = 0;
    out8_h[0] = on ? 1 : 0;
    out8_h[1] = on ? (g_fence_none ? 0 : 1) : -1;  // release scope of a steady-state step packet: 0 none, 1 agent
    out8_h[2] = (int64_t)g_kernels.size();
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_queues.find(reinterpret_cast<hipStream_t>(stream));
    if (it != g_queues.end()) {
      out8_h[3] = (int64_t)it->second->n_dispatched;
      out8_h[4] = (int64_t)it->second->n_joins;
      out8_h[5] = (int64_t)it->second->n_gates;
      out8_h[6] = it->second->open ? 1 : 0;
      out8_h[7] = (int64_t)it->second->n_idle_opens;
    }
  }
  return CHX_OK;
}
extern "C" const char* chx_step_queue_status(void) { return g_err; }
extern "C" int chx_step_queue_selfcheck(chx_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!chx_aql_on()) return -1;
  chx_aql_join(st);
  Queue* Q = queue_for(st);
  if (!Q || g_queue_error) return -1;
  if (!g_fence_none) return g_selfcheck_result > 0 ? g_selfcheck_result : -1;  // (fences are on: nothing rests on the assumption)
  g_selfcheck_result = -1;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    selfcheck_fences(Q, st, true);
  }
  Q->checked_at = Q->n_dispatched;
  return g_selfcheck_result;
}
extern "C" int64_t chx_step_queue_selfcheck_runs(void) { return g_selfcheck_runs; }
extern "C" int chx_queue_debug(int32_t delay_us) {
  const int old = g_debug_delay_us;
  if (delay_us >= 0) g_debug_delay_us = delay_us;
  return old;
}
extern "C" int64_t chx_queue_debug_spins(void) { return (int64_t)g_debug_spins; }
