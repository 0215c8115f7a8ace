// libchx: the step queue -- fused step launches dispatched as hand-written AQL packets on a private HSA queue
// (chx_aql.hip).  Internal, not exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct AqlKernel {
  uint64_t object;  // kernel descriptor address in the loaded code object
  uint32_t kernarg_size, group_size, private_size;
};

// true once the step queue is usable in this process (CHX_AQL=0 turns it off; a failed initialisation turns it off
// for good and the launches go through hipLaunchKernelGGL as before)
bool chx_aql_on();
// the kernel whose demangled name starts with `prefix` ("void k_lj_step<4, true, false, false>"), or NULL
const AqlKernel* chx_aql_kernel(const char* prefix);
// release fence of a steady-state step launch whose outputs all leave as write-through stores: 0 = none, 1 = agent
int chx_aql_hot_release();
// One kernel dispatch packet behind everything enqueued so far on `st` (HIP launches included: the first packet after
// a join waits for a marker launched on `st`) and on the step queue paired with it.  `release_agent`: the packet ends with
// an agent-scope release (a launch that leaves plain stores behind); otherwise with the scope chx_aql_hot_release() names.
// `final_packet`: the caller knows this to be the last launch of a run (the launch that also writes the state in user order):
// it carries the join's completion signal and system-scope release itself, and a join that follows at once only waits.
int chx_aql_dispatch(hipStream_t st, const AqlKernel* k, unsigned blocks, const void* kernarg, size_t nbytes, bool release_agent,
                     bool final_packet = false);
// The same for a two-dimensional grid of 256-thread workgroups (the batched step launch: x = a replica's blocks, y = the
// replica).  `beside_previous`: no barrier bit -- the packet may start while the one dispatched just before it runs (the
// classes of one batched step); it still stays behind everything that one waited for.  `leaves_queue`: the launch writes
// something that is read from outside the queue (the batch of trailing kicks: the states run_batch returns) -- what the
// debug mode delays.
int chx_aql_dispatch_grid(hipStream_t st, const AqlKernel* k, dim3 grid, const void* kernarg, size_t nbytes, bool release_agent,
                          bool beside_previous, bool leaves_queue = false);
// Host wait until every packet dispatched for `st` has completed (no-op when none is outstanding).  Called by every
// libchx entry point before it launches on `st` (chx_stream) and by the Python layer where it used to wait for the stream.
void chx_aql_join(hipStream_t st);
// number of step queues with packets outstanding (fast path of chx_stream)
extern int chx_aql_open_count;

// ---- any libchx kernel through the queue (round 5, second half: the whole of run() rides on it) ----
// Packets are outstanding for `st`:
// a launch that goes to the queue now stays behind them without a join.
bool chx_aql_active(hipStream_t st);
// the kernel whose host-side stub is `host_fn` (hipKernelNameRefByPtr -> the code objects' symbol table), or NULL
const AqlKernel* chx_aql_kernel_of(const void* host_fn, hipStream_t st);
// One packet for any grid; `explicit_bytes` of packed arguments (the implicit arguments of code object v5 -- block
// counts, group sizes, grid dimensions -- are appended when the kernel's segment asks for them); agent-scope release, and
// an agent-scope acquire unless `hot_acquire` (the Monte Carlo trial: the steps' acquire-less rule, chx_aql.hip).
int chx_aql_dispatch_any(hipStream_t st, const AqlKernel* k, dim3 grid, dim3 block, const void* kernarg, size_t explicit_bytes,
                         bool hot_acquire = false);
// The packet just written for `st` -- the list build's summary launch, which only reads what the step kernels read and
// writes the build's status block -- may run BESIDE the next step packet: that one goes out without the barrier bit (it
// still starts behind everything in front of the summary, whose own barrier bit held it back until then).  Any other
// dispatch in between takes the permission away.
void chx_aql_allow_overlap(hipStream_t st);
// open the queue for `st` (gate behind the stream's work) without dispatching anything: what follows on the queue-capable
// entry points then rides on the queue
int chx_aql_open(hipStream_t st);
