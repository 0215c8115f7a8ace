// libchx: one launch macro for the kernels that may ride on the step queue (chx_aql.h).  Internal, not exported.
//
//   CHX_LAUNCH(kernel, grid, block, stream, args...)
//
// launches `kernel` through hipLaunchKernelGGL -- unless packets are outstanding on the step queue paired with `stream`
// (a run() is in progress): then the launch is written as an AQL packet behind them, so that the caller's one-stream
// picture holds without a join.  The argument block is laid out from the KERNEL's parameter types (each argument
// converted to its parameter's type, placed at that type's alignment -- what the compiler's kernarg segment does for the
// plain structs, pointers and scalars these kernels take), the implicit arguments are appended by the dispatcher.
#pragma once
#include <cstring>
#include <type_traits>

#include "chx_aql.h"
#include "chx_common.h"

namespace chx_launch_detail {
template <typename T>
inline size_t put(char* buf, size_t off, const T& v) {
  static_assert(std::is_trivially_copyable<T>::value, "kernel arguments are plain data");
  off = (off + alignof(T) - 1) & ~(alignof(T) - 1);
  std::memcpy(buf + off, &v, sizeof(T));
  return off + sizeof(T);
}
inline size_t pack(char*, size_t off) { return off; }
template <typename T, typename... Rest>
inline size_t pack(char* buf, size_t off, const T& v, const Rest&... rest) {
  return pack(buf, put(buf, off, v), rest...);
}
}  // namespace chx_launch_detail

template <bool HOT = false, typename... KArgs, typename... Args>
inline int chx_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
  if (chx_aql_active(st)) {
    const AqlKernel* k = chx_aql_kernel_of(reinterpret_cast<const void*>(kernel), st);
    if (k != nullptr) {
      alignas(16) char buf[768];
      static_assert((0 + ... + (sizeof(KArgs) + 16)) <= sizeof(buf), "argument block too large for the packer's buffer");
      std::memset(buf, 0, sizeof(buf));
      const size_t n = chx_launch_detail::pack(buf, 0, static_cast<KArgs>(args)...);
      const int rc = chx_aql_dispatch_any(st, k, grid, block, buf, n, HOT);
      chx_note_launch_failure(rc);  // (most call sites do not look at the return value: chx_check_launch reports it)
      return rc;
    }
    chx_aql_join(st);  // (a kernel no loaded code object holds: keep the order at least, then launch through HIP)
  }
  hipLaunchKernelGGL(kernel, grid, block, 0, st, static_cast<KArgs>(args)...);
  return CHX_OK;
}
#define CHX_LAUNCH(kernel, grid, block, st, ...) chx_launch(kernel, grid, block, st, __VA_ARGS__)
// the same for a kernel on a microsecond budget whose inputs its predecessors on the queue wrote through or released: as a
// packet it goes without an acquire fence (the Monte Carlo trial)
#define CHX_LAUNCH_HOT(kernel, grid, block, st, ...) chx_launch<true>(kernel, grid, block, st, __VA_ARGS__)
