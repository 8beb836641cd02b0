// host_common.h -- host-side plumbing shared by the API translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/hdk_hip.h"

namespace hdk {

// thread-local message for hdk_hip_last_error(); no exception ever crosses the ABI
void set_error(const char* fmt, ...);
void clear_error();
// per-device state (lazy): sets the device and returns the stream to use (`stream` or the manager's)
int32_t device_enter(int32_t device_id, void* stream, hipStream_t* out);
const hdk_hip_device_properties* device_props(int32_t device_id);
// the device's interrupt word (device memory, 0 = run): polled by launches with HDK_HIP_LAUNCH_CHECK_INTERRUPT
const int32_t* device_interrupt_word(int32_t device_id);
// init_groups.hip: the row-wise fill for a buffer named by GROUPBY_BUF[0] (device memory)
int32_t launch_init_row_wise_indirect(int64_t* const* groupby_buf, const int64_t* init_vals, uint32_t entry_count,
                                      uint32_t key_count, uint32_t key_width, uint32_t row_size_quad, int keyless,
                                      const hdk_hip_device_properties* props, hipStream_t s);

#define HDK_HIP_CHECK(expr)                                                                  \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      hdk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? HDK_HIP_ERR_OUT_OF_GPU_MEM : HDK_HIP_ERR_RUNTIME;   \
    }                                                                                        \
  } while (0)

#define HDK_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      hdk::set_error(__VA_ARGS__);      \
      return HDK_HIP_ERR_INVALID_ARG;   \
    }                                   \
  } while (0)

// Stream-ordered scratch of a launch: handed back with hipFreeAsync on EVERY way out of the scope, error returns
// included (the free is ordered after the kernels already enqueued on the stream).
struct AsyncScratch {
  void* p = nullptr;
  hipStream_t s = nullptr;
  explicit AsyncScratch(hipStream_t stream) : s(stream) {}
  AsyncScratch(const AsyncScratch&) = delete;
  AsyncScratch& operator=(const AsyncScratch&) = delete;
  ~AsyncScratch() {
    if (p) (void)hipFreeAsync(p, s);
  }
};

}  // namespace hdk
