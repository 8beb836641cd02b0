// watch.h -- runtime interrupt and dynamic watchdog for the scan kernels.
//
// Reference: the GPU runtime keeps a `runtime_interrupt_flag` and a cycle budget in module globals, the row loop
// polls them (check_interrupt / dynamic_watchdog, QE/cuda_mapd_rt.cu:105-148; host side QE/GpuInterrupt.cpp,
// QE/DynamicWatchdog.cpp:36-84) and the query ends with ERR_INTERRUPTED / ERR_OUT_OF_TIME.  Here the state is one
// device global, armed per launch by a one-thread kernel on the launch stream (hdk_hip_launch: flags
// HDK_HIP_LAUNCH_CHECK_INTERRUPT, hdk_hip_kernel_options::watchdog_ms) and flipped from the host by
// hdk_hip_set_interrupt on a stream of its own.  Kernels poll once per tile; a launch that asked for neither pays
// one register test per tile.
#pragma once
#include "device_common.h"

namespace hdk {

struct WatchState {
  int32_t interrupt;  // != 0: stop (hdk_hip_set_interrupt)
  uint32_t flags;     // bit 0: poll `interrupt`; bit 1: poll the deadline
  uint64_t deadline;  // s_memrealtime ticks (100 MHz)
};
__device__ WatchState g_watch;

struct Watch {
  uint32_t flags;
  uint64_t deadline;
};

HDK_DEV Watch watch_begin() {
  Watch w;
  w.flags = g_watch.flags;
  w.deadline = g_watch.deadline;
  return w;
}

// 0, or the error the launch has to end with.  Every lane reads the same words.
HDK_DEV int32_t watch_poll(const Watch& w) {
  if ((w.flags & 1u) && __hip_atomic_load(&g_watch.interrupt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    return HDK_HIP_ERR_INTERRUPTED;
  }
  if ((w.flags & 2u) && __builtin_amdgcn_s_memrealtime() > w.deadline) {
    return HDK_HIP_ERR_OUT_OF_TIME;
  }
  return 0;
}

// For tile loops that contain barriers: one thread polls, the block agrees (one extra barrier per tile, paid only
// by launches that asked for the watch).  `s_flag` is a word of LDS.
HDK_DEV int32_t watch_poll_block(const Watch& w, int32_t* s_flag) {
  if (threadIdx.x == 0) {
    *s_flag = watch_poll(w);
  }
  __syncthreads();
  const int32_t r = *s_flag;
  __syncthreads();
  return r;
}

// at the top of a barrier-free tile loop (`for (; tile < ...; tile += gridDim.x)` nested in the fragment loop): leave
// every fragment's loop; whatever follows the loops (slab flush, error record) still runs
#define HDK_WATCH_TILE(watch, err, tile)        \
  if ((watch).flags) {                          \
    if (const int32_t w_ = watch_poll(watch)) { \
      (err) = w_;                               \
      (tile) = INT64_MAX - gridDim.x;           \
      break;                                    \
    }                                           \
  }

__global__ void k_arm_watch(uint32_t flags, uint32_t watchdog_ms) {
  g_watch.flags = flags;
  g_watch.deadline = __builtin_amdgcn_s_memrealtime() + static_cast<uint64_t>(watchdog_ms) * 100000ull;
}

}  // namespace hdk
