// watch.h -- runtime interrupt and dynamic watchdog for the scan kernels.
//
// Reference: the GPU runtime keeps a `runtime_interrupt_flag` and a cycle budget in module globals, the row loop
// polls them (check_interrupt / dynamic_watchdog, QE/cuda_mapd_rt.cu:105-148; host side QE/GpuInterrupt.cpp,
// QE/DynamicWatchdog.cpp:36-84) and the query ends with ERR_INTERRUPTED / ERR_OUT_OF_TIME.  The reference keeps that
// state per module, i.e. per query; here `flags` and `deadline` belong to the LAUNCH: they live in the head of the
// launch's workspace, right behind the plan copy (LaunchWatch at kWatchOffset), written on the launch stream -- zeros
// together with the plan upload, or by the one-thread kernel k_arm_watch when the launch asks for the interrupt poll
// (HDK_HIP_LAUNCH_CHECK_INTERRUPT) or a time budget (hdk_hip_kernel_options::watchdog_ms), and always for
// PLAN_RESIDENT launches (whose workspace may still hold the words of an earlier launch; a kernel node when the
// launch is captured into a hipGraph, so that a replay re-arms itself).  Launches on other streams or with other
// workspaces neither see nor move them.  Only the `interrupt` word is per device (Executor::interrupt stops whatever
// runs there); hdk_hip_set_interrupt flips it from a stream of its own.  Kernels poll once per tile; a launch that
// asked for neither pays one register test per tile.
#pragma once
#include "device_common.h"

namespace hdk {

struct LaunchWatch {   // per launch, at kWatchOffset of its workspace
  uint32_t flags;      // bit 0: poll `interrupt`; bit 1: poll the deadline
  uint32_t pad_;
  uint64_t deadline;   // s_memrealtime ticks (100 MHz)
};

struct Watch {
  uint32_t flags;
  uint64_t deadline;
  const int32_t* interrupt;  // the device's interrupt word (runtime.hip: device_interrupt_word)
};

HDK_DEV Watch watch_begin(const KernParams& kp) {
  Watch w;
  w.flags = kp.watch->flags;
  w.deadline = kp.watch->deadline;
  w.interrupt = kp.interrupt;
  return w;
}

// 0, or the error the launch has to end with.  Every lane reads the same words.
HDK_DEV int32_t watch_poll(const Watch& w) {
  if ((w.flags & 1u) && __hip_atomic_load(w.interrupt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
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

}  // namespace hdk
