// HipKernel.h -- HDK-side kernel object for GpuMgrPlatform::HIP: `DeviceKernel` over the fixed kernel library
// (reference interface: omniscidb/QueryEngine/DeviceKernel.h:25-65; CUDA twin: CudaDeviceKernel,
// QE/DeviceKernel.cpp:45-160; factory create_device_kernel, :211-225).
//
// Include AFTER the reference's "QueryEngine/DeviceKernel.h" (which declares DeviceKernel, DeviceClock,
// KernelOptions, CompilationContext); this header only adds the HIP classes.  The JIT'ed module of the CUDA
// path is replaced by a POD plan: `HipPlanContext` is the CompilationContext that NativeCodegen returns for the
// HIP platform (QE/NativeCodegen.cpp:1403-1461 is where the platform switch sits), produced by pattern-matching
// the RelAlgExecutionUnit; shapes outside the library throw QueryMustRunOnCpu there.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "hdk_hip.h"

namespace hip_rt {

class HipPlanContext : public CompilationContext {
 public:
  explicit HipPlanContext(const hdk_hip_plan& p) : plan(p) {}
  hdk_hip_plan plan;
};

// DeviceClock over a pair of HIP events is not needed: the library brackets the scan kernel itself when asked
// (HDK_HIP_LAUNCH_RECORD_EVENTS) and hands the elapsed time back through hdk_hip_collect_scan_times.
class HipDeviceClock : public DeviceClock {
 public:
  explicit HipDeviceClock(int device_id) : device_id_(device_id) {}
  void start() override {
    int32_t n = 0;
    (void)hdk_hip_collect_scan_times(device_id_, nullptr, 0, &n);  // drop what was recorded before
  }
  int stop() override {  // milliseconds, like CudaDeviceClock::stop (QE/DeviceKernel.cpp:33-41)
    float ms[16];
    int32_t n = 0;
    if (hdk_hip_collect_scan_times(device_id_, ms, 16, &n) != HDK_HIP_OK) {
      return 0;
    }
    float total = 0.f;
    for (int32_t i = 0; i < n && i < 16; ++i) {
      total += ms[i];
    }
    return static_cast<int>(total);
  }

 private:
  int device_id_;
};

// Scratch the library asks for per launch (plan copy + per-block slabs); HDK passes its per-kernel
// GpuAllocator (DataMgr/Allocators/GpuAllocator.h), whose buffers live until the kernel's results are copied.
struct HipWorkspaceAllocator {
  virtual int8_t* alloc(size_t num_bytes) = 0;
  virtual ~HipWorkspaceAllocator() = default;
};

class HipKernel : public DeviceKernel {
 public:
  // default_grid: what HipMgr::getGridSize() answers; Executor::gridSize() hands that back through KernelOptions unless
  // exec.override_gpu_grid_size is set (QE/Execute.cpp gridSize, QE/QueryExecutionContext.cpp:307-312)
  HipKernel(const HipPlanContext* ctx, int device_id, HipWorkspaceAllocator* allocator, uint64_t total_rows = 0,
            bool timed = false, unsigned default_grid = 0)
      : ctx_(ctx), device_id_(device_id), allocator_(allocator), total_rows_(total_rows), timed_(timed),
        default_grid_(default_grid) {}

  // kernelParams: the 12 device pointers of QueryExecutionContext::prepareKernelParams
  // (QE/QueryExecutionContext.h:111-125), same order, same contents
  void launch(const KernelOptions& ko, std::vector<int8_t*>& kernelParams) override {
    if (kernelParams.size() != static_cast<size_t>(HDK_KP_COUNT)) {
      throw std::runtime_error("HipKernel: expected the 12-pointer kernel parameter block");
    }
    hdk_hip_kernel_options o{};
    // The library sizes its persistent grids per kernel (blocks that are resident); a grid HDK was
    // explicitly configured with (anything but the manager's own answer) is passed on.  ko.literalsOffset /
    // ko.hoistLiterals have nothing to act on: literals live in the plan, kernelParams[LITERALS] is not read.
    o.grid_dim_x = (default_grid_ && ko.gridDimX != default_grid_) ? ko.gridDimX : 0;
    o.block_dim_x = ko.blockDimX;
    o.shared_mem_bytes = ko.sharedMemBytes;
    o.flags = (timed_ ? HDK_HIP_LAUNCH_RECORD_EVENTS : 0u) | (interruptible_ ? HDK_HIP_LAUNCH_CHECK_INTERRUPT : 0u);
    o.total_rows = total_rows_;
    o.watchdog_ms = watchdog_ms_;
    size_t ws_bytes = 0;
    check(hdk_hip_workspace_size(&ctx_->plan, &o, device_id_, &ws_bytes));
    int8_t* ws = allocator_->alloc(ws_bytes);
    check(hdk_hip_launch(&ctx_->plan, kernelParams.data(), &o, device_id_, /*stream=*/nullptr, ws, ws_bytes));
  }

  // QueryExecutionContext::launchGpuCode calls these before launch() when the watchdog / interrupt are enabled
  // (QE/QueryExecutionContext.cpp:314-340).  time_limit is the budget in ms (g_dynamic_watchdog_time_limit); the
  // CUDA path converts it to cycles, the library counts the 100 MHz realtime clock itself.
  void initializeDynamicWatchdog(bool could_interrupt, uint64_t /*cycle_budget*/, size_t time_limit) override {
    watchdog_ms_ = static_cast<uint32_t>(time_limit);
    interruptible_ = interruptible_ || could_interrupt;
  }
  void initializeRuntimeInterrupter() override {
    check(hdk_hip_set_interrupt(device_id_, 0));  // re-arm: Executor::interrupt() raises it (hdk_hip_set_interrupt(dev, 1))
    interruptible_ = true;
  }

  std::unique_ptr<DeviceClock> make_clock() override { return std::make_unique<HipDeviceClock>(device_id_); }

 private:
  static void check(int32_t status) {
    if (status != HDK_HIP_OK) {
      throw std::runtime_error(std::string("hdk_hip: ") + hdk_hip_last_error());
    }
  }
  const HipPlanContext* ctx_;
  int device_id_;
  HipWorkspaceAllocator* allocator_;
  uint64_t total_rows_;
  bool timed_;
  unsigned default_grid_;
  uint32_t watchdog_ms_{0};
  bool interruptible_{false};
};

// the case create_device_kernel (QE/DeviceKernel.cpp:211-225) gains:
//   case GpuMgrPlatform::HIP:
//     return std::make_unique<hip_rt::HipKernel>(dynamic_cast<const hip_rt::HipPlanContext*>(ctx), device_id, alloc);

}  // namespace hip_rt
