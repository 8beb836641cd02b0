// HipMgr.h -- the HDK-side binding of the device manager: `class HipMgr : public GpuMgr`.
//
// This is the file a maintainer adds next to omniscidb/CudaMgr/ and omniscidb/L0Mgr/.  It is written
// against the reference's own interface header (omniscidb/DataMgr/GpuMgr.h:29-79) and forwards every
// virtual to the C ABI of libhdk_hip.so (include/hdk_hip.h, hdk_hip_mgr_*).  It is compile-checked
// against /root/reference where that tree exists (tests/test_glue_compiles.py); it cannot be linked
// into HDK here because HDK itself cannot be built in this image (Boost/TBB/LLVM absent).
//
// Requires one enumerator added to omniscidb/Shared/GpuPlatform.h:22:
//     enum GpuMgrPlatform { CUDA, L0, HIP };
#pragma once

#include <stdexcept>
#include <string>

#include "DataMgr/GpuMgr.h"  // reference header
#include "hdk_hip.h"

#ifndef HDK_GPU_PLATFORM_HIP
#define HDK_GPU_PLATFORM_HIP static_cast<GpuMgrPlatform>(2)  // GpuMgrPlatform::HIP once the enum has it
#endif

namespace hip_mgr {

class HipOutOfMemory : public DeviceException {  // cf. CudaMgr's CudaErrorException::isOutOfMemory
 public:
  explicit HipOutOfMemory(const std::string& msg) : DeviceException(msg) {}
  bool isOutOfMemory() const override { return true; }
};

inline void check(int32_t status) {
  if (status == HDK_HIP_OK) {
    return;
  }
  const std::string msg = std::string("HipMgr: ") + hdk_hip_last_error();
  if (status == HDK_HIP_ERR_OUT_OF_GPU_MEM) {
    throw HipOutOfMemory(msg);  // Executor maps it to ERR_OUT_OF_GPU_MEM (QE/Execute.cpp:3426-3427)
  }
  throw DeviceException(msg);
}

class HipMgr : public GpuMgr {
 public:
  explicit HipMgr(const int num_gpus = -1, const int start_gpu = 0) : start_gpu_(start_gpu) {
    int32_t n = 0;
    check(hdk_hip_mgr_get_device_count(&n));
    device_count_ = num_gpus < 0 ? n - start_gpu : num_gpus;
    if (device_count_ <= 0 || start_gpu + device_count_ > n) {
      throw DeviceException("HipMgr: no usable HIP device");
    }
    for (int d = 0; d < device_count_; ++d) {
      hdk_hip_device_properties p;
      check(hdk_hip_mgr_get_device_properties(start_gpu_ + d, &p));
      if (d == 0) {
        props0_ = p;
        min_cu_ = p.num_cu;
        min_smem_ = p.shared_mem_per_block;
      } else {
        min_cu_ = p.num_cu < min_cu_ ? p.num_cu : min_cu_;
        min_smem_ = p.shared_mem_per_block < min_smem_ ? p.shared_mem_per_block : min_smem_;
      }
    }
  }

  void copyHostToDevice(int8_t* device_ptr, const int8_t* host_ptr, const size_t num_bytes,
                        const int device_num) override {
    check(hdk_hip_mgr_copy_host_to_device(device_ptr, host_ptr, num_bytes, start_gpu_ + device_num));
  }
  void copyHostToDeviceAsync(int8_t* device_ptr, const int8_t* host_ptr, const size_t num_bytes,
                             const int device_num) override {
    check(hdk_hip_mgr_copy_host_to_device_async(device_ptr, host_ptr, num_bytes, start_gpu_ + device_num));
  }
  void synchronizeStream(const int device_num) override {
    check(hdk_hip_mgr_synchronize_stream(start_gpu_ + device_num));
  }
  void copyDeviceToHost(int8_t* host_ptr, const int8_t* device_ptr, const size_t num_bytes,
                        const int device_num) override {
    check(hdk_hip_mgr_copy_device_to_host(host_ptr, device_ptr, num_bytes, start_gpu_ + device_num));
  }
  void copyDeviceToDevice(int8_t* dest_ptr, int8_t* src_ptr, const size_t num_bytes, const int dest_device_num,
                          const int src_device_num) override {
    check(hdk_hip_mgr_copy_device_to_device(dest_ptr, src_ptr, num_bytes, start_gpu_ + dest_device_num,
                                            start_gpu_ + src_device_num));
  }
  void zeroDeviceMem(int8_t* device_ptr, const size_t num_bytes, const int device_num) override {
    check(hdk_hip_mgr_zero_device_mem(device_ptr, num_bytes, start_gpu_ + device_num));
  }
  void setDeviceMem(int8_t* device_ptr, const unsigned char uc, const size_t num_bytes,
                    const int device_num) override {
    check(hdk_hip_mgr_set_device_mem(device_ptr, uc, num_bytes, start_gpu_ + device_num));
  }
  int8_t* allocateDeviceMem(const size_t num_bytes, const int device_num) override {
    int8_t* p = nullptr;
    check(hdk_hip_mgr_allocate_device_mem(num_bytes, start_gpu_ + device_num, &p));
    return p;
  }
  void freeDeviceMem(int8_t* device_ptr) override { check(hdk_hip_mgr_free_device_mem(device_ptr)); }
  void setContext(const int device_num) const override { check(hdk_hip_mgr_set_context(start_gpu_ + device_num)); }
  void synchronizeDevices() const override { check(hdk_hip_mgr_synchronize_devices()); }
  int getDeviceCount() const override { return device_count_; }
  GpuMgrPlatform getPlatform() const override { return HDK_GPU_PLATFORM_HIP; }
  size_t getTotalMem(const int device_num) const override {
    hdk_hip_device_properties p;
    check(hdk_hip_mgr_get_device_properties(start_gpu_ + device_num, &p));
    return p.global_mem;
  }
  uint32_t getMaxBlockSize() const override { return static_cast<uint32_t>(props0_.max_threads_per_block); }
  int8_t getSubGroupSize() const override { return static_cast<int8_t>(props0_.wavefront_size); }  // 64
  uint32_t getGridSize() const override { return static_cast<uint32_t>(props0_.grid_size); }
  uint32_t getMinEUNumForAllDevices() const override { return static_cast<uint32_t>(min_cu_); }
  bool hasSharedMemoryAtomicsSupport() const override { return true; }
  bool canLoadAsync() const override { return true; }
  bool hasFP64Support() const override { return true; }
  size_t getMinSharedMemoryPerBlockForAllDevices() const override { return min_smem_; }

  int getStartGpu() const { return start_gpu_; }

 private:
  int start_gpu_;
  int device_count_{0};
  hdk_hip_device_properties props0_{};
  int min_cu_{0};
  size_t min_smem_{0};
};

}  // namespace hip_mgr
