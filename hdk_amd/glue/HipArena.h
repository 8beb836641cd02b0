// HipArena.h -- a minimal buffer provider over HipMgr with BufferProvider's method set.
//
// Inside HDK nothing of this is needed: once `HipMgr` is registered in DataMgr::populateDeviceMgrs
// (omniscidb/DataMgr/DataMgr.cpp:193-222) the existing DataMgrBufferProvider serves GPU_LEVEL buffers out of
// GpuBufferMgr's slabs unchanged.  A STANDALONE user of the library (the C++ harness of tests/cpp, a tool that links
// libhdk_hip.so without HDK) has no DataMgr; this class gives it the same surface --
//     free / alloc(memory_level, device_id, num_bytes) / copyToDevice / copyToDeviceAsyncIfPossible / copyToDeviceAsync /
//     synchronizeStream / copyFromDevice / zeroDeviceMem / setDeviceMem / setContext
// (omniscidb/BufferProvider/BufferProvider.h:23-64, same names, same argument order) -- over plain device allocations.
// It does not derive from BufferProvider: that header pulls in DataMgr/AbstractBuffer.h (Logger -> Boost), which this
// image cannot compile; `Buffer` below carries the three things the executor reads from an AbstractBuffer
// (getMemoryPtr, size, getDeviceId).
//
// copyToDeviceAsyncIfPossible stages pageable host memory through a pinned bounce buffer
// (hdk_hip_mgr_allocate_pinned_host_mem = CudaMgr::allocatePinnedHostMem, CudaMgr/CudaMgr.h:120): an asynchronous copy
// from pageable memory is silently synchronous, which is what "if possible" papers over in the reference
// (DataMgrBufferProvider::copyToDeviceAsyncIfPossible falls back to the synchronous copy when the platform cannot).
#pragma once

#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>

#include "HipMgr.h"

namespace hip_mgr {

enum ArenaMemoryLevel { ARENA_DISK_LEVEL = 0, ARENA_CPU_LEVEL = 1, ARENA_GPU_LEVEL = 2 };  // Data_Namespace::MemoryLevel's order

class HipArena {
 public:
  struct Buffer {  // what the executor needs of an AbstractBuffer
    int8_t* getMemoryPtr() const { return ptr; }
    size_t size() const { return bytes; }
    int getDeviceId() const { return device_id; }
    int8_t* ptr{nullptr};
    size_t bytes{0};
    int device_id{0};
    int level{ARENA_GPU_LEVEL};
  };

  explicit HipArena(HipMgr* mgr, size_t pinned_bounce_bytes = 4u << 20) : mgr_(mgr), bounce_bytes_(pinned_bounce_bytes) {}
  HipArena(const HipArena&) = delete;
  HipArena& operator=(const HipArena&) = delete;
  ~HipArena() {
    for (auto& kv : live_) release(kv.second.get());
    for (auto& kv : bounce_) {
      if (kv.second) (void)hdk_hip_mgr_free_pinned_host_mem(kv.second);
    }
  }

  // allocator APIs (GpuAllocator)
  Buffer* alloc(const int memory_level, const int device_id, const size_t num_bytes) {
    auto b = std::make_unique<Buffer>();
    b->bytes = num_bytes ? num_bytes : 8;
    b->device_id = device_id;
    b->level = memory_level;
    if (memory_level == ARENA_GPU_LEVEL) {
      b->ptr = mgr_->allocateDeviceMem(b->bytes, device_id);  // throws HipOutOfMemory -> ERR_OUT_OF_GPU_MEM
    } else {
      check(hdk_hip_mgr_allocate_pinned_host_mem(b->bytes, &b->ptr));  // CPU_LEVEL buffers are pinned: async copies stay async
    }
    Buffer* raw = b.get();
    std::lock_guard<std::mutex> lk(mu_);
    live_[raw] = std::move(b);
    return raw;
  }
  void free(Buffer* buffer) {
    std::unique_ptr<Buffer> own;
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = live_.find(buffer);
      if (it == live_.end()) return;
      own = std::move(it->second);
      live_.erase(it);
    }
    release(own.get());
  }

  void copyToDevice(int8_t* device_ptr, const int8_t* host_ptr, const size_t num_bytes, const int device_id) const {
    mgr_->copyHostToDevice(device_ptr, host_ptr, num_bytes, device_id);
  }
  void copyToDeviceAsync(int8_t* device_ptr, const int8_t* host_ptr, const size_t num_bytes, const int device_id) const {
    mgr_->copyHostToDeviceAsync(device_ptr, host_ptr, num_bytes, device_id);  // host_ptr must stay valid until synchronizeStream
  }
  void copyToDeviceAsyncIfPossible(int8_t* device_ptr, const int8_t* host_ptr, const size_t num_bytes, const int device_id) {
    if (num_bytes == 0) return;
    if (num_bytes > bounce_bytes_) {  // large: a plain (synchronous) copy is the honest answer
      mgr_->copyHostToDevice(device_ptr, host_ptr, num_bytes, device_id);
      return;
    }
    // ONE bounce buffer PER DEVICE: the copy it last carried ran on that device's stream, so waiting for that stream is
    // waiting for the right copy (a buffer shared by all devices would be overwritten under a copy still in flight to
    // another device).  The map is guarded; the wait and the memcpy are not done under the lock of another device's buffer.
    int8_t* bounce = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      int8_t*& slot = bounce_[device_id];
      if (!slot) check(hdk_hip_mgr_allocate_pinned_host_mem(bounce_bytes_, &slot));
      bounce = slot;
    }
    std::lock_guard<std::mutex> dev_lk(bounce_mu_[static_cast<size_t>(device_id) % kBounceLocks]);
    mgr_->synchronizeStream(device_id);  // the previous user of THIS device's bounce buffer is done
    std::memcpy(bounce, host_ptr, num_bytes);
    mgr_->copyHostToDeviceAsync(device_ptr, bounce, num_bytes, device_id);
  }
  void synchronizeStream(const int device_id) const { mgr_->synchronizeStream(device_id); }
  void copyFromDevice(int8_t* host_ptr, const int8_t* device_ptr, const size_t num_bytes, const int device_id) const {
    mgr_->copyDeviceToHost(host_ptr, device_ptr, num_bytes, device_id);
  }
  void zeroDeviceMem(int8_t* device_ptr, const size_t num_bytes, const int device_id) const {
    mgr_->zeroDeviceMem(device_ptr, num_bytes, device_id);
  }
  void setDeviceMem(int8_t* device_ptr, unsigned char uc, const size_t num_bytes, const int device_id) const {
    mgr_->setDeviceMem(device_ptr, uc, num_bytes, device_id);
  }
  void setContext(const int device_id) { mgr_->setContext(device_id); }

  size_t liveBuffers() const {
    std::lock_guard<std::mutex> lk(mu_);
    return live_.size();
  }
  HipMgr* mgr() const { return mgr_; }

 private:
  void release(Buffer* b) {
    if (!b || !b->ptr) return;
    if (b->level == ARENA_GPU_LEVEL) {
      mgr_->freeDeviceMem(b->ptr);
    } else {
      (void)hdk_hip_mgr_free_pinned_host_mem(b->ptr);
    }
    b->ptr = nullptr;
  }
  HipMgr* mgr_;
  size_t bounce_bytes_;
  static constexpr size_t kBounceLocks = 16;
  std::unordered_map<int, int8_t*> bounce_;   // device id -> its pinned bounce buffer (under mu_)
  std::mutex bounce_mu_[kBounceLocks];        // one user at a time per device's buffer
  mutable std::mutex mu_;
  std::unordered_map<Buffer*, std::unique_ptr<Buffer>> live_;
};

}  // namespace hip_mgr
