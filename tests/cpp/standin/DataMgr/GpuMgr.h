// Stand-in for the reference's omniscidb/DataMgr/GpuMgr.h:23-79 (+ Shared/GpuPlatform.h:21), used ONLY where the
// reference tree is absent (the GPU box): same class names, virtuals and signatures, so that hdk_amd/glue/HipMgr.h
// compiles unchanged.  Where /root/reference exists the harness is built against the real header instead
// (tests/cpp/Makefile) -- that build is the one that proves HipMgr overrides the reference's interface.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

enum GpuMgrPlatform { CUDA, L0 };

class DeviceException : public std::runtime_error {
 public:
  DeviceException(const std::string& msg) : std::runtime_error(msg) {}
  virtual bool isOutOfMemory() const { return false; }
};

struct GpuMgr {
  virtual ~GpuMgr() = default;
  virtual void copyHostToDevice(int8_t* device_ptr, const int8_t* host_ptr, const size_t num_bytes, const int device_num) = 0;
  virtual void copyHostToDeviceAsync(int8_t* device_ptr, const int8_t* host_ptr, const size_t num_bytes, const int device_num) = 0;
  virtual void synchronizeStream(const int device_num) = 0;
  virtual void copyDeviceToHost(int8_t* host_ptr, const int8_t* device_ptr, const size_t num_bytes, const int device_num) = 0;
  virtual void copyDeviceToDevice(int8_t* dest_ptr, int8_t* src_ptr, const size_t num_bytes, const int dest_device_num,
                                  const int src_device_num) = 0;
  virtual void zeroDeviceMem(int8_t* device_ptr, const size_t num_bytes, const int device_num) = 0;
  virtual void setDeviceMem(int8_t* device_ptr, const unsigned char uc, const size_t num_bytes, const int device_num) = 0;
  virtual int8_t* allocateDeviceMem(const size_t num_bytes, const int device_num) = 0;
  virtual void freeDeviceMem(int8_t* device_ptr) = 0;
  virtual void setContext(const int device_num) const = 0;
  virtual void synchronizeDevices() const = 0;
  virtual int getDeviceCount() const = 0;
  virtual GpuMgrPlatform getPlatform() const = 0;
  virtual size_t getTotalMem(const int device_num) const = 0;
  virtual uint32_t getMaxBlockSize() const = 0;
  virtual int8_t getSubGroupSize() const = 0;
  virtual uint32_t getGridSize() const = 0;
  virtual uint32_t getMinEUNumForAllDevices() const = 0;
  virtual bool hasSharedMemoryAtomicsSupport() const = 0;
  virtual bool canLoadAsync() const = 0;
  virtual bool hasFP64Support() const { return true; }
  virtual size_t getMinSharedMemoryPerBlockForAllDevices() const = 0;
};
