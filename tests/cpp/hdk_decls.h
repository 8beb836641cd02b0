// Declarations shaped like the reference's, for the interface headers that cannot be included here because they pull
// in LLVM / Boost (absent from this image):
//   QueryEngine/DeviceKernel.h:25-65        DeviceClock, KernelOptions, DeviceKernel, create_device_kernel
//   QueryEngine/CompilationContext.h:23-26  CompilationContext  (the header includes ExecutionEngineWrapper.h -> LLVM)
//   QueryEngine/JoinHashTable/Runtime/HashJoinRuntime.h:43-57,100-124  HashEntryInfo, ColumnType, JoinChunk,
//                                           JoinColumn, JoinColumnTypeInfo  (the header includes Logger.h -> Boost)
//   ResultSet/ResultType.h:28-34            QueryDescriptionType
// Same names, members, member order and signatures; nothing else.  DataMgr/GpuMgr.h is NOT restated anywhere: the
// harness is built against the reference's own header (tests/cpp/Makefile).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "DataMgr/GpuMgr.h"

class CompilationContext {
 public:
  virtual ~CompilationContext() {}
};

class DeviceClock {
 public:
  virtual void start() = 0;
  virtual int stop() = 0;
  virtual ~DeviceClock() = default;
};

struct KernelOptions {
  unsigned int gridDimX = 1;
  unsigned int gridDimY = 1;
  unsigned int gridDimZ = 1;
  unsigned int blockDimX = 1;
  unsigned int blockDimY = 1;
  unsigned int blockDimZ = 1;
  unsigned int sharedMemBytes = 0;
  unsigned int literalsOffset = 0;
  bool hoistLiterals = true;
};

class DeviceKernel {
 public:
  virtual void launch(const KernelOptions& ko, std::vector<int8_t*>& kernelParams) = 0;
  virtual void initializeDynamicWatchdog(bool could_interrupt, uint64_t cycle_budget, size_t time_limit) {}
  virtual void initializeRuntimeInterrupter() {}
  virtual std::unique_ptr<DeviceClock> make_clock() = 0;
  virtual ~DeviceKernel() = default;
};

std::unique_ptr<DeviceKernel> create_device_kernel(const CompilationContext* ctx, GpuMgrPlatform platform, int device_id);

enum class QueryDescriptionType { GroupByPerfectHash, GroupByBaselineHash, Projection, NonGroupedAggregate, Estimator };

struct HashEntryInfo {
  alignas(sizeof(int64_t)) size_t hash_entry_count;
  alignas(sizeof(int64_t)) int64_t bucket_normalization;
};

enum ColumnType { SmallDate = 0, Signed = 1, Unsigned = 2, Double = 3 };

struct JoinChunk {
  const int8_t* col_buff;
  size_t num_elems;
  size_t row_id;
};

struct JoinColumn {
  const int8_t* col_chunks_buff;
  size_t col_chunks_buff_sz;
  size_t num_chunks;
  size_t num_elems;
  size_t elem_sz;
};

struct JoinColumnTypeInfo {
  const size_t elem_sz;
  const int64_t min_val;
  const int64_t max_val;
  const int64_t null_val;
  const bool uses_bw_eq;
  const int64_t translated_null_val;
  const ColumnType column_type;
};
