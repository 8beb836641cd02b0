// HipReduce.h -- the multi-device merge of partial results in ONE process, compiled against RCCL.
//
// The reference runs one host thread per execution kernel (tbb::task_group, QE/Execute.cpp:2776-2788), serialises per
// device (gpu_exec_mutex_, QE/ExecutionKernel.cpp:182-186), copies every device's partial ResultSet to the HOST and reduces
// there (Executor::reduceMultiDeviceResults / reduceMultiDeviceResultSets, QE/Execute.cpp:1224-1336,2606-2641).  With the
// partial buffers resident in HBM the same step stays on the devices:
//
//   all_gather_and_fold   GroupByPerfectHash / NonGroupedAggregate: dense buffers, a few KB each -- ncclAllGather of the
//                         per-device buffers, then hdk_hip_reduce_buffers (ResultSetReduction's slot-wise agg_*[_skip_val],
//                         QE/ResultSetReduction.cpp:1234-1330) folds the other devices' into each device's own: every
//                         device ends with the merged buffer, as after an all-reduce, but with the exact NULL rules an
//                         ncclSum cannot give;
//   exchange_tuples       GroupByBaselineHash: no table leaves a device.  hdk_hip_scatter_to_owners puts every ROW's tuple
//                         into the segment of the device that owns its key (owner = key_hash % G), one grouped
//                         ncclSend / ncclRecv all-to-all with EQUAL splits moves the segments over xGMI, and
//                         hdk_hip_aggregate_from_ranks builds each owner's table from what it received -- disjoint key
//                         sets, every group on the reference's probe sequence (reduceOneEntryBaseline's re-insert,
//                         QE/ResultSetReduction.cpp:694-731, done once per row instead of once per partial entry).
//
// One communicator and one stream per device (ncclCommInitAll: the single-process form), one host thread per device for the
// duration of a step; nothing here ever replaces the process image, and any HIP / RCCL / library failure is an exception
// (the caller's main() exits non-zero).  Python twin: hdk_amd/distributed.py (torch.distributed, one process per GPU).
#pragma once
#include <rccl/rccl.h>  // (hip_runtime_api.h comes with it)

#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "hdk_hip.h"

namespace hip_rt {

inline void rccl_check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) {
    throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
  }
}
inline void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
}
inline void lib_check(int32_t st, const char* what) {
  if (st != HDK_HIP_OK) {
    throw std::runtime_error(std::string(what) + ": " + hdk_hip_last_error());
  }
}

// the devices of one node as RCCL sees them from a single process
class HipDeviceGroup {
 public:
  explicit HipDeviceGroup(const std::vector<int>& devices) : devices_(devices), comms_(devices.size()), streams_(devices.size()) {
    if (devices.empty()) {
      throw std::runtime_error("HipDeviceGroup: no devices");
    }
    rccl_check(ncclCommInitAll(comms_.data(), static_cast<int>(devices_.size()), devices_.data()), "ncclCommInitAll");
    for (size_t i = 0; i < devices_.size(); ++i) {
      hip_check(hipSetDevice(devices_[i]), "hipSetDevice");
      hip_check(hipStreamCreateWithFlags(&streams_[i], hipStreamNonBlocking), "hipStreamCreate");
    }
  }
  HipDeviceGroup(const HipDeviceGroup&) = delete;
  HipDeviceGroup& operator=(const HipDeviceGroup&) = delete;
  ~HipDeviceGroup() {
    for (size_t i = 0; i < devices_.size(); ++i) {
      (void)hipSetDevice(devices_[i]);
      (void)hipStreamSynchronize(streams_[i]);
      (void)hipStreamDestroy(streams_[i]);
      (void)ncclCommDestroy(comms_[i]);
    }
  }
  int size() const { return static_cast<int>(devices_.size()); }
  int device(int i) const { return devices_[static_cast<size_t>(i)]; }
  ncclComm_t comm(int i) const { return comms_[static_cast<size_t>(i)]; }
  hipStream_t stream(int i) const { return streams_[static_cast<size_t>(i)]; }

  // fn(i) on one host thread per device, each with its device current; the first failure is rethrown after all have ended
  template <class F>
  void parallel(F fn) const {
    std::vector<std::thread> threads;
    std::exception_ptr first;
    std::mutex mu;
    for (int i = 0; i < size(); ++i) {
      threads.emplace_back([&, i] {
        try {
          hip_check(hipSetDevice(devices_[static_cast<size_t>(i)]), "hipSetDevice");
          fn(i);
        } catch (...) {
          std::lock_guard<std::mutex> lk(mu);
          if (!first) first = std::current_exception();
        }
      });
    }
    for (auto& t : threads) t.join();
    if (first) std::rethrow_exception(first);
  }

 private:
  std::vector<int> devices_;
  std::vector<ncclComm_t> comms_;
  std::vector<hipStream_t> streams_;
};

// what device i brings to all_gather_and_fold
struct PartialBuffer {
  int64_t* buf;        // its partial result, `quads` words: merged in place
  int64_t* gathered;   // scratch, size() x quads words
  int32_t* dev_error;  // device word for the fold's error code
};

// Every device ends with the fold of all partial buffers in `buf` (entry-wise ResultSetReduction).  init_vals: the HOST
// array the buffers were initialised from (hdk_hip_reduce_buffers).
inline void all_gather_and_fold(const HipDeviceGroup& g, const hdk_hip_plan& plan, uint32_t entry_count, size_t quads,
                                const std::vector<PartialBuffer>& parts, const int64_t* init_vals) {
  const int G = g.size();
  if (static_cast<int>(parts.size()) != G) {
    throw std::runtime_error("all_gather_and_fold: one PartialBuffer per device");
  }
  g.parallel([&](int i) {
    const PartialBuffer& p = parts[static_cast<size_t>(i)];
    rccl_check(ncclAllGather(p.buf, p.gathered, quads, ncclInt64, g.comm(i), g.stream(i)), "ncclAllGather");
    if (G > 1) {
      std::vector<const int64_t*> that;
      std::vector<uint32_t> counts;
      for (int j = 0; j < G; ++j) {
        if (j != i) {
          that.push_back(p.gathered + static_cast<size_t>(j) * quads);
          counts.push_back(entry_count);
        }
      }
      lib_check(hdk_hip_reduce_buffers(&plan, p.buf, entry_count, that.data(), counts.data(), G - 1, init_vals, p.dev_error, g.device(i),
                                       g.stream(i)),
                "hdk_hip_reduce_buffers");
    }
    hip_check(hipStreamSynchronize(g.stream(i)), "hipStreamSynchronize");
  });
}

// what device i brings to exchange_tuples
struct TupleExchangeRank {
  int8_t* const* scan_params;   // the 12 launch pointers over ITS fragments (GROUPBY_BUF unused)
  int8_t* const* owner_params;  // GROUPBY_BUF[0] = its owner table, INIT_AGG_VALS, ERROR_CODE
  int8_t* send;                 // size() segments, 256-byte aligned
  int8_t* recv;
  void* ws_scatter;
  void* ws_aggregate;
};

// One step of an open-addressing group-by over all devices: rows -> owner segments -> all-to-all -> owner tables.
// `shape` from hdk_hip_exchange_shape_for(plan, ko, size(), owner_entry_count, ...): the same for every device.
// Afterwards owner i's ERROR_CODE word is 0 or HDK_HIP_ERR_EXCHANGE_INCOMPLETE (skew / stale statistics: redo the step with
// partial tables).
inline void exchange_tuples(const HipDeviceGroup& g, const hdk_hip_plan& plan, const hdk_hip_kernel_options& ko,
                            const hdk_hip_exchange_shape& shape, const std::vector<TupleExchangeRank>& ranks) {
  const int G = g.size();
  if (static_cast<int>(ranks.size()) != G || static_cast<int>(shape.num_owners) != G) {
    throw std::runtime_error("exchange_tuples: one rank per device and a shape made for that many owners");
  }
  g.parallel([&](int i) {
    const TupleExchangeRank& r = ranks[static_cast<size_t>(i)];
    lib_check(hdk_hip_scatter_to_owners(&plan, r.scan_params, &ko, &shape, r.send, g.device(i), g.stream(i), r.ws_scatter,
                                        shape.scatter_workspace_bytes),
              "hdk_hip_scatter_to_owners");
    // all-to-all with equal splits: segment o of `send` -> device o, where it lands as segment i of `recv`
    rccl_check(ncclGroupStart(), "ncclGroupStart");
    for (int o = 0; o < G; ++o) {
      rccl_check(ncclSend(r.send + static_cast<size_t>(o) * shape.segment_bytes, shape.segment_bytes, ncclInt8, o, g.comm(i), g.stream(i)),
                 "ncclSend");
      rccl_check(ncclRecv(r.recv + static_cast<size_t>(o) * shape.segment_bytes, shape.segment_bytes, ncclInt8, o, g.comm(i), g.stream(i)),
                 "ncclRecv");
    }
    rccl_check(ncclGroupEnd(), "ncclGroupEnd");
    lib_check(hdk_hip_aggregate_from_ranks(&plan, r.owner_params, &ko, &shape, r.recv, g.device(i), g.stream(i), r.ws_aggregate,
                                           shape.aggregate_workspace_bytes),
              "hdk_hip_aggregate_from_ranks");
    hip_check(hipStreamSynchronize(g.stream(i)), "hipStreamSynchronize");
  });
}

}  // namespace hip_rt
