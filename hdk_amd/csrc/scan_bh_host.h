// scan_bh_host.h -- what scan_bh.hip (open-addressing group-by in LDS, scan_bh.h) exports to the other translation units.
#pragma once
#include "host_common.h"
#include "device_common.h"

namespace hdk {

// kernel name of the launch when the LDS open-addressing strategy takes the plan, else nullptr
const char* bh_lds_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko);
// enqueues the scan (which folds its groups into GROUPBY_BUF[0] itself); *launched = false: not this strategy's plan
int32_t launch_bh_lds(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                      const hdk_hip_device_properties* props, hipStream_t s, bool* launched);

// the open-addressing interpreter ARMED behind another strategy's passes: runs only when *run_if == 1 (scan_agg.hip: the sliced
// join of a baseline-hash plan); folds into GROUPBY_BUF[0] itself
int32_t launch_bh_vec_armed(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                            const hdk_hip_device_properties* props, hipStream_t s, const uint32_t* run_if);
// slabs of a DENSE internal table ([slab][entry][word], agg_common.h's slab words: NULL-count words already hold the non-NULL
// counts) folded into the plan's open-addressing table: internal entry i is the group of key key_lo + i (or the NULL key)
struct BhDenseFold {
  const int64_t* slabs;
  uint32_t num_slabs;
  uint32_t entries;          // internal entries
  uint32_t out_entry_count;  // the plan's table
  int32_t wpe;
  int32_t wop[1 + 2 * HDK_HIP_MAX_TARGETS];
  uint32_t nword_mask;
  uint32_t null_entry;       // internal entry of the NULL key, or 0xFFFFFFFF
  int64_t key_lo;
  int64_t null_key;
};
int32_t launch_bh_fold_dense(const BhDenseFold& fold, const hdk_hip_plan* d_plan, const KernParams& kp, hipStream_t s);
// scan_agg.hip: a baseline-hash plan behind the sliced join (internal dense table + hdk_bh_fold_dense)
const char* baseline_sliced_join_names(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props);
int32_t launch_baseline_sliced_join(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                                    const hdk_hip_device_properties* props, hipStream_t s, bool* launched);

// scan_bh_packed.hip: the packed form (scan_bh_packed.h), tried first
const char* bh_packed_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko);
int32_t launch_bh_packed(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                         const hdk_hip_device_properties* props, hipStream_t s, bool* launched);

}  // namespace hdk
