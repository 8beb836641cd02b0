// scan_bhm_host.h -- what the multi-argument on-chip group-by (scan_bhm.hip) and the translation units around it export to
// each other.
#pragma once
#include "host_common.h"
#include "device_common.h"

namespace hdk {

// scan_bhm.hip: kernel names of the launch when the multi-argument / multi-key on-chip group-by takes the plan, else nullptr;
// the launch itself (*launched = false: not this strategy's plan, or no scratch for its slabs)
const char* bhm_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko);
int32_t launch_bhm(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                   const hdk_hip_device_properties* props, hipStream_t s, bool* launched);

// scan_agg.hip: hdk_finalize over slabs of the generic partial-aggregate words ([slab][entry][word], agg_common.h) that another
// strategy wrote; skip_if != nullptr: the kernel does nothing when *skip_if != 0
int32_t launch_finalize_slabs(const hdk_hip_plan* d_plan, const int64_t* slabs, int64_t** groupby_buf, uint32_t num_slabs,
                              uint32_t entry_count, const uint32_t* skip_if, hipStream_t s);

// scan_baseline.hip: the global-atomics kernel (the reference's own scheme, any plan) ARMED behind another strategy: it runs
// only when *run_if != 0
int32_t launch_scan_global_armed(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp,
                                 const hdk_hip_kernel_options* ko, const hdk_hip_device_properties* props, hipStream_t s,
                                 const uint32_t* run_if);

}  // namespace hdk
