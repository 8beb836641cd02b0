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

// scan_bh_packed.hip: the packed form (scan_bh_packed.h), tried first
const char* bh_packed_kernel_name(const hdk_hip_plan* p, const hdk_hip_kernel_options* ko);
int32_t launch_bh_packed(const hdk_hip_plan* plan, const hdk_hip_plan* d_plan, const KernParams& kp, const hdk_hip_kernel_options* ko,
                         const hdk_hip_device_properties* props, hipStream_t s, bool* launched);

}  // namespace hdk
