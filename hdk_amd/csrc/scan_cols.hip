// scan_cols.hip -- the column-by-column streaming kernel for non-grouped aggregates over several columns
// (scan_agg_cols.h: hdk_scan_agg_cols) and its launcher; scan_agg.hip holds the matcher (match_cols).
#include "host_match.h"
#include "scan_agg_cols.h"

namespace hdk {

int32_t launch_cols(const ColsArgs& ca, const LaunchShape& shape, hipStream_t s) {
  hipLaunchKernelGGL(hdk_scan_agg_cols<kColsU>, dim3(shape.grid), dim3(kColsBlock), 0, s, ca);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

}  // namespace hdk
