// launch_common.h -- launch-shape decisions shared by the scan kernels' translation unit.
#pragma once
#include "agg_common.h"
#include "host_common.h"

namespace hdk {

enum Strategy { STRAT_LDS = 0, STRAT_GLOBAL = 1, STRAT_PROJECT = 2 };

// LDS budget for the privatised table: 32 KiB per 256-thread block keeps 4-5 blocks (16-20 waves)
// resident per CU out of the 160 KiB LDS.
constexpr uint64_t kLdsWordBudget = 4096;
// Tables between 32 and 60 KiB still go to LDS, unreplicated (2 blocks per CU): LDS atomics on a few
// thousand entries stream at HBM speed, the global-atomics alternative is capped at ~2.4e10 rows/s.
constexpr uint64_t kLdsMaxTableWords = 7680;
// head of a launch's workspace: [plan copy | LaunchWatch (watch.h)], padded to 256 bytes
constexpr size_t kWatchOffset = (sizeof(hdk_hip_plan) + 15) & ~static_cast<size_t>(15);
constexpr size_t kPlanRegionBytes = (kWatchOffset + 16 + 255) & ~static_cast<size_t>(255);

struct LaunchShape {
  Strategy strategy;
  uint32_t grid;
  uint32_t block;
  uint32_t rep;
  uint32_t lds_bytes;
  uint32_t wpe;
  uint32_t entry_count;
  uint64_t slab_words;  // words per block slab (STRAT_LDS)
};

}  // namespace hdk
