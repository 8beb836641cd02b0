// switches.h -- the HDK_HIP_* environment switches of libhdk_hip.so (tests and A/B measurements; production needs none).
// They are read ONCE per process -- the first time any of them is asked for, under std::call_once -- and never again on a
// launch path: no getenv per launch, nothing that races with a host calling setenv.  hdk_hip_reload_switches() (include/
// hdk_hip.h) re-reads them all; the Python harness calls it when it sees the environment change (hdk_amd/_lib.py).
#pragma once

namespace hdk {

#define HDK_SWITCH_LIST(X)          \
  X(BH_BLOCKS_PER_CU)               \
  X(BH_DIRECT_FOLD)                 \
  X(BH_FOLD_GROUPS)                 \
  X(BH_PARTITIONS_ALWAYS)           \
  X(BHM_BLOCKS_PER_CU)              \
  X(BHM_DYNAMIC)                    \
  X(BHM_FLAG_IS_ERROR)              \
  X(BHM_PART_GENERATION)            \
  X(BHM_PART_MIN_BINS)              \
  X(BHM_PART_REPLICAS)              \
  X(BHM_PART_SAMPLE_STRIDE)         \
  X(BHM_PART_TUPLES)                \
  X(BHM_WIDE_TUPLES)                \
  X(BUILD_PARTITION_MIN_ROWS)       \
  X(BUILD_TWO_LEVELS)               \
  X(COLS_BLOCKS_PER_CU)             \
  X(FAST_NO_XMODE)                  \
  X(KEYS_NO_WIDE_BLOCK)             \
  X(NO_BATCHED_MATCHING_SETS)       \
  X(NO_BH_DENSE)                    \
  X(NO_BH_DENSE_PARTITIONS)         \
  X(NO_BH_DIRECT)                   \
  X(NO_BH_LDS)                      \
  X(NO_BH_MOD_KEYS)                 \
  X(NO_BH_PACKED)                   \
  X(NO_BH_PARTITIONS)               \
  X(NO_BH_PLAIN)                    \
  X(NO_BHM)                         \
  X(NO_BHM_PARTITIONS)              \
  X(NO_COLS_KERNEL)                 \
  X(NO_PERFECT_PARTITIONS)          \
  X(NO_SLICED2)                     \
  X(PART_AOS)                       \
  X(PART_GENERAL)                   \
  X(PART_G_LOG2)                    \
  X(PART_TRACE)                     \
  X(PART_WIDE)                      \
  X(PERFECT_PARTITIONS_ALWAYS)      \
  X(PERFECT_SLICE_LOG2)             \
  X(POOL_KEEP_MB)                   \
  X(PP_BLOCKS_PER_CU)               \
  X(PROJECT_NO_FAST_JOIN)           \
  X(PROJECT_ONE_PASS)               \
  X(PROJECT_STATUS_SLACK)           \
  X(PROJECT_WRITER)                 \
  X(S2_NO_PACKED_PAIR)              \
  X(S2_NO_Y)                        \
  X(SCATTER_BLOCKS_PER_CU)          \
  X(SLICED2_ALWAYS)                 \
  X(SLICE_FINE_KEYS)                \
  X(SLICE_GENERAL)                  \
  X(SLICE_TWO_LEVELS)               \
  X(SLICE_WIDE)

enum SwitchId {
#define HDK_SW_ENUM(name) SW_##name,
  HDK_SWITCH_LIST(HDK_SW_ENUM)
#undef HDK_SW_ENUM
      SW_COUNT
};

// value of HDK_HIP_<name> as it was when the switches were last read, or nullptr when unset (runtime.hip)
const char* hdk_sw(SwitchId id);

}  // namespace hdk
