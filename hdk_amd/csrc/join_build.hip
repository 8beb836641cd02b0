// join_build.hip -- perfect-hash join table build on the device.
//
// New kernels for the contract of the reference's *_on_device build functions
// (QE/JoinHashTable/Runtime/HashJoinRuntime.h:66-68,158-200; CUDA bodies
// QE/JoinHashTable/Runtime/HashJoinRuntimeGpu.cu:32-190 wrapping HashJoinRuntime.cpp:127-293,
// 589-853):
//   one-to-one : buff[(key-min)/bucket] = CAS(invalid -> row index); a taken slot reports -1 so the
//                caller falls back to one-to-many (Builders/PerfectHashTableBuilder.h:134-141)
//   one-to-many: [pos | count | row ids]: histogram -> inclusive scan -> scatter
// The reference walks the column with a per-thread strided iterator over JoinChunks
// (JoinColumnIterator.h:30-100); here each chunk is swept by the whole grid with coalesced loads,
// and the scan is a hand-written 3-kernel block scan instead of thrust::inclusive_scan.
#include <cstring>
#include "device_common.h"
#include "host_common.h"

namespace hdk {

constexpr int kJoinBlock = 256;

__global__ __launch_bounds__(kJoinBlock) void k_fill_i32(int32_t* __restrict__ buff, int64_t n, int32_t val) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kJoinBlock;
  int64_t i = static_cast<int64_t>(blockIdx.x) * kJoinBlock + threadIdx.x;
  // 16-B stores over the aligned body, scalar stores for head/tail
  const int64_t head = min<int64_t>(n, ((16 - (reinterpret_cast<uintptr_t>(buff) & 15)) & 15) / 4);
  if (i < head) {
    buff[i] = val;
  }
  int4* body = reinterpret_cast<int4*>(buff + head);
  const int64_t nvec = (n - head) / 4;
  const int4 v4 = make_int4(val, val, val, val);
  for (int64_t j = i; j < nvec; j += stride) {
    body[j] = v4;
  }
  const int64_t tail0 = head + nvec * 4;
  if (tail0 + i < n && i < 4) {
    buff[tail0 + i] = val;
  }
}

// JoinColumnIterator::getElementSwitch (JoinColumnIterator.h:36-62)
HDK_DEV int64_t join_elem(const int8_t* __restrict__ chunk, size_t i, int elem_sz, int column_type) {
  switch (column_type) {
    case HDK_JC_SMALL_DATE: {
      const int64_t v = decode_col(chunk, elem_sz, HDK_COL_INT, static_cast<int64_t>(i));
      const int64_t nullv = elem_sz == 4 ? static_cast<int64_t>(HDK_NULL_INT) : static_cast<int64_t>(HDK_NULL_SMALLINT);
      return v == nullv ? nullv : v * 86400;  // fixed_width_small_date_decode
    }
    case HDK_JC_UNSIGNED:
      return decode_col(chunk, elem_sz, HDK_COL_UNSIGNED, static_cast<int64_t>(i));
    case HDK_JC_DOUBLE:
      return static_cast<int64_t>(bits_to_double(decode_col(chunk, 8, HDK_COL_DOUBLE, static_cast<int64_t>(i))));
    default:
      return decode_col(chunk, elem_sz, HDK_COL_INT, static_cast<int64_t>(i));
  }
}

enum BuildMode { BUILD_ONE_TO_ONE = 0, BUILD_COUNT = 1, BUILD_FILL_IDS = 2 };

struct BuildArgs {
  int32_t* buff;           // one-to-one: the table; count: count_buff; fill: pos_buff (start of table)
  int64_t hash_entry_count;
  int32_t invalid_slot_val;
  int32_t for_semi_join;
  int32_t* dev_err;
  const hdk_hip_join_chunk* chunks;
  size_t num_chunks;
  hdk_hip_join_column_type_info ti;
  int64_t bucket;  // <= 1: plain
};

template <int MODE>
__global__ __launch_bounds__(kJoinBlock) void k_join_build(BuildArgs a) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kJoinBlock;
  const size_t start = static_cast<size_t>(blockIdx.x) * kJoinBlock + threadIdx.x;
  const int elem_sz = static_cast<int>(a.ti.elem_sz);
  size_t index_base = 0;
  for (size_t c = 0; c < a.num_chunks; ++c) {
    const hdk_hip_join_chunk ch = a.chunks[c];
    for (size_t i = start; i < ch.num_elems; i += stride) {
      int64_t elem = join_elem(ch.col_buff, i, elem_sz, a.ti.column_type);
      if (elem == a.ti.null_val) {
        if (a.ti.uses_bw_eq) {
          elem = a.ti.translated_null_val;
        } else {
          continue;
        }
      }
      int64_t slot = elem - a.ti.min_val;
      if (a.bucket > 1) {
        slot /= a.bucket;
      }
      if (static_cast<uint64_t>(slot) >= static_cast<uint64_t>(a.hash_entry_count)) {
        if (a.dev_err) {
          atomicMin(a.dev_err, -2);  // key outside [min,max]: the metadata the table was sized from is stale
        }
        continue;
      }
      const int32_t index = static_cast<int32_t>(index_base + i);
      if (MODE == BUILD_ONE_TO_ONE) {
        // fill_one_to_one_hashtable / fill_hashtable_for_semi_join (JoinHashImpl.h:55-80)
        const int32_t old = atomicCAS(a.buff + slot, a.invalid_slot_val, index);
        if (old != a.invalid_slot_val && !a.for_semi_join) {
          atomicMin(a.dev_err, -1);
        }
      } else if (MODE == BUILD_COUNT) {
        atomicAdd(a.buff + slot, 1);  // count_matches (HashJoinRuntime.cpp:589-636)
      } else {
        // fill_row_ids (HashJoinRuntime.cpp:770-822)
        int32_t* pos_buff = a.buff;
        int32_t* count_buff = a.buff + a.hash_entry_count;
        int32_t* id_buff = count_buff + a.hash_entry_count;
        const int32_t id_idx = atomicAdd(count_buff + slot, 1) + pos_buff[slot];
        id_buff[id_idx] = index;
      }
    }
    index_base += ch.num_elems;
  }
}

// ---- in-place inclusive scan of int32 (3 kernels) -------------------------------------------
constexpr int kScanItems = 16;
constexpr int kScanChunk = kJoinBlock * kScanItems;

HDK_DEV int32_t block_exclusive_scan(int32_t v, int32_t* total) {
  __shared__ int32_t wave_sums[kJoinBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  int32_t incl = v;
  for (int d = 1; d < kWave; d <<= 1) {
    const int32_t n = __shfl_up(incl, d, kWave);
    if (lane >= d) {
      incl += n;
    }
  }
  if (lane == kWave - 1) {
    wave_sums[wave] = incl;
  }
  __syncthreads();
  int32_t wave_off = 0;
  int32_t tot = 0;
  for (int w = 0; w < kJoinBlock / kWave; ++w) {
    if (w < wave) {
      wave_off += wave_sums[w];
    }
    tot += wave_sums[w];
  }
  __syncthreads();
  *total = tot;
  return wave_off + incl - v;
}

__global__ __launch_bounds__(kJoinBlock) void k_scan_block_sums(const int32_t* __restrict__ data, int64_t n,
                                                                int32_t* __restrict__ block_sums) {
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk + static_cast<int64_t>(threadIdx.x) * kScanItems;
  int32_t s = 0;
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    if (i < n) {
      s += data[i];
    }
  }
  int32_t total;
  block_exclusive_scan(s, &total);
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x] = total;
  }
}

__global__ __launch_bounds__(kJoinBlock) void k_scan_sums_inplace(int32_t* __restrict__ sums, int64_t nb) {
  // single block: exclusive scan of the block sums, chunk by chunk with a running carry
  int32_t carry = 0;
  for (int64_t base = 0; base < nb; base += kJoinBlock) {
    const int64_t i = base + threadIdx.x;
    const int32_t v = i < nb ? sums[i] : 0;
    int32_t total;
    const int32_t ex = block_exclusive_scan(v, &total);
    if (i < nb) {
      sums[i] = carry + ex;
    }
    carry += total;
  }
}

__global__ __launch_bounds__(kJoinBlock) void k_scan_apply(int32_t* __restrict__ data, int64_t n,
                                                           const int32_t* __restrict__ block_offs) {
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk + static_cast<int64_t>(threadIdx.x) * kScanItems;
  int32_t vals[kScanItems];
  int32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    vals[k] = i < n ? data[i] : 0;
    s += vals[k];
  }
  int32_t total;
  int32_t run = block_exclusive_scan(s, &total) + block_offs[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    run += vals[k];
    if (i < n) {
      data[i] = run;  // inclusive
    }
  }
}

// set_valid_pos_flag / set_valid_pos (HashJoinRuntimeGpu.cu:110-133) fused: runs before the scan for
// the flag and after it for the value, so it takes both forms through `after_scan`.
__global__ __launch_bounds__(kJoinBlock) void k_set_valid_pos(int32_t* __restrict__ pos_buff,
                                                              const int32_t* __restrict__ count_buff,
                                                              int64_t entry_count, int after_scan) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kJoinBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kJoinBlock + threadIdx.x; i < entry_count; i += stride) {
    if (!after_scan) {
      if (count_buff[i]) {
        pos_buff[i] = 0;  // VALID_POS_FLAG
      }
    } else if (pos_buff[i] == 0) {
      pos_buff[i] = i ? count_buff[i - 1] : 0;
    }
  }
}

static unsigned grid_for(size_t n, int32_t device_id) {
  const hdk_hip_device_properties* props = device_props(device_id);
  size_t blocks = (n + kJoinBlock - 1) / kJoinBlock;
  const size_t cap = static_cast<size_t>(props->num_cu) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  return static_cast<unsigned>(blocks);
}

static int32_t inclusive_scan_inplace(int32_t* data, int64_t n, hipStream_t s) {
  if (n <= 0) {
    return HDK_HIP_OK;
  }
  const int64_t nb = (n + kScanChunk - 1) / kScanChunk;
  AsyncScratch sums_mem(s);
  HDK_HIP_CHECK(hipMallocAsync(&sums_mem.p, static_cast<size_t>(nb) * sizeof(int32_t), s));
  int32_t* sums = static_cast<int32_t*>(sums_mem.p);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(static_cast<unsigned>(nb)), dim3(kJoinBlock), 0, s, data, n, sums);
  hipLaunchKernelGGL(k_scan_sums_inplace, dim3(1), dim3(kJoinBlock), 0, s, sums, nb);
  hipLaunchKernelGGL(k_scan_apply, dim3(static_cast<unsigned>(nb)), dim3(kJoinBlock), 0, s, data, n, sums);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

static int32_t check_join_args(const hdk_hip_join_column& jc, const hdk_hip_join_column_type_info& ti) {
  HDK_REQUIRE(jc.num_chunks == 0 || jc.col_chunks_buff, "JoinColumn.col_chunks_buff is NULL");
  HDK_REQUIRE(ti.elem_sz == 1 || ti.elem_sz == 2 || ti.elem_sz == 4 || ti.elem_sz == 8,
              "join column element size must be 1/2/4/8");
  return HDK_HIP_OK;
}

static int32_t one_to_one(int32_t* buff, int32_t invalid_slot_val, int32_t for_semi_join, int32_t* dev_err_buff,
                          const hdk_hip_join_column& jc, const hdk_hip_join_column_type_info& ti,
                          int64_t bucket, int32_t device_id, void* stream) {
  HDK_REQUIRE(buff && dev_err_buff, "NULL buffer");
  int32_t st = check_join_args(jc, ti);
  if (st) return st;
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  if (jc.num_elems == 0) {
    return HDK_HIP_OK;
  }
  BuildArgs a;
  a.buff = buff;
  // HashEntryInfo{max - min + 1 (+ 1 for the NULLs of a kBwEq join), bucket}.getNormalizedHashEntryCount()
  // (PerfectJoinHashTable.cpp:45-85, HashJoinRuntime.h:46-55): the slots the caller allocated and initialised
  const int64_t range = ti.max_val - ti.min_val + 1 + (ti.uses_bw_eq ? 1 : 0);
  a.hash_entry_count = bucket > 1 ? (range + bucket - 1) / bucket : range;
  a.invalid_slot_val = invalid_slot_val;
  a.for_semi_join = for_semi_join;
  a.dev_err = dev_err_buff;
  a.chunks = reinterpret_cast<const hdk_hip_join_chunk*>(jc.col_chunks_buff);
  a.num_chunks = jc.num_chunks;
  a.ti = ti;
  a.bucket = bucket;
  hipLaunchKernelGGL(k_join_build<BUILD_ONE_TO_ONE>, dim3(grid_for(jc.num_elems, device_id)), dim3(kJoinBlock),
                     0, s, a);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

static int32_t one_to_many(int32_t* buff, int64_t hash_entry_count, int32_t invalid_slot_val,
                           const hdk_hip_join_column& jc, const hdk_hip_join_column_type_info& ti,
                           int64_t bucket, int32_t device_id, void* stream) {
  HDK_REQUIRE(buff, "NULL buffer");
  HDK_REQUIRE(hash_entry_count > 0, "hash_entry_count must be positive");
  int32_t st = check_join_args(jc, ti);
  if (st) return st;
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  int32_t* pos_buff = buff;
  int32_t* count_buff = buff + hash_entry_count;
  HDK_HIP_CHECK(hipMemsetAsync(count_buff, 0, static_cast<size_t>(hash_entry_count) * sizeof(int32_t), s));
  BuildArgs a;
  a.hash_entry_count = hash_entry_count;
  a.invalid_slot_val = invalid_slot_val;
  a.for_semi_join = 0;
  a.dev_err = nullptr;
  a.chunks = reinterpret_cast<const hdk_hip_join_chunk*>(jc.col_chunks_buff);
  a.num_chunks = jc.num_chunks;
  a.ti = ti;
  a.bucket = bucket;
  const unsigned g = grid_for(jc.num_elems, device_id);
  const unsigned ge = grid_for(static_cast<size_t>(hash_entry_count), device_id);
  if (jc.num_elems) {
    a.buff = count_buff;
    hipLaunchKernelGGL(k_join_build<BUILD_COUNT>, dim3(g), dim3(kJoinBlock), 0, s, a);
  }
  hipLaunchKernelGGL(k_set_valid_pos, dim3(ge), dim3(kJoinBlock), 0, s, pos_buff, count_buff, hash_entry_count, 0);
  st = inclusive_scan_inplace(count_buff, hash_entry_count, s);
  if (st) return st;
  hipLaunchKernelGGL(k_set_valid_pos, dim3(ge), dim3(kJoinBlock), 0, s, pos_buff, count_buff, hash_entry_count, 1);
  HDK_HIP_CHECK(hipMemsetAsync(count_buff, 0, static_cast<size_t>(hash_entry_count) * sizeof(int32_t), s));
  if (jc.num_elems) {
    a.buff = pos_buff;
    hipLaunchKernelGGL(k_join_build<BUILD_FILL_IDS>, dim3(g), dim3(kJoinBlock), 0, s, a);
  }
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// ---- keyed ("baseline") tables: composite / wide keys, open addressing with MurmurHash1 ---------------
// write_baseline_hash_slot + get_matching_baseline_hash_slot_at (HashJoinRuntime.cpp:357-452),
// count_matches_baseline / fill_row_ids_baseline (:723-768, :889-950), key handler
// HashJoinKeyHandlers.h:36-100 (a row with a NULL component is skipped).
enum KeyedMode { KEYED_ONE_TO_ONE = 0, KEYED_DICT = 1, KEYED_COUNT = 2, KEYED_FILL_IDS = 3 };

struct KeyedArgs {
  int8_t* hash_buff;        // keys (+ payload for one-to-one)
  int32_t* otm;             // one-to-many: [pos | count | ids] after the dictionary
  int64_t entry_count;
  int32_t invalid_slot_val;
  int32_t for_semi_join;
  int32_t kc;
  int32_t* dev_err;
  const hdk_hip_join_chunk* chunks[HDK_HIP_MAX_JOIN_KEYS];
  size_t num_chunks;
  hdk_hip_join_column_type_info ti[HDK_HIP_MAX_JOIN_KEYS];
};

template <typename T>
HDK_DEV T keyed_invalid() {
  return sizeof(T) == 8 ? static_cast<T>(HDK_EMPTY_KEY_64) : static_cast<T>(HDK_EMPTY_KEY_32);
}

template <typename T>
HDK_DEV uint32_t keyed_hash(const int64_t* key, int kc) {
  uint32_t words[2 * HDK_HIP_MAX_JOIN_KEYS];
  int nw = 0;
#pragma unroll
  for (int i = 0; i < HDK_HIP_MAX_JOIN_KEYS; ++i) {
    if (i < kc) {
      if constexpr (sizeof(T) == 8) {
        words[nw++] = static_cast<uint32_t>(static_cast<uint64_t>(key[i]));
        words[nw++] = static_cast<uint32_t>(static_cast<uint64_t>(key[i]) >> 32);
      } else {
        words[nw++] = static_cast<uint32_t>(key[i]);
      }
    }
  }
  return murmur_hash1_words(words, nw);
}

// Find the slot of `key`, claiming an empty one when CLAIM.  Wave-safe publication: a reader that
// finds the first component written but a later one still empty re-examines the slot on the next
// trip of the loop instead of spinning (the writer may be a lane of the same wave).
template <typename T, bool CLAIM>
HDK_DEV int64_t keyed_slot(T* dict, uint32_t entries, int comps, int kc, const int64_t* key) {
  const T invalid = keyed_invalid<T>();
  uint32_t probe = keyed_hash<T>(key, kc) % entries;
  uint32_t steps = 0;
  int64_t result = -2;
  while (result == -2) {
    T* e = dict + static_cast<size_t>(probe) * comps;
    T first;
    bool won = false;
    if (CLAIM) {
      if constexpr (sizeof(T) == 8) {
        first = static_cast<T>(atomicCAS(reinterpret_cast<unsigned long long*>(e), static_cast<unsigned long long>(invalid),
                                         static_cast<unsigned long long>(static_cast<T>(key[0]))));
      } else {
        first = static_cast<T>(atomicCAS(reinterpret_cast<unsigned int*>(e), static_cast<unsigned int>(invalid),
                                         static_cast<unsigned int>(static_cast<T>(key[0]))));
      }
      won = first == invalid;
      if (won) {
        for (int i = 1; i < kc; ++i) {
          __hip_atomic_store(e + i, static_cast<T>(key[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        result = probe;
      }
    } else {
      first = e[0];
    }
    bool advance = false;
    if (!won) {
      if (first == static_cast<T>(key[0]) && first != invalid) {
        bool eq = true;
        bool pending = false;
        for (int i = 1; i < kc; ++i) {
          const T v = __hip_atomic_load(e + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v == invalid && static_cast<T>(key[i]) != invalid) {
            pending = true;
          } else if (v != static_cast<T>(key[i])) {
            eq = false;
          }
        }
        if (!eq) {
          advance = true;
        } else if (!pending || !CLAIM) {
          result = pending ? -1 : static_cast<int64_t>(probe);
        }
      } else if (!CLAIM && first == invalid) {
        result = -1;
      } else {
        advance = true;
      }
    }
    if (advance) {
      probe = probe + 1 == entries ? 0 : probe + 1;
      if (++steps >= entries) {
        result = -1;
      }
    }
  }
  return result;
}

template <typename T, int MODE>
__global__ __launch_bounds__(kJoinBlock) void k_keyed_build(KeyedArgs a) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kJoinBlock;
  const size_t start = static_cast<size_t>(blockIdx.x) * kJoinBlock + threadIdx.x;
  const int kc = a.kc;
  const int comps = kc + (MODE == KEYED_ONE_TO_ONE ? 1 : 0);
  const uint32_t entries = static_cast<uint32_t>(a.entry_count);
  T* dict = reinterpret_cast<T*>(a.hash_buff);
  size_t index_base = 0;
  for (size_t c = 0; c < a.num_chunks; ++c) {
    const size_t n = a.chunks[0][c].num_elems;
    for (size_t i = start; i < n; i += stride) {
      int64_t key[HDK_HIP_MAX_JOIN_KEYS];
      bool skip = false;
#pragma unroll
      for (int k = 0; k < HDK_HIP_MAX_JOIN_KEYS; ++k) {
        key[k] = 0;
        if (k < kc) {
          const int64_t elem = join_elem(a.chunks[k][c].col_buff, i, static_cast<int>(a.ti[k].elem_sz), a.ti[k].column_type);
          if (elem == a.ti[k].null_val && !a.ti[k].uses_bw_eq) {
            skip = true;
          }
          key[k] = elem;
        }
      }
      if (skip) {
        continue;
      }
      const int32_t index = static_cast<int32_t>(index_base + i);
      const int64_t slot = keyed_slot<T, (MODE == KEYED_ONE_TO_ONE || MODE == KEYED_DICT)>(dict, entries, comps, kc, key);
      if (slot < 0) {
        if (a.dev_err) {
          atomicMin(a.dev_err, -2);  // table full
        }
        continue;
      }
      if (MODE == KEYED_ONE_TO_ONE) {
        T* val = dict + static_cast<size_t>(slot) * comps + kc;
        T old;
        if constexpr (sizeof(T) == 8) {
          old = static_cast<T>(atomicCAS(reinterpret_cast<unsigned long long*>(val),
                                         static_cast<unsigned long long>(static_cast<T>(a.invalid_slot_val)),
                                         static_cast<unsigned long long>(static_cast<T>(index))));
        } else {
          old = static_cast<T>(atomicCAS(reinterpret_cast<unsigned int*>(val), static_cast<unsigned int>(a.invalid_slot_val),
                                         static_cast<unsigned int>(index)));
        }
        if (old != static_cast<T>(a.invalid_slot_val) && !a.for_semi_join) {
          atomicMin(a.dev_err, -1);  // duplicate key: the caller builds a one-to-many table instead
        }
      } else if (MODE == KEYED_COUNT) {
        atomicAdd(a.otm + entries + slot, 1);
      } else if (MODE == KEYED_FILL_IDS) {
        const int32_t at = a.otm[slot] + atomicAdd(a.otm + entries + slot, 1);
        a.otm[2 * static_cast<size_t>(entries) + at] = index;
      }
    }
    index_base += n;
  }
}

template <typename T>
__global__ __launch_bounds__(kJoinBlock) void k_keyed_init(T* buff, int64_t entry_count, int comps, int kc,
                                                           int32_t invalid_slot_val) {
  const int64_t n = entry_count * comps;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kJoinBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kJoinBlock + threadIdx.x; i < n; i += stride) {
    buff[i] = (i % comps) < kc ? keyed_invalid<T>() : static_cast<T>(invalid_slot_val);
  }
}

static int32_t keyed_args(KeyedArgs* a, size_t key_component_count, int32_t key_component_width,
                          const hdk_hip_join_column* cols, const hdk_hip_join_column_type_info* ti) {
  HDK_REQUIRE(key_component_count >= 1 && key_component_count <= HDK_HIP_MAX_JOIN_KEYS,
              "key_component_count must be in [1, %d]", HDK_HIP_MAX_JOIN_KEYS);
  HDK_REQUIRE(key_component_width == 4 || key_component_width == 8, "key_component_width must be 4 or 8");
  HDK_REQUIRE(cols && ti, "NULL join columns");
  memset(a, 0, sizeof(*a));
  a->kc = static_cast<int32_t>(key_component_count);
  for (size_t k = 0; k < key_component_count; ++k) {
    const int32_t st = check_join_args(cols[k], ti[k]);
    if (st) return st;
    HDK_REQUIRE(cols[k].num_chunks == cols[0].num_chunks && cols[k].num_elems == cols[0].num_elems,
                "key columns of one table must be fragmented alike");
    a->chunks[k] = reinterpret_cast<const hdk_hip_join_chunk*>(cols[k].col_chunks_buff);
    a->ti[k] = ti[k];
  }
  a->num_chunks = cols[0].num_chunks;
  return HDK_HIP_OK;
}

// ---- fused one-to-one table: [row id | payload words] per slot (HDK_JOIN_ONE_TO_ONE_FUSED) ------------
constexpr int kMaxFusedCols = 7;
struct FusedArgs {
  const int32_t* table;
  int64_t entry_count;
  const int8_t* cols[kMaxFusedCols];
  int32_t widths[kMaxFusedCols];
  int32_t kinds[kMaxFusedCols];
  int32_t ncols;
  int64_t* out;
};

__global__ __launch_bounds__(kJoinBlock) void k_build_fused(FusedArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kJoinBlock;
  const int64_t es = 1 + a.ncols;
  for (int64_t slot = static_cast<int64_t>(blockIdx.x) * kJoinBlock + threadIdx.x; slot < a.entry_count; slot += stride) {
    const int32_t rid = a.table[slot];
    int64_t* o = a.out + slot * es;
    o[0] = rid;
    for (int c = 0; c < a.ncols; ++c) {
      o[1 + c] = rid >= 0 ? decode_col(a.cols[c], a.widths[c], a.kinds[c], rid) : 0;
    }
  }
}

}  // namespace hdk

using namespace hdk;

extern "C" int32_t hdk_hip_build_fused_join_table(const int32_t* table, int64_t entry_count,
                                                  const int8_t* const* inner_cols, const int32_t* widths,
                                                  const int32_t* kinds, int32_t ncols, int64_t* out,
                                                  int32_t device_id, void* stream) {
  HDK_REQUIRE(table && out, "NULL buffer");
  HDK_REQUIRE(ncols >= 0 && ncols <= kMaxFusedCols, "0..%d payload columns", kMaxFusedCols);
  HDK_REQUIRE(ncols == 0 || (inner_cols && widths && kinds), "NULL column description");
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  if (entry_count <= 0) {
    return HDK_HIP_OK;
  }
  FusedArgs a;
  a.table = table;
  a.entry_count = entry_count;
  a.ncols = ncols;
  a.out = out;
  for (int c = 0; c < kMaxFusedCols; ++c) {
    a.cols[c] = c < ncols ? inner_cols[c] : nullptr;
    a.widths[c] = c < ncols ? widths[c] : 0;
    a.kinds[c] = c < ncols ? kinds[c] : 0;
  }
  hipLaunchKernelGGL(k_build_fused, dim3(grid_for(static_cast<size_t>(entry_count), device_id)), dim3(kJoinBlock), 0, s, a);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_init_hash_join_buff(int32_t* buff, int64_t entry_count, int32_t invalid_slot_val,
                                               int32_t device_id, void* stream) {
  HDK_REQUIRE(buff || entry_count == 0, "NULL buffer");
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  if (entry_count <= 0) {
    return HDK_HIP_OK;
  }
  hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(static_cast<size_t>(entry_count) / 4 + 1, device_id)),
                     dim3(kJoinBlock), 0, s, buff, entry_count, invalid_slot_val);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_fill_hash_join_buff(int32_t* buff, int32_t invalid_slot_val, int32_t for_semi_join,
                                               int32_t* dev_err_buff, hdk_hip_join_column join_column,
                                               hdk_hip_join_column_type_info type_info, int32_t device_id,
                                               void* stream) {
  return one_to_one(buff, invalid_slot_val, for_semi_join, dev_err_buff, join_column, type_info, 1, device_id,
                    stream);
}

extern "C" int32_t hdk_hip_fill_hash_join_buff_bucketized(int32_t* buff, int32_t invalid_slot_val,
                                                          int32_t for_semi_join, int32_t* dev_err_buff,
                                                          hdk_hip_join_column join_column,
                                                          hdk_hip_join_column_type_info type_info,
                                                          int64_t bucket_normalization, int32_t device_id,
                                                          void* stream) {
  HDK_REQUIRE(bucket_normalization > 0, "bucket_normalization must be positive");
  return one_to_one(buff, invalid_slot_val, for_semi_join, dev_err_buff, join_column, type_info,
                    bucket_normalization, device_id, stream);
}

extern "C" int32_t hdk_hip_fill_one_to_many_hash_table(int32_t* buff, hdk_hip_hash_entry_info hash_entry_info,
                                                       int32_t invalid_slot_val,
                                                       hdk_hip_join_column join_column,
                                                       hdk_hip_join_column_type_info type_info,
                                                       int32_t device_id, void* stream) {
  return one_to_many(buff, static_cast<int64_t>(hash_entry_info.hash_entry_count), invalid_slot_val,
                     join_column, type_info, 1, device_id, stream);
}

extern "C" int32_t hdk_hip_fill_one_to_many_hash_table_bucketized(int32_t* buff,
                                                                  hdk_hip_hash_entry_info hash_entry_info,
                                                                  int32_t invalid_slot_val,
                                                                  hdk_hip_join_column join_column,
                                                                  hdk_hip_join_column_type_info type_info,
                                                                  int32_t device_id, void* stream) {
  HDK_REQUIRE(hash_entry_info.bucket_normalization > 0, "bucket_normalization must be positive");
  // HashEntryInfo::getNormalizedHashEntryCount (HashJoinRuntime.h:47-55)
  const size_t b = static_cast<size_t>(hash_entry_info.bucket_normalization);
  const size_t n = hash_entry_info.hash_entry_count / b + (hash_entry_info.hash_entry_count % b ? 1 : 0);
  return one_to_many(buff, static_cast<int64_t>(n), invalid_slot_val, join_column, type_info,
                     hash_entry_info.bucket_normalization, device_id, stream);
}

extern "C" int32_t hdk_hip_init_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count,
                                                        size_t key_component_count, int32_t key_component_width,
                                                        int32_t with_val_slot, int32_t invalid_slot_val,
                                                        int32_t device_id, void* stream) {
  HDK_REQUIRE(hash_buff && entry_count > 0, "bad buffer / entry_count");
  HDK_REQUIRE(key_component_count >= 1 && key_component_count <= HDK_HIP_MAX_JOIN_KEYS, "bad key_component_count");
  HDK_REQUIRE(key_component_width == 4 || key_component_width == 8, "key_component_width must be 4 or 8");
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  const int kc = static_cast<int>(key_component_count);
  const int comps = kc + (with_val_slot ? 1 : 0);
  const unsigned g = grid_for(static_cast<size_t>(entry_count) * comps, device_id);
  if (key_component_width == 4) {
    hipLaunchKernelGGL(k_keyed_init<int32_t>, dim3(g), dim3(kJoinBlock), 0, s, reinterpret_cast<int32_t*>(hash_buff),
                       entry_count, comps, kc, invalid_slot_val);
  } else {
    hipLaunchKernelGGL(k_keyed_init<int64_t>, dim3(g), dim3(kJoinBlock), 0, s, reinterpret_cast<int64_t*>(hash_buff),
                       entry_count, comps, kc, invalid_slot_val);
  }
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_fill_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                                        int32_t for_semi_join, size_t key_component_count,
                                                        int32_t key_component_width, int32_t with_val_slot,
                                                        int32_t* dev_err_buff, const hdk_hip_join_column* join_column_per_key,
                                                        const hdk_hip_join_column_type_info* type_info_per_key,
                                                        int32_t device_id, void* stream) {
  HDK_REQUIRE(hash_buff && dev_err_buff && entry_count > 0 && entry_count < (int64_t(1) << 31), "bad arguments");
  KeyedArgs a;
  int32_t st = keyed_args(&a, key_component_count, key_component_width, join_column_per_key, type_info_per_key);
  if (st) return st;
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  if (join_column_per_key[0].num_elems == 0) return HDK_HIP_OK;
  a.hash_buff = hash_buff;
  a.entry_count = entry_count;
  a.invalid_slot_val = invalid_slot_val;
  a.for_semi_join = for_semi_join;
  a.dev_err = dev_err_buff;
  const unsigned g = grid_for(join_column_per_key[0].num_elems, device_id);
  if (key_component_width == 4) {
    if (with_val_slot) {
      hipLaunchKernelGGL((k_keyed_build<int32_t, KEYED_ONE_TO_ONE>), dim3(g), dim3(kJoinBlock), 0, s, a);
    } else {
      hipLaunchKernelGGL((k_keyed_build<int32_t, KEYED_DICT>), dim3(g), dim3(kJoinBlock), 0, s, a);
    }
  } else if (with_val_slot) {
    hipLaunchKernelGGL((k_keyed_build<int64_t, KEYED_ONE_TO_ONE>), dim3(g), dim3(kJoinBlock), 0, s, a);
  } else {
    hipLaunchKernelGGL((k_keyed_build<int64_t, KEYED_DICT>), dim3(g), dim3(kJoinBlock), 0, s, a);
  }
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_fill_one_to_many_baseline_hash_table(int32_t* buff, const int8_t* composite_key_dict,
                                                                int64_t hash_entry_count, int32_t invalid_slot_val,
                                                                size_t key_component_count, int32_t key_component_width,
                                                                const hdk_hip_join_column* join_column_per_key,
                                                                const hdk_hip_join_column_type_info* type_info_per_key,
                                                                int32_t device_id, void* stream) {
  HDK_REQUIRE(buff && composite_key_dict && hash_entry_count > 0 && hash_entry_count < (int64_t(1) << 31),
              "bad arguments");
  KeyedArgs a;
  int32_t st = keyed_args(&a, key_component_count, key_component_width, join_column_per_key, type_info_per_key);
  if (st) return st;
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  a.hash_buff = const_cast<int8_t*>(composite_key_dict);
  a.otm = buff;
  a.entry_count = hash_entry_count;
  a.invalid_slot_val = invalid_slot_val;
  int32_t* pos_buff = buff;
  int32_t* count_buff = buff + hash_entry_count;
  const size_t n = join_column_per_key[0].num_elems;
  const unsigned g = grid_for(n, device_id);
  const unsigned ge = grid_for(static_cast<size_t>(hash_entry_count), device_id);
  hipLaunchKernelGGL(k_fill_i32, dim3(ge), dim3(kJoinBlock), 0, s, pos_buff, hash_entry_count, invalid_slot_val);
  HDK_HIP_CHECK(hipMemsetAsync(count_buff, 0, static_cast<size_t>(hash_entry_count) * sizeof(int32_t), s));
  if (n) {
    if (key_component_width == 4) {
      hipLaunchKernelGGL((k_keyed_build<int32_t, KEYED_COUNT>), dim3(g), dim3(kJoinBlock), 0, s, a);
    } else {
      hipLaunchKernelGGL((k_keyed_build<int64_t, KEYED_COUNT>), dim3(g), dim3(kJoinBlock), 0, s, a);
    }
  }
  hipLaunchKernelGGL(k_set_valid_pos, dim3(ge), dim3(kJoinBlock), 0, s, pos_buff, count_buff, hash_entry_count, 0);
  st = inclusive_scan_inplace(count_buff, hash_entry_count, s);
  if (st) return st;
  hipLaunchKernelGGL(k_set_valid_pos, dim3(ge), dim3(kJoinBlock), 0, s, pos_buff, count_buff, hash_entry_count, 1);
  HDK_HIP_CHECK(hipMemsetAsync(count_buff, 0, static_cast<size_t>(hash_entry_count) * sizeof(int32_t), s));
  if (n) {
    if (key_component_width == 4) {
      hipLaunchKernelGGL((k_keyed_build<int32_t, KEYED_FILL_IDS>), dim3(g), dim3(kJoinBlock), 0, s, a);
    } else {
      hipLaunchKernelGGL((k_keyed_build<int64_t, KEYED_FILL_IDS>), dim3(g), dim3(kJoinBlock), 0, s, a);
    }
  }
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}
