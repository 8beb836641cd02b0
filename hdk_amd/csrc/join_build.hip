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
#include "switches.h"

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
  const uint32_t* run_if;  // nullptr: always; else only when *run_if != 0 (armed behind the partitioned build, join_build_part.h)
};

template <int MODE>
__global__ __launch_bounds__(kJoinBlock) void k_join_build(BuildArgs a) {
  if (a.run_if && *a.run_if == 0) {
    return;
  }
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

static int64_t pb_min_rows();
static int32_t one_to_one_partitioned(int32_t* buff, int32_t invalid_slot_val, int32_t* dev_err_buff, const hdk_hip_join_column& jc,
                                      const hdk_hip_join_column_type_info& ti, int64_t bucket, const int8_t* const* pcols,
                                      const int32_t* widths, const int32_t* kinds, int np, int64_t* fused, void* scratch,
                                      size_t scratch_bytes, int32_t device_id, hipStream_t s, bool* done);

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
  if (!for_semi_join && pb_min_rows() > 0 && static_cast<int64_t>(jc.num_elems) >= pb_min_rows()) {
    bool done = false;
    st = one_to_one_partitioned(buff, invalid_slot_val, dev_err_buff, jc, ti, bucket, nullptr, nullptr, nullptr, 0, nullptr,
                                nullptr, 0, device_id, s, &done);
    if (st || done) return st;
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
  a.run_if = nullptr;
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
  a.run_if = nullptr;
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
  const uint32_t* run_if;  // as BuildArgs::run_if
};

__global__ __launch_bounds__(kJoinBlock) void k_build_fused(FusedArgs a) {
  if (a.run_if && *a.run_if == 0) {
    return;
  }
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

#include "join_build_part.h"

namespace hdk {

// ---- the partitioned one-to-one build (join_build_part.h): geometry, scratch layout, launches ---------------------------
struct PbLayout {
  size_t off_fill1, off_fill2, off_t1, off_t2, cursor_bytes, total;
};

static size_t pb_up(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }

// false: this table is not one for the partitioned build (no rows, more row ids or slots than 32 bits address)
static bool pb_geometry(PbArgs* a, PbLayout* l, int64_t rows, int64_t entries, int np) {
  if (rows <= 0 || entries <= 0 || rows > INT32_MAX || entries > INT32_MAX || np < 0 || np > kPbMaxPayload) return false;
  // A SPARSE table (a dimension of 3 M rows over a key range of 2^31) is not for this build: the build pass rewrites every
  // slot and level 2 keeps 8 bytes of scratch per slot -- gigabytes moved and allocated where the atomic build does one CAS
  // per row.  Dense enough = at most 16 slots per row.
  if (entries / 16 > rows) return false;
  const int tw = 1 + np;
  a->np = np;
  a->hash_entry_count = entries;
  a->slice_log2 = np ? kPbSliceLog2Fused : kPbSliceLog2;
  a->nslices = static_cast<uint32_t>((entries + (1ll << a->slice_log2) - 1) >> a->slice_log2);
  uint32_t fpc_log2 = 0;
  while (((a->nslices + (1u << fpc_log2) - 1) >> fpc_log2) > static_cast<uint32_t>(kPbMaxBins)) ++fpc_log2;
  if (const char* e = hdk_sw(SW_BUILD_TWO_LEVELS)) {  // tests: two levels on small tables
    const uint32_t want = static_cast<uint32_t>(atoi(e));
    if (want > fpc_log2 && want <= 8) fpc_log2 = want;
  }
  if (fpc_log2 > 8) return false;
  a->fpc_log2 = fpc_log2;
  a->two_level = fpc_log2 ? 1u : 0u;
  a->nb1 = (a->nslices + (1u << fpc_log2) - 1) >> fpc_log2;
  const uint64_t nsub = static_cast<uint64_t>(a->nb1) * kPbXcds;
  a->cap1 = ((static_cast<uint64_t>(rows) / nsub) * 5 / 4 + 4096 + 15) & ~15ull;
  a->cap2 = 1ull << a->slice_log2;
  if (a->cap1 > 0xFFFFFFF0ull) return false;
  l->off_fill1 = 256;  // [0]: the flag word
  l->off_fill2 = l->off_fill1 + pb_up(nsub * kPbCursorStride * 4);
  l->cursor_bytes = l->off_fill2 + (a->two_level ? pb_up(static_cast<size_t>(a->nslices) * kPbCursor2Stride * 4) : 0);
  l->off_t1 = l->cursor_bytes;
  l->off_t2 = l->off_t1 + pb_up(nsub * a->cap1 * tw * 8);
  l->total = l->off_t2 + (a->two_level ? pb_up(static_cast<size_t>(a->nslices) * a->cap2 * tw * 8) : 0);
  if (l->total > (static_cast<size_t>(16) << 30)) return false;  // (never starve the stream pool: the atomic build needs no scratch)
  return true;
}

template <int TW>
static int32_t pb_launch(const PbArgs& a, const hdk_hip_device_properties* props, hipStream_t s) {
  constexpr int VR = TW == 1 ? 8 : (TW == 2 ? 4 : 2);
  const size_t lds_sc = PbStage<TW, VR>::lds_bytes();
  const unsigned cu = static_cast<unsigned>(props->num_cu);
  hipLaunchKernelGGL((k_pb_scatter<TW, VR>), dim3(2 * cu), dim3(kPbBlock), lds_sc, s, a);
  if (a.two_level) {
    // level 2: `members2` blocks per level-1 bin, a bin's blocks congruent modulo 8 (one XCD); about three resident blocks per CU
    PbArgs a2 = a;
    const unsigned bins8 = (a.nb1 + kPbXcds - 1) / kPbXcds;
    unsigned m2 = (3 * cu) / (bins8 * kPbXcds);
    if (m2 < 1) m2 = 1;
    if (m2 > 16) m2 = 16;
    a2.members2 = m2;
    hipLaunchKernelGGL((k_pb_scatter2<TW, VR>), dim3(bins8 * kPbXcds * m2), dim3(kPbBlock), lds_sc, s, a2);
  }
  const size_t lds_b = (static_cast<size_t>(1) << a.slice_log2) * (4 + 8 * (TW - 1));
  HDK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pb_build<TW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds_b)));
  const int bblock = TW == 1 ? kPbBuildBlock : kPbBuildBlockFused;
  unsigned per_cu = static_cast<unsigned>((160u * 1024u) / (lds_b + 1024));
  if (per_cu < 1) per_cu = 1;
  if (per_cu * bblock > 2048u) per_cu = 2048u / bblock;
  unsigned gb = per_cu * cu;
  if (gb > a.nslices) gb = a.nslices;
  hipLaunchKernelGGL((k_pb_build<TW>), dim3(gb), dim3(bblock), lds_b, s, a);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// rows from which the transparent entry points partition (HDK_HIP_BUILD_PARTITION_MIN_ROWS; 0 = never)
static int64_t pb_min_rows() {
  if (const char* e = hdk_sw(SW_BUILD_PARTITION_MIN_ROWS)) return atoll(e);
  return 2000000;
}

// The whole build on `s`: partitioned passes, then the atomic kernels armed on the overflow flag.  `buff` must have been
// initialised (hdk_hip_init_hash_join_buff), as for the atomic build.  scratch == nullptr: from the stream's pool.
static int32_t one_to_one_partitioned(int32_t* buff, int32_t invalid_slot_val, int32_t* dev_err_buff, const hdk_hip_join_column& jc,
                                      const hdk_hip_join_column_type_info& ti, int64_t bucket, const int8_t* const* pcols,
                                      const int32_t* widths, const int32_t* kinds, int np, int64_t* fused, void* scratch,
                                      size_t scratch_bytes, int32_t device_id, hipStream_t s, bool* done) {
  *done = false;
  PbArgs a;
  memset(&a, 0, sizeof(a));
  PbLayout l;
  const int64_t range = ti.max_val - ti.min_val + 1 + (ti.uses_bw_eq ? 1 : 0);
  const int64_t entries = bucket > 1 ? (range + bucket - 1) / bucket : range;
  if (!pb_geometry(&a, &l, static_cast<int64_t>(jc.num_elems), entries, fused ? np : 0)) return HDK_HIP_OK;
  AsyncScratch own(s);
  int8_t* base = static_cast<int8_t*>(scratch);
  if (!base) {
    if (hipMallocAsync(&own.p, l.total, s) != hipSuccess) {
      (void)hipGetLastError();
      return HDK_HIP_OK;  // no room for the tuples: the atomic build needs none
    }
    base = static_cast<int8_t*>(own.p);
  } else {
    HDK_REQUIRE(scratch_bytes >= l.total, "scratch too small: %zu bytes, hdk_hip_join_build_scratch_bytes says %zu", scratch_bytes, l.total);
  }
  HDK_HIP_CHECK(hipMemsetAsync(base, 0, l.cursor_bytes, s));
  a.chunks = reinterpret_cast<const hdk_hip_join_chunk*>(jc.col_chunks_buff);
  a.num_chunks = jc.num_chunks;
  a.ti = ti;
  a.bucket = bucket;
  a.invalid_slot_val = invalid_slot_val;
  a.dev_err = dev_err_buff;
  for (int c = 0; c < a.np; ++c) {
    a.pcols[c] = pcols[c];
    a.pwidths[c] = widths[c];
    a.pkinds[c] = kinds[c];
  }
  a.flag = reinterpret_cast<uint32_t*>(base);
  a.fill1 = reinterpret_cast<uint32_t*>(base + l.off_fill1);
  a.fill2 = reinterpret_cast<uint32_t*>(base + l.off_fill2);
  a.tuples1 = reinterpret_cast<int64_t*>(base + l.off_t1);
  a.tuples2 = reinterpret_cast<int64_t*>(base + l.off_t2);
  a.buff = buff;
  a.fused = fused;
  const hdk_hip_device_properties* props = device_props(device_id);
  int32_t st;
  switch (1 + a.np) {
    case 1: st = pb_launch<1>(a, props, s); break;
    case 2: st = pb_launch<2>(a, props, s); break;
    case 3: st = pb_launch<3>(a, props, s); break;
    default: st = pb_launch<4>(a, props, s); break;
  }
  if (st) return st;
  // armed: a key distribution that overflowed a sub-slab is built with atomics (the table is still as initialised)
  BuildArgs b;
  b.buff = buff;
  b.hash_entry_count = entries;
  b.invalid_slot_val = invalid_slot_val;
  b.for_semi_join = 0;
  b.dev_err = dev_err_buff;
  b.chunks = a.chunks;
  b.num_chunks = a.num_chunks;
  b.ti = ti;
  b.bucket = bucket;
  b.run_if = a.flag;
  hipLaunchKernelGGL(k_join_build<BUILD_ONE_TO_ONE>, dim3(grid_for(jc.num_elems, device_id)), dim3(kJoinBlock), 0, s, b);
  if (fused) {
    FusedArgs f;
    f.table = buff;
    f.entry_count = entries;
    f.ncols = np;
    f.out = fused;
    f.run_if = a.flag;
    for (int c = 0; c < kMaxFusedCols; ++c) {
      f.cols[c] = c < np ? pcols[c] : nullptr;
      f.widths[c] = c < np ? widths[c] : 0;
      f.kinds[c] = c < np ? kinds[c] : 0;
    }
    hipLaunchKernelGGL(k_build_fused, dim3(grid_for(static_cast<size_t>(entries), device_id)), dim3(kJoinBlock), 0, s, f);
  }
  HDK_HIP_CHECK(hipGetLastError());
  *done = true;
  return HDK_HIP_OK;
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
  a.run_if = nullptr;
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

extern "C" size_t hdk_hip_join_build_scratch_bytes(int64_t num_rows, int64_t entry_count, int32_t ncols) {
  PbArgs a;
  PbLayout l;
  if (!pb_geometry(&a, &l, num_rows, entry_count, ncols)) return 0;
  return l.total;
}

extern "C" int32_t hdk_hip_fill_hash_join_buff_fused(int32_t* buff, int32_t invalid_slot_val, int32_t for_semi_join,
                                                     int32_t* dev_err_buff, hdk_hip_join_column join_column,
                                                     hdk_hip_join_column_type_info type_info, int64_t bucket_normalization,
                                                     const int8_t* const* inner_cols, const int32_t* widths,
                                                     const int32_t* kinds, int32_t ncols, int64_t* fused_out, void* scratch,
                                                     size_t scratch_bytes, int32_t device_id, void* stream) {
  HDK_REQUIRE(buff && dev_err_buff && fused_out, "NULL buffer");
  HDK_REQUIRE(bucket_normalization > 0, "bucket_normalization must be positive");
  HDK_REQUIRE(ncols >= 0 && ncols <= kMaxFusedCols, "0..%d payload columns", kMaxFusedCols);
  HDK_REQUIRE(ncols == 0 || (inner_cols && widths && kinds), "NULL column description");
  int32_t st = check_join_args(join_column, type_info);
  if (st) return st;
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  const int64_t range = type_info.max_val - type_info.min_val + 1 + (type_info.uses_bw_eq ? 1 : 0);
  const int64_t entries = bucket_normalization > 1 ? (range + bucket_normalization - 1) / bucket_normalization : range;
  if (join_column.num_elems && !for_semi_join && ncols <= kPbMaxPayload && pb_min_rows() > 0 &&
      static_cast<int64_t>(join_column.num_elems) >= pb_min_rows()) {
    bool done = false;
    st = one_to_one_partitioned(buff, invalid_slot_val, dev_err_buff, join_column, type_info, bucket_normalization, inner_cols,
                                widths, kinds, ncols, fused_out, scratch, scratch_bytes, device_id, s, &done);
    if (st || done) return st;
  }
  // small, semi-join or wide tables: the table with atomics, then the fused form derived from it
  st = one_to_one(buff, invalid_slot_val, for_semi_join, dev_err_buff, join_column, type_info, bucket_normalization, device_id,
                  stream);
  if (st) return st;
  return hdk_hip_build_fused_join_table(buff, entries, inner_cols, widths, kinds, ncols, fused_out, device_id, stream);
}
