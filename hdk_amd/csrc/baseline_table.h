// baseline_table.h -- open-addressing (baseline hash) group lookup on the device.
//
// Device restatement of get_group_value / get_group_value_columnar_slot with the CUDA twin's claim
// protocol (reference QE/GroupByRuntime.cpp:31-113; QE/cuda_mapd_rt.cu:167-261): MurmurHash3 of the
// packed key, linear probe, CAS on the first key component, remaining components published by the
// winner and awaited by readers.
#pragma once
#include "device_common.h"

namespace hdk {

// key_hash = MurmurHash3(key, width*count, 0) (QE/GroupByRuntime.cpp:24-29) over the packed key
// components.  The 32-bit words are taken from the typed values (no pointer punning: reading an
// int64 key array through a uint32 pointer lets the optimiser drop the stores that filled it).
template <typename K>
HDK_DEV uint32_t key_hash_dev(const K* key, int key_count) {
  uint32_t h1 = 0;
  const uint32_t c1 = 0xcc9e2d51;
  const uint32_t c2 = 0x1b873593;
  auto mix = [&](uint32_t k1) {
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64;
  };
  for (int i = 0; i < key_count; ++i) {
    if constexpr (sizeof(K) == 8) {
      const uint64_t v = static_cast<uint64_t>(key[i]);
      mix(static_cast<uint32_t>(v));
      mix(static_cast<uint32_t>(v >> 32));
    } else {
      mix(static_cast<uint32_t>(key[i]));
    }
  }
  h1 ^= static_cast<uint32_t>(key_count * sizeof(K));
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6b;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35;
  h1 ^= h1 >> 16;
  return h1;
}

template <typename K>
HDK_DEV K empty_key();
template <>
HDK_DEV int64_t empty_key<int64_t>() { return HDK_EMPTY_KEY_64; }
template <>
HDK_DEV int32_t empty_key<int32_t>() { return HDK_EMPTY_KEY_32; }

HDK_DEV int64_t atomic_load_i64(const int64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
HDK_DEV int32_t atomic_load_i32(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The plan fields find_or_claim needs, read once per kernel, pinned to SGPRs and expressed as integer
// strides -- no boolean survives into the probe loop.  Background: with `columnar` kept as a bool,
// hipcc (ROCm 7.2) materialises `!columnar` as a lane mask (v_cndmask + v_cmp) INSIDE the probe loop,
// i.e. under that trip's shrinking EXEC, and when two inlined copies of find_or_claim follow each other
// it reuses the first copy's mask for the second copy's uniform branch (`s_and vcc, exec, mask`): lanes
// that left the first loop early then address the table with the wrong layout.  Seen as OUT_OF_SLOTS and
// a 50x slowdown in hdk_part_aggregate once it applied two tuples per trip.
struct TableShape {
  int key_count;
  uint32_t row_quads;
  uint32_t entry_stride_bytes;  // from one entry's first key component to the next entry's
  uint32_t columnar_mask;       // all ones for a columnar table, else 0: key stride = (entry_count & mask) | (~mask & 1)
};
HDK_DEV TableShape table_shape(const hdk_hip_plan* p) {
  TableShape s;
  s.key_count = __builtin_amdgcn_readfirstlane(p->key_count);
  s.row_quads = __builtin_amdgcn_readfirstlane(p->row_size_quad);
  const uint32_t columnar = __builtin_amdgcn_readfirstlane(static_cast<int>(p->output_columnar)) != 0 ? 1u : 0u;
  const uint32_t key_width = __builtin_amdgcn_readfirstlane(p->key_width);
  s.columnar_mask = 0u - columnar;
  s.entry_stride_bytes = (key_width & s.columnar_mask) | ((s.row_quads * 8u) & ~s.columnar_mask);
  return s;
}

// Find or claim the entry of `key` (packed components of width K) in a baseline table.
// Returns the entry index or -1 (table full); *fresh tells whether this call created it.
// Row-wise (get_matching_group_value, QE/cuda_mapd_rt.cu:167-203): CAS the first component, write
// the rest, readers spin until the last component is published.  Columnar
// (get_matching_group_value_columnar_slot, :229-261) likewise per key column.
// WRAP = true: the reference's probe sequence over the whole table, (start + i) % entry_count, -1 once every entry
// has been looked at.  WRAP = false: probe from `start` to the END of the `entry_count` entries and stop there (-1):
// a table region aggregated on its own (scan_agg_partitioned.h) places a group exactly where the whole-table
// sequence would, or not at all.
template <typename K, bool WRAP>
HDK_DEV int64_t find_or_claim_from(const TableShape shape, int64_t* buf, uint32_t entry_count, uint32_t start, const K* key,
                                   bool* fresh) {
  const int nk = shape.key_count;
  uint32_t probe = start;
  const K ek = empty_key<K>();
  uint32_t steps = 0;
  int64_t result = -2;  // -2: still probing
  *fresh = false;
  // Wave-safe form of the reference's "winner publishes, readers spin" protocol: a reader that finds
  // a half-published key does NOT spin inside the iteration (the publisher may be a lane of the same
  // wave, masked off until the branches reconverge) -- it re-examines the same slot on the next trip
  // of this loop, by which time every lane of the wave has passed the publishing block.
  while (result == -2) {
    K* k0 = reinterpret_cast<K*>(reinterpret_cast<int8_t*>(buf) + static_cast<size_t>(probe) * shape.entry_stride_bytes);
    const size_t kstride = (entry_count & shape.columnar_mask) | (~shape.columnar_mask & 1u);
    // Look before claiming: a plain load is several times cheaper than an atomic on this part
    // (scripts/microbench/atomics.hip: ~2.4e10 atomics/s chip-wide vs >5e10 random loads/s), and once
    // a group exists every later row of it only needs the load.  A slot never returns to EMPTY.
    K old;
    if constexpr (sizeof(K) == 8) {
      old = static_cast<K>(atomic_load_i64(reinterpret_cast<const int64_t*>(k0)));
    } else {
      old = static_cast<K>(atomic_load_i32(reinterpret_cast<const int32_t*>(k0)));
    }
    if (old == ek) {
      if constexpr (sizeof(K) == 8) {
        old = static_cast<K>(atomicCAS(reinterpret_cast<unsigned long long*>(k0), static_cast<unsigned long long>(ek),
                                       static_cast<unsigned long long>(key[0])));
      } else {
        old = static_cast<K>(atomicCAS(reinterpret_cast<unsigned int*>(k0), static_cast<unsigned int>(ek),
                                       static_cast<unsigned int>(key[0])));
      }
    }
    const bool won = old == ek;
    if (won) {
      for (int i = 1; i < nk; ++i) {
        __hip_atomic_store(k0 + i * kstride, key[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *fresh = true;
      result = probe;
    }
    bool advance = !won;
    if (!won && old == key[0]) {
      bool match = true;
      bool pending = false;
      for (int i = 1; i < nk; ++i) {
        K v;
        if constexpr (sizeof(K) == 8) {
          v = atomic_load_i64(reinterpret_cast<const int64_t*>(k0 + i * kstride));
        } else {
          v = atomic_load_i32(reinterpret_cast<const int32_t*>(k0 + i * kstride));
        }
        if (v == ek && key[i] != ek) {
          pending = true;  // the winner has not published this component yet
        } else if (v != key[i]) {
          match = false;
        }
      }
      if (!match) {
        advance = true;
      } else if (pending) {
        advance = false;  // look at this slot again
      } else {
        result = probe;
        advance = false;
      }
    }
    if (result == -2 && advance) {
      if (WRAP) {
        probe = probe + 1 == entry_count ? 0 : probe + 1;
        if (++steps >= entry_count) {
          result = -1;  // wrapped around: table full
        }
      } else if (++probe == entry_count) {
        result = -1;  // ran off the end of the region
      }
    }
  }
  return result;
}

template <typename K>
HDK_DEV int64_t find_or_claim(const TableShape shape, int64_t* buf, uint32_t entry_count, const K* key, bool* fresh) {
  return find_or_claim_from<K, true>(shape, buf, entry_count, key_hash_dev<K>(key, shape.key_count) % entry_count, key, fresh);
}

template <typename K>
HDK_DEV int64_t find_or_claim(const hdk_hip_plan* p, int64_t* buf, uint32_t entry_count, const K* key, bool* fresh) {
  return find_or_claim<K>(table_shape(p), buf, entry_count, key, fresh);
}

}  // namespace hdk
