// reduce.hip -- merge partial result buffers on the device.
//
// New kernels for what the reference does on the host with tbb (ResultSetReduction::reduce,
// QE/ResultSetReduction.cpp:174-330, driven by Executor::reduceMultiDeviceResultSets,
// QE/Execute.cpp:1224-1336): per-GPU / per-launch partial buffers that already sit in HBM (e.g.
// after an RCCL all-gather or all-to-all) are folded into `this_buf` without a D->H round trip.
//   perfect hash / non-grouped: entry-wise, slot-wise reduceOneSlot (:1234-1330) -- one thread per
//       entry walks the partials in argument order, so the result does not depend on scheduling;
//   baseline hash: every non-empty entry of each partial is re-inserted (reduceOneEntryBaseline,
//       :694-731): CAS-claim of the key in `this_buf`, fill_slots for a fresh entry, reduceOneSlot
//       otherwise.  Partials are merged one launch after another; inside one partial keys are
//       unique, so an entry of `this_buf` has a single writer per launch.
#include <mutex>
#include <string.h>

#include "baseline_table.h"
#include "device_common.h"
#include "host_common.h"

namespace hdk {

// Device copies of the plans the reductions run with.  A merge step is issued once per query step with the same
// plan; allocating scratch and uploading 6 KB from pageable memory on every call put a host-side stall of the
// whole stream into each step (scripts/multi_gpu_floor.py: host enqueue time = GPU time).  A handful of plans per
// device are kept (byte-compared); the first upload is a blocking copy, so the copy is valid for every stream
// afterwards.  Entries are recycled round-robin, never while a call that got one can still be running: a slot is
// reused only after kPlanCacheSlots other plans have come by, and the reductions are short.
constexpr int kPlanCacheSlots = 16;
struct PlanCacheEntry {
  hdk_hip_plan host;
  hdk_hip_plan* dev = nullptr;
  bool used = false;
};
struct PlanCache {
  std::mutex mu;
  PlanCacheEntry e[kPlanCacheSlots];
  int next = 0;
};
static PlanCache g_plan_cache[16];

static int32_t cached_device_plan(const hdk_hip_plan* plan, int32_t device_id, const hdk_hip_plan** out) {
  PlanCache& c = g_plan_cache[device_id & 15];
  std::lock_guard<std::mutex> lk(c.mu);
  for (int i = 0; i < kPlanCacheSlots; ++i) {
    if (c.e[i].used && memcmp(&c.e[i].host, plan, sizeof(hdk_hip_plan)) == 0) {
      *out = c.e[i].dev;
      return HDK_HIP_OK;
    }
  }
  PlanCacheEntry& e = c.e[c.next];
  c.next = (c.next + 1) % kPlanCacheSlots;
  if (!e.dev) {
    HDK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e.dev), sizeof(hdk_hip_plan)));
  }
  e.used = false;
  HDK_HIP_CHECK(hipDeviceSynchronize());  // (a recycled slot: nothing still reads the old copy)
  HDK_HIP_CHECK(hipMemcpy(e.dev, plan, sizeof(hdk_hip_plan), hipMemcpyHostToDevice));
  memcpy(&e.host, plan, sizeof(hdk_hip_plan));
  e.used = true;
  *out = e.dev;
  return HDK_HIP_OK;
}

constexpr int kRedBlock = 256;
constexpr int kMaxSlots = 2 * HDK_HIP_MAX_TARGETS;

struct SlotInit {
  int64_t v[kMaxSlots];
};

// reduceOneSlotSingleValue (QE/ResultSetReduction.cpp:1186-1230): by slot width, against the slot's init value; true =
// two different values ("Multiple distinct values encountered")
HDK_DEV bool reduce_single_value(const hdk_hip_target& tg, int8_t* this1, const int8_t* that1, int64_t init_val) {
  if (tg.slot_width == 4) {
    const int32_t l = *reinterpret_cast<const int32_t*>(this1), r = *reinterpret_cast<const int32_t*>(that1);
    if (r == static_cast<int32_t>(init_val)) return false;
    if (l == static_cast<int32_t>(init_val)) {
      *reinterpret_cast<int32_t*>(this1) = r;
      return false;
    }
    return l != r;
  }
  const int64_t l = *reinterpret_cast<const int64_t*>(this1), r = *reinterpret_cast<const int64_t*>(that1);
  if (r == init_val) return false;
  if (l == init_val) {
    *reinterpret_cast<int64_t*>(this1) = r;
    return false;
  }
  return l != r;
}

// reduceOneSlot with AGGREGATE_ONE_[NULLABLE_]VALUE / AGGREGATE_ONE_COUNT
// (QE/ResultSetReduction.cpp:1026-1107,1234-1385)
HDK_DEV void reduce_slot(const hdk_hip_target& tg, int8_t* this1, int8_t* this2, const int8_t* that1,
                         const int8_t* that2, int64_t init_val) {
  const int agg = tg.agg;
  if (agg == HDK_AGG_ID) {
    if (tg.slot_width == 0) {  // projected key of a baseline table: no slot (target_groupby_indices)
      return;
    }
    if (tg.slot_width == 4) {
      const int32_t rhs = *reinterpret_cast<const int32_t*>(that1);
      if (rhs != init_val) *reinterpret_cast<int32_t*>(this1) = rhs;
    } else {
      const int64_t rhs = *reinterpret_cast<const int64_t*>(that1);
      if (rhs != init_val) *reinterpret_cast<int64_t*>(this1) = rhs;
    }
    return;
  }
  if (agg == HDK_AGG_COUNT || agg == HDK_AGG_AVG) {
    int8_t* cs = agg == HDK_AGG_COUNT ? this1 : this2;
    const int8_t* co = agg == HDK_AGG_COUNT ? that1 : that2;
    const int cw = agg == HDK_AGG_COUNT ? tg.slot_width : tg.slot2_width;
    if (cw == 4) {
      *reinterpret_cast<uint32_t*>(cs) += *reinterpret_cast<const uint32_t*>(co);
    } else {
      *reinterpret_cast<uint64_t*>(cs) += *reinterpret_cast<const uint64_t*>(co);
    }
    if (agg == HDK_AGG_COUNT) {
      return;
    }
  }
  const bool skip = tg.skip_null;
  if (tg.slot_width == 1 || tg.slot_width == 2) {
    // logical-sized MIN / MAX slots (AGGREGATE_ONE_NULLABLE_VALUE_SMALL, QE/ResultSetReduction.cpp:1136-1172)
    small_min_max(agg, skip, tg.slot_width, this1, tg.slot_width == 2 ? static_cast<int64_t>(*reinterpret_cast<const int16_t*>(that1))
                                                                      : static_cast<int64_t>(*that1), init_val);
    return;
  }
  if (tg.arg_is_fp == HDK_FP_SLOT_FLOAT) {
    // get_width_for_slot -> sizeof(float) (QE/ResultSetReduction.cpp:1176-1185): AGGREGATE_ONE_NULLABLE_VALUE on the
    // low 4 bytes with agg_*_float[_skip_val]; the skip value is the low half of the slot's init value
    int32_t* s = reinterpret_cast<int32_t*>(this1);
    const int32_t obits = *reinterpret_cast<const int32_t*>(that1);
    const int32_t nbits = static_cast<int32_t>(init_val);
    const float o = __int_as_float(obits);
    if (skip && o == __int_as_float(nbits)) return;  // `val != skip_val` is a value compare
    const int32_t old = *s;
    if (skip && old == nbits) {                      // the accumulator is compared bit-wise
      *s = obits;
      return;
    }
    const float a = __int_as_float(old);
    float r;
    if (agg == HDK_AGG_MIN) {
      r = (o < a) ? o : a;
    } else if (agg == HDK_AGG_MAX) {
      r = (a < o) ? o : a;
    } else {
      r = a + o;
    }
    *s = __float_as_int(r);
    return;
  }
  if (tg.slot_width == 4) {  // integers
    int32_t* s = reinterpret_cast<int32_t*>(this1);
    const int32_t o = *reinterpret_cast<const int32_t*>(that1);
    const int32_t nv = static_cast<int32_t>(init_val);
    if (skip && o == nv) return;
    const int32_t old = *s;
    if (skip && old == nv) {
      *s = o;
    } else if (agg == HDK_AGG_MIN) {
      *s = old < o ? old : o;
    } else if (agg == HDK_AGG_MAX) {
      *s = old > o ? old : o;
    } else {
      *s = static_cast<int32_t>(static_cast<uint32_t>(old) + static_cast<uint32_t>(o));
    }
    return;
  }
  int64_t* s = reinterpret_cast<int64_t*>(this1);
  const int64_t o = *reinterpret_cast<const int64_t*>(that1);
  const int64_t old = *s;
  if (tg.arg_is_fp) {
    const double od = bits_to_double(o);
    if (skip && od == bits_to_double(init_val)) return;  // `val != skip_val` is a value compare
    if (skip && old == init_val) {                        // the accumulator is compared bit-wise
      *s = o;
      return;
    }
    const double a = bits_to_double(old);
    double r;
    if (agg == HDK_AGG_MIN) {
      r = (od < a) ? od : a;
    } else if (agg == HDK_AGG_MAX) {
      r = (a < od) ? od : a;
    } else {
      r = a + od;
    }
    *s = double_to_bits(r);
    return;
  }
  if (skip && o == init_val) return;
  if (skip && old == init_val) {
    *s = o;
  } else if (agg == HDK_AGG_MIN) {
    *s = old < o ? old : o;
  } else if (agg == HDK_AGG_MAX) {
    *s = old > o ? old : o;
  } else {
    *s = static_cast<int64_t>(static_cast<uint64_t>(old) + static_cast<uint64_t>(o));
  }
}

HDK_DEV int64_t read_slot(const int8_t* p, int w) {
  return w == 4 ? static_cast<int64_t>(*reinterpret_cast<const int32_t*>(p)) : *reinterpret_cast<const int64_t*>(p);
}

HDK_DEV void slot_ptrs(const hdk_hip_plan* p, int64_t* buf, uint32_t entry_count, uint32_t entry, int t,
                       int first_slot, int8_t** s1, int8_t** s2) {
  const hdk_hip_target& tg = p->targets[t];
  if (p->query_kind == HDK_Q_NON_GROUPED) {
    *s1 = reinterpret_cast<int8_t*>(buf + first_slot);
    *s2 = reinterpret_cast<int8_t*>(buf + first_slot + 1);
  } else if (p->output_columnar) {
    *s1 = reinterpret_cast<int8_t*>(buf) + columnar_slot_off(p, entry_count, first_slot) +
          static_cast<size_t>(entry) * tg.slot_width;
    *s2 = tg.agg == HDK_AGG_AVG ? reinterpret_cast<int8_t*>(buf) + columnar_slot_off(p, entry_count, first_slot + 1) +
                                      static_cast<size_t>(entry) * tg.slot2_width
                                : nullptr;
  } else {
    int8_t* row = reinterpret_cast<int8_t*>(buf + static_cast<size_t>(entry) * p->row_size_quad);
    *s1 = row + tg.slot_off;
    *s2 = row + tg.slot2_off;
  }
}

// ResultSetStorage::isEmptyEntry[Columnar] (RS/ResultSetStorage.cpp:439-521)
HDK_DEV bool is_empty_entry(const hdk_hip_plan* p, const int64_t* buf, uint32_t entry_count, uint32_t e,
                            const SlotInit& init) {
  if (p->query_kind == HDK_Q_NON_GROUPED) {
    return false;
  }
  if (p->keyless) {
    const int ks = p->idx_target_as_key;
    int s = 0;
    const int nt = p->num_targets;
    for (int t = 0; t < nt; ++t) {
      const hdk_hip_target& tg = p->targets[t];
      const int n = tg.agg == HDK_AGG_AVG ? 2 : 1;
      if (ks < s + n) {
        int8_t *s1, *s2;
        slot_ptrs(p, const_cast<int64_t*>(buf), entry_count, e, t, s, &s1, &s2);
        const bool second = ks != s;
        const int w = second ? tg.slot2_width : tg.slot_width;
        int64_t iv = init.v[0];
#pragma unroll
        for (int k = 1; k < kMaxSlots; ++k) {
          if (k == ks) iv = init.v[k];
        }
        if (w == 4) iv = static_cast<int32_t>(iv);
        return read_slot(second ? s2 : s1, w) == iv;
      }
      s += n;
    }
    return true;
  }
  if (p->output_columnar) {
    return buf[e] == HDK_EMPTY_KEY_64;
  }
  const int64_t* keys = buf + static_cast<size_t>(e) * p->row_size_quad;
  return p->key_width == 4 ? *reinterpret_cast<const int32_t*>(keys) == HDK_EMPTY_KEY_32
                           : *keys == HDK_EMPTY_KEY_64;
}

HDK_DEV int64_t slot_init(const SlotInit& init, int idx) {
  int64_t iv = init.v[0];
#pragma unroll
  for (int k = 1; k < kMaxSlots; ++k) {
    if (k == idx) iv = init.v[k];
  }
  return iv;
}

constexpr int kMaxThat = 16;
struct ThatList {
  const int64_t* buf[kMaxThat];
  int32_t n;
};

// perfect hash / non-grouped: one thread per entry, partials in argument order
__global__ __launch_bounds__(kRedBlock) void k_reduce_entrywise(const hdk_hip_plan* __restrict__ p, int64_t* this_buf,
                                                                uint32_t entry_count, ThatList that, SlotInit init,
                                                                int32_t* dev_error) {
  const uint32_t e = blockIdx.x * kRedBlock + threadIdx.x;
  if (e >= entry_count) {
    return;
  }
  const int nt = p->num_targets;
  const int nk = p->key_count;
  for (int b = 0; b < that.n; ++b) {
    const int64_t* tb = that.buf[0];
#pragma unroll
    for (int k = 1; k < kMaxThat; ++k) {
      if (k == b) tb = that.buf[k];
    }
    if (is_empty_entry(p, tb, entry_count, e, init)) {
      continue;
    }
    if (p->query_kind == HDK_Q_PERFECT_HASH && !p->keyless) {  // copyKeyColWise / rowwise key copy
      if (p->output_columnar) {
        for (int k = 0; k < nk; ++k) {
          this_buf[static_cast<size_t>(k) * entry_count + e] = tb[static_cast<size_t>(k) * entry_count + e];
        }
      } else {
        for (int k = 0; k < nk; ++k) {
          this_buf[static_cast<size_t>(e) * p->row_size_quad + k] = tb[static_cast<size_t>(e) * p->row_size_quad + k];
        }
      }
    }
    int s = 0;
    for (int t = 0; t < nt; ++t) {
      int8_t *a1, *a2, *b1, *b2;
      slot_ptrs(p, this_buf, entry_count, e, t, s, &a1, &a2);
      slot_ptrs(p, const_cast<int64_t*>(tb), entry_count, e, t, s, &b1, &b2);
      if (p->targets[t].agg == HDK_AGG_SINGLE_VALUE) {
        if (reduce_single_value(p->targets[t], a1, b1, slot_init(init, s))) {
          record_error(dev_error, HDK_HIP_ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES);
        }
      } else {
        reduce_slot(p->targets[t], a1, a2, b1, b2, slot_init(init, s));
      }
      s += p->targets[t].agg == HDK_AGG_AVG ? 2 : 1;
    }
  }
}

// ---- baseline re-insert (find_or_claim: baseline_table.h) ------------------------------------------
template <typename K>
__global__ __launch_bounds__(kRedBlock) void k_reduce_baseline(const hdk_hip_plan* __restrict__ p, int64_t* this_buf,
                                                               uint32_t this_entry_count, const int64_t* that_buf,
                                                               uint32_t that_entry_count, SlotInit init,
                                                               int32_t* dev_error) {
  const uint32_t stride = gridDim.x * kRedBlock;
  const int nk = p->key_count;
  const int nt = p->num_targets;
  for (uint32_t e = blockIdx.x * kRedBlock + threadIdx.x; e < that_entry_count; e += stride) {
    if (is_empty_entry(p, that_buf, that_entry_count, e, init)) {
      continue;
    }
    K key[HDK_HIP_MAX_KEYS];
#pragma unroll
    for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
      if (k < nk) {
        key[k] = p->output_columnar
                     ? reinterpret_cast<const K*>(that_buf)[static_cast<size_t>(k) * that_entry_count + e]
                     : reinterpret_cast<const K*>(that_buf + static_cast<size_t>(e) * p->row_size_quad)[k];
      } else {
        key[k] = 0;
      }
    }
    bool fresh = false;
    const int64_t dst = find_or_claim<K>(p, this_buf, this_entry_count, key, &fresh);
    if (dst < 0) {
      record_error(dev_error, HDK_HIP_ERR_OUT_OF_SLOTS);
      continue;
    }
    int s = 0;
    for (int t = 0; t < nt; ++t) {
      const hdk_hip_target& tg = p->targets[t];
      int8_t *a1, *a2, *b1, *b2;
      slot_ptrs(p, this_buf, this_entry_count, static_cast<uint32_t>(dst), t, s, &a1, &a2);
      slot_ptrs(p, const_cast<int64_t*>(that_buf), that_entry_count, e, t, s, &b1, &b2);
      if (fresh) {  // fill_slots (ResultSetReduction.cpp:560-600): plain copy into the new entry
        if (tg.slot_width == 0) {
          // no slot
        } else if (tg.slot_width == 4) {
          *reinterpret_cast<int32_t*>(a1) = *reinterpret_cast<const int32_t*>(b1);
        } else {
          *reinterpret_cast<int64_t*>(a1) = *reinterpret_cast<const int64_t*>(b1);
        }
        if (tg.agg == HDK_AGG_AVG) {
          if (tg.slot2_width == 4) {
            *reinterpret_cast<int32_t*>(a2) = *reinterpret_cast<const int32_t*>(b2);
          } else {
            *reinterpret_cast<int64_t*>(a2) = *reinterpret_cast<const int64_t*>(b2);
          }
        }
      } else if (tg.agg == HDK_AGG_SINGLE_VALUE) {
        if (reduce_single_value(tg, a1, b1, slot_init(init, s))) {
          record_error(dev_error, HDK_HIP_ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES);
        }
      } else {
        reduce_slot(tg, a1, a2, b1, b2, slot_init(init, s));
      }
      s += tg.agg == HDK_AGG_AVG ? 2 : 1;
    }
  }
}


// ---- owner partition of a baseline table (multi-GPU exchange step, include/hdk_hip.h) ---------------
constexpr int kMaxOwners = 64;

struct OwnerSegs {
  int64_t* buf[kMaxOwners];
  uint32_t count[kMaxOwners];
};

// owner of a key: the high part of hash * G, so an owner's keys still spread over every home slot
// h % entry_count of its table (h % G would leave it only the slots congruent to the owner id)
HDK_DEV uint32_t owner_of(uint32_t h, uint32_t num_owners) {
  return static_cast<uint32_t>((static_cast<uint64_t>(h) * num_owners) >> 32);
}

template <typename K>
HDK_DEV uint32_t entry_owner(const hdk_hip_plan* p, const int64_t* buf, uint32_t entry_count, uint32_t e,
                             uint32_t num_owners, K (&key)[HDK_HIP_MAX_KEYS]) {
  const int nk = p->key_count;
#pragma unroll
  for (int k = 0; k < HDK_HIP_MAX_KEYS; ++k) {
    if (k < nk) {
      key[k] = p->output_columnar ? reinterpret_cast<const K*>(buf)[static_cast<size_t>(k) * entry_count + e]
                                  : reinterpret_cast<const K*>(buf + static_cast<size_t>(e) * p->row_size_quad)[k];
    } else {
      key[k] = 0;
    }
  }
  return owner_of(key_hash_dev<K>(key, nk), num_owners);
}

// MODE 0: count entries per owner; MODE 1: scatter them into the owners' segments.
// Per tile of kRedBlock x kPartT entries: positions inside the tile come from LDS counters (one returning LDS atomic
// per wave and owner: lanes holding the same owner are ranked with a ballot), the tile's share of every owner's
// segment from ONE global atomic per owner.  (Round 2 took a global atomic per wave and owner: 25 M returning
// atomics on eight addresses for a 200 M-entry table -- 266 ms per pass, profiles/r03_multi_gpu_floor_before.json.)
constexpr int kPartT = 8;
template <typename K, int MODE>
__global__ __launch_bounds__(kRedBlock) void k_partition_baseline(const hdk_hip_plan* __restrict__ p,
                                                                  const int64_t* buf, uint32_t entry_count,
                                                                  uint32_t num_owners, SlotInit init,
                                                                  uint32_t* cursors, OwnerSegs segs) {
  __shared__ uint32_t s_cnt[kMaxOwners], s_base[kMaxOwners];
  const uint32_t lane = threadIdx.x & 63;
  const int nk = p->key_count;
  const int nt = p->num_targets;
  constexpr uint32_t kTile = kRedBlock * kPartT;
  const uint32_t ntiles = (entry_count + kTile - 1) / kTile;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (threadIdx.x < kMaxOwners) {
      s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    uint32_t owner[kPartT], rank[kPartT];
#pragma unroll
    for (int j = 0; j < kPartT; ++j) {
      const uint32_t e = tile * kTile + static_cast<uint32_t>(j) * kRedBlock + threadIdx.x;
      K key[HDK_HIP_MAX_KEYS];
      owner[j] = 0xffffffffu;
      rank[j] = 0;
      if (e < entry_count && !is_empty_entry(p, buf, entry_count, e, init)) {
        owner[j] = entry_owner<K>(p, buf, entry_count, e, num_owners, key);
      }
      for (uint32_t o = 0; o < num_owners; ++o) {
        const uint64_t mask = __ballot(owner[j] == o);
        if (mask == 0) {
          continue;
        }
        const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(mask))) - 1;
        uint32_t base = 0;
        if (lane == leader) {
          base = atomicAdd(&s_cnt[o], static_cast<uint32_t>(__popcll(mask)));
        }
        base = __shfl(base, leader, 64);
        if (owner[j] == o) {
          rank[j] = base + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1)));
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < num_owners) {
      const uint32_t n = s_cnt[threadIdx.x];
      s_base[threadIdx.x] = n ? atomicAdd(cursors + threadIdx.x, n) : 0;
    }
    __syncthreads();
    if (MODE == 0) {
      continue;  // (the barrier at the top of the next tile separates the reads of s_base from its next writes)
    }
#pragma unroll
    for (int j = 0; j < kPartT; ++j) {
      if (owner[j] == 0xffffffffu) {
        continue;
      }
      const uint32_t e = tile * kTile + static_cast<uint32_t>(j) * kRedBlock + threadIdx.x;
      const uint32_t dst = s_base[owner[j]] + rank[j];
      int64_t* seg = segs.buf[0];
      uint32_t seg_n = segs.count[0];
#pragma unroll
      for (int k = 1; k < kMaxOwners; ++k) {
        if (static_cast<uint32_t>(k) == owner[j]) {
          seg = segs.buf[k];
          seg_n = segs.count[k];
        }
      }
      for (int k = 0; k < nk; ++k) {
        const K kv = p->output_columnar ? reinterpret_cast<const K*>(buf)[static_cast<size_t>(k) * entry_count + e]
                                        : reinterpret_cast<const K*>(buf + static_cast<size_t>(e) * p->row_size_quad)[k];
        if (p->output_columnar) {
          reinterpret_cast<K*>(seg)[static_cast<size_t>(k) * seg_n + dst] = kv;
        } else {
          reinterpret_cast<K*>(seg + static_cast<size_t>(dst) * p->row_size_quad)[k] = kv;
        }
      }
      int sl = 0;
      for (int t = 0; t < nt; ++t) {
        const hdk_hip_target& tg = p->targets[t];
        int8_t *a1, *a2, *b1, *b2;
        slot_ptrs(p, seg, seg_n, dst, t, sl, &a1, &a2);
        slot_ptrs(p, const_cast<int64_t*>(buf), entry_count, e, t, sl, &b1, &b2);
        if (tg.slot_width == 0) {
          // no slot
        } else if (tg.slot_width == 4) {
          *reinterpret_cast<int32_t*>(a1) = *reinterpret_cast<const int32_t*>(b1);
        } else {
          *reinterpret_cast<int64_t*>(a1) = *reinterpret_cast<const int64_t*>(b1);
        }
        if (tg.agg == HDK_AGG_AVG) {
          if (tg.slot2_width == 4) {
            *reinterpret_cast<int32_t*>(a2) = *reinterpret_cast<const int32_t*>(b2);
          } else {
            *reinterpret_cast<int64_t*>(a2) = *reinterpret_cast<const int64_t*>(b2);
          }
        }
        sl += tg.agg == HDK_AGG_AVG ? 2 : 1;
      }
    }
  }
}

static size_t table_quads(const hdk_hip_plan* p, uint32_t entry_count) {
  if (!p->output_columnar) {
    return static_cast<size_t>(entry_count) * p->row_size_quad;
  }
  int nslots = 0;
  for (int t = 0; t < p->num_targets; ++t) {
    nslots += p->targets[t].agg == HDK_AGG_AVG ? 2 : 1;
  }
  size_t off = p->keyless ? 0 : static_cast<size_t>(p->key_count) * ((static_cast<size_t>(entry_count) * 8 + 7) & ~size_t(7));
  int s = 0;
  for (int t = 0; t < p->num_targets && s < nslots; ++t) {
    off = (off + 7) & ~size_t(7);
    off += static_cast<size_t>(entry_count) * p->targets[t].slot_width;
    ++s;
    if (p->targets[t].agg == HDK_AGG_AVG) {
      off = (off + 7) & ~size_t(7);
      off += static_cast<size_t>(entry_count) * p->targets[t].slot2_width;
      ++s;
    }
  }
  return ((off + 7) & ~size_t(7)) / 8;
}

static void fill_slot_init(const hdk_hip_plan* plan, const int64_t* init_vals, SlotInit* init) {
  int nslots = 0;
  for (int t = 0; t < plan->num_targets; ++t) {
    nslots += plan->targets[t].agg == HDK_AGG_AVG ? 2 : 1;
  }
  for (int i = 0; i < kMaxSlots; ++i) {
    init->v[i] = i < nslots ? init_vals[i] : 0;
  }
}

template <int MODE>
static int32_t run_partition(const hdk_hip_plan* plan, const int64_t* buf, uint32_t entry_count, const int64_t* init_vals,
                             int32_t num_owners, uint32_t* counts_out, const OwnerSegs& segs, int32_t device_id,
                             hipStream_t s) {
  SlotInit init;
  fill_slot_init(plan, init_vals, &init);
  AsyncScratch mem(s);  // [plan | cursors]
  constexpr size_t kPlanBytes = (sizeof(hdk_hip_plan) + 255) & ~size_t(255);
  HDK_HIP_CHECK(hipMallocAsync(&mem.p, kPlanBytes + kMaxOwners * sizeof(uint32_t), s));
  int8_t* scratch = static_cast<int8_t*>(mem.p);
  hdk_hip_plan* d_plan = reinterpret_cast<hdk_hip_plan*>(scratch);
  uint32_t* cursors = reinterpret_cast<uint32_t*>(scratch + kPlanBytes);
  HDK_HIP_CHECK(hipMemcpyAsync(d_plan, plan, sizeof(hdk_hip_plan), hipMemcpyHostToDevice, s));
  HDK_HIP_CHECK(hipMemsetAsync(cursors, 0, kMaxOwners * sizeof(uint32_t), s));
  const hdk_hip_device_properties* props = device_props(device_id);
  size_t blocks = (static_cast<size_t>(entry_count) + kRedBlock * kPartT - 1) / (kRedBlock * kPartT);
  const size_t cap = static_cast<size_t>(props->num_cu) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  const bool k32 = !plan->output_columnar && plan->key_width == 4;
  if (k32) {
    hipLaunchKernelGGL((k_partition_baseline<int32_t, MODE>), dim3(static_cast<unsigned>(blocks)), dim3(kRedBlock), 0, s,
                       d_plan, buf, entry_count, static_cast<uint32_t>(num_owners), init, cursors, segs);
  } else {
    hipLaunchKernelGGL((k_partition_baseline<int64_t, MODE>), dim3(static_cast<unsigned>(blocks)), dim3(kRedBlock), 0, s,
                       d_plan, buf, entry_count, static_cast<uint32_t>(num_owners), init, cursors, segs);
  }
  HDK_HIP_CHECK(hipGetLastError());
  if (counts_out) {
    HDK_HIP_CHECK(hipMemcpyAsync(counts_out, cursors, num_owners * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HDK_HIP_CHECK(hipStreamSynchronize(s));
  }
  return HDK_HIP_OK;
}

int32_t validate_plan_layout(const hdk_hip_plan* p);  // scan_agg.hip: the layout half of the plan check

}  // namespace hdk

using namespace hdk;

extern "C" int32_t hdk_hip_reduce_buffers(const hdk_hip_plan* plan, int64_t* this_buf, uint32_t this_entry_count,
                                          const int64_t* const* that_bufs, const uint32_t* that_entry_counts,
                                          int32_t num_that, const int64_t* init_vals, int32_t* dev_error,
                                          int32_t device_id, void* stream) {
  int32_t st = validate_plan_layout(plan);
  if (st) return st;
  HDK_REQUIRE(this_buf && (num_that == 0 || (that_bufs && that_entry_counts)) && init_vals, "NULL argument");
  HDK_REQUIRE(num_that >= 0, "num_that must be >= 0");
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  if (num_that == 0) {
    return HDK_HIP_OK;
  }
  int nslots = 0;
  for (int t = 0; t < plan->num_targets; ++t) {
    nslots += plan->targets[t].agg == HDK_AGG_AVG ? 2 : 1;
  }
  SlotInit init;
  for (int i = 0; i < kMaxSlots; ++i) {
    init.v[i] = i < nslots ? init_vals[i] : 0;  // init_vals is a HOST array (ResultSetStorage::target_init_vals_)
  }
  // device copy of the plan (cached per device)
  const hdk_hip_plan* d_plan = nullptr;
  st = cached_device_plan(plan, device_id, &d_plan);
  if (st) return st;
  const hdk_hip_device_properties* props = device_props(device_id);
  if (plan->query_kind != HDK_Q_BASELINE_HASH) {
    const uint32_t n = plan->query_kind == HDK_Q_NON_GROUPED ? 1u : this_entry_count;
    for (int i = 0; i < num_that; ++i) {
      if (plan->query_kind != HDK_Q_NON_GROUPED && that_entry_counts[i] != this_entry_count) {
        set_error("perfect-hash partials must have the same entry count (%u != %u)", that_entry_counts[i],
                  this_entry_count);
        return HDK_HIP_ERR_INVALID_ARG;
      }
    }
    for (int t = 0; t < plan->num_targets; ++t) {
      HDK_REQUIRE(plan->targets[t].agg != HDK_AGG_SINGLE_VALUE || dev_error, "a SINGLE_VALUE target needs dev_error");
    }
    for (int base = 0; base < num_that; base += kMaxThat) {
      ThatList tl;
      tl.n = num_that - base < kMaxThat ? num_that - base : kMaxThat;
      for (int i = 0; i < kMaxThat; ++i) {
        tl.buf[i] = i < tl.n ? that_bufs[base + i] : nullptr;
      }
      hipLaunchKernelGGL(k_reduce_entrywise, dim3((n + kRedBlock - 1) / kRedBlock), dim3(kRedBlock), 0, s, d_plan,
                         this_buf, n, tl, init, dev_error);
    }
  } else {
    HDK_REQUIRE(dev_error, "dev_error is NULL");
    for (int i = 0; i < num_that; ++i) {
      if (that_entry_counts[i] == 0) continue;
      size_t blocks = (static_cast<size_t>(that_entry_counts[i]) + kRedBlock - 1) / kRedBlock;
      const size_t cap = static_cast<size_t>(props->num_cu) * 8;
      if (blocks > cap) blocks = cap;
      if (plan->key_width == 4) {
        hipLaunchKernelGGL(k_reduce_baseline<int32_t>, dim3(static_cast<unsigned>(blocks)), dim3(kRedBlock), 0, s,
                           d_plan, this_buf, this_entry_count, that_bufs[i], that_entry_counts[i], init, dev_error);
      } else {
        hipLaunchKernelGGL(k_reduce_baseline<int64_t>, dim3(static_cast<unsigned>(blocks)), dim3(kRedBlock), 0, s,
                           d_plan, this_buf, this_entry_count, that_bufs[i], that_entry_counts[i], init, dev_error);
      }
    }
  }
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_baseline_table_quads(const hdk_hip_plan* plan, uint32_t entry_count, int64_t* quads) {
  const int32_t st = validate_plan_layout(plan);
  if (st) return st;
  HDK_REQUIRE(quads, "quads is NULL");
  HDK_REQUIRE(plan->query_kind == HDK_Q_BASELINE_HASH, "not a baseline-hash plan");
  *quads = static_cast<int64_t>(table_quads(plan, entry_count));
  return HDK_HIP_OK;
}

static int32_t check_partition_args(const hdk_hip_plan* plan, const int64_t* buf, const int64_t* init_vals,
                                    int32_t num_owners) {
  const int32_t st = validate_plan_layout(plan);
  if (st) return st;
  HDK_REQUIRE(plan->query_kind == HDK_Q_BASELINE_HASH, "not a baseline-hash plan");
  HDK_REQUIRE(buf && init_vals, "NULL argument");
  HDK_REQUIRE(num_owners >= 1 && num_owners <= kMaxOwners, "num_owners must be in [1, 64]");
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_partition_baseline_count(const hdk_hip_plan* plan, const int64_t* buf, uint32_t entry_count,
                                                    const int64_t* init_vals, int32_t num_owners, uint32_t* counts,
                                                    int32_t device_id, void* stream) {
  int32_t st = check_partition_args(plan, buf, init_vals, num_owners);
  if (st) return st;
  HDK_REQUIRE(counts, "counts is NULL");
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  OwnerSegs segs = {};
  return run_partition<0>(plan, buf, entry_count, init_vals, num_owners, counts, segs, device_id, s);
}

extern "C" int32_t hdk_hip_partition_baseline(const hdk_hip_plan* plan, const int64_t* buf, uint32_t entry_count,
                                              const int64_t* init_vals, int32_t num_owners, const uint32_t* counts,
                                              int64_t* const* seg_bufs, int32_t device_id, void* stream) {
  int32_t st = check_partition_args(plan, buf, init_vals, num_owners);
  if (st) return st;
  HDK_REQUIRE(counts && seg_bufs, "NULL argument");
  hipStream_t s;
  st = device_enter(device_id, stream, &s);
  if (st) return st;
  OwnerSegs segs = {};
  for (int o = 0; o < num_owners; ++o) {
    HDK_REQUIRE(counts[o] == 0 || seg_bufs[o], "segment buffer is NULL");
    segs.buf[o] = seg_bufs[o];
    segs.count[o] = counts[o];
  }
  return run_partition<1>(plan, buf, entry_count, init_vals, num_owners, nullptr, segs, device_id, s);
}
