// scan_agg_partitioned.h -- open-addressing group-by through radix partitioning + LDS aggregation.
//
// Why: random global atomics run at ~2.4e10/s chip-wide whatever the table size
// (scripts/microbench/atomics.hip), so hdk_scan_agg_baseline_direct cannot pass ~2e10 rows/s.  The only
// way around the memory-side atomic units is locality: bring all rows of an entry range together and
// aggregate them in LDS.  Same result contract as the other baseline kernels (reference get_group_value +
// agg_*: QE/GroupByRuntime.cpp:31-200, QE/RuntimeFunctions.cpp:387-875), INCLUDING the placement: a group
// sits where the reference's probe sequence (key_hash % entry_count, then linearly on) finds it, so the
// table can be handed to anything that looks groups up -- a second launch into the same buffer,
// hdk_hip_reduce_buffers (ResultSetReduction::reduceOneEntryBaseline, QE/ResultSetReduction.cpp:694-731).
//
//   home(key)   = key_hash(key) % entry_count                      (the reference's first probe)
//   region f    = home / S: table entries [f * S, min((f + 1) * S, entry_count)),  S rows = 60 KiB of LDS
//   pass 1  filter rows, scatter (keys + argument columns -> tuples of <= 3 words) into P1 coarse slabs
//           (c = f >> p2_log2)
//   pass 2  scatter each coarse slab into its P2 = 2^p2_log2 fine slabs (one per region)
//   pass 3  one block per region: load its (initialised) image into LDS, insert/aggregate the slab's tuples
//           there, probing from home - f * S to the END of the region (no wrap), store the image back.  A
//           tuple whose probe runs off the region's end is set aside ("spilled")
//   pass 4  spilled tuples, and tuples that did not fit their slab (slabs are sized for a uniform hash with
//           slack, not counted), go through the ordinary whole-table find_or_claim with global atomics: the
//           probe walks over the full tail of the tuple's region into the next one, exactly as the reference's
//   skew    when even the overflow area fills up (a heavy-hitter key: its rows all land in one slab) the
//           scatter passes raise a device-side flag, the later passes return at once, and the launch's last
//           kernel -- hdk_scan_agg_baseline_direct, armed by that flag -- redoes the job with global atomics on
//           the still untouched table.  No host round trip.
//
// Scatter, per batch of kPartTile tuples: LDS histogram by bin -> one slab claim per bin -> LDS staging ordered
// by bin -> 16-byte copy-out of the runs as they come (a run is ~20 tuples: it starts and ends inside 128-byte lines
// that the neighbouring runs complete).  What the passes wait for is that copy-out (scripts/microbench/partition.hip):
// whole 128-byte-aligned lines 1.7 ms per 256 M tuples (4.7 TB/s of 32 B/tuple), unaligned runs with cursors shared
// by the whole chip 2.45 ms -- and 1.9 ms when a line is only ever written from ONE XCD, because the two halves of
// a line then merge in that XCD's L2 (the eight L2s are not coherent with each other; a line written from two of them
// reaches memory as two partial writes).  Hence:
//   * level 1: a coarse slab is kPartXcds sub-slabs with a cursor each, a block appends to the sub-slab of the XCD it
//     runs on (HW_REG_XCC_ID); pass 2 reads the eight sub-slabs as one tile sequence;
//   * level 2: all kPartG2X blocks of a coarse slab are launched on one XCD (block id % 8 selects the slab inside a
//     set of eight), so the fine slabs need no split.
//   (Placement is an affinity for speed; any block may write any sub-slab and the result is the same.)
// Slab cursors sit one per 128-byte line (128 cursors in 512 bytes serialise on four lines: 2.68 -> 2.34 ms).
// Two other ways to whole-line writes were measured and lost: padding every run to whole lines with tuples of another
// partition -- which the next pass drops for free because it recomputes every tuple's partition anyway (part_padding;
// still selectable with HDK_HIP_PART_G_LOG2=3 for measurements: 4.9 + 3.1 + 2.4 ms against 2.6 + 2.7 + 2.0 at the time)
// -- and carrying a bin's remainder into the block's next batch (3.4 / 2.45 ms: the extra registers and LDS cost the
// third block per CU).
//
// Bytes are the lever once the passes run at the copy rate, so the tuple is as small as the plan's column statistics
// allow (hdk_hip_col::has_stats -- the ChunkStats the reference's planner reads through getExpressionRange):
//   narrow   one key that fits 32 bits and ONE integer argument column whose values fit 32 bits: a tuple is ONE word
//            [argument as int32 : key as int32] -- 8 bytes instead of 16.  A nullable argument column gives up INT32_MIN
//            for its in-band NULL (so its non-NULL minimum must be above it); the readers hand the aggregate functions
//            the column's own 64-bit values and sentinel back.  The home is recomputed by each pass (the passes wait
//            for memory, not for the hash); a batch holds twice the tuples, so a run is as many BYTES as before.  A
//            value that contradicts the statistics raises the same device-side flag as skew: the atomics kernel
//            redoes the launch from the columns, the answer is right whatever the metadata said.
//
// Multi-GPU (no reference counterpart -- the reference merges per-device tables on the host,
// QE/Execute.cpp:1224-1336): with `owners` != 0 the keys are split by owner = mulhi32(key_hash, owners) and every
// owner holds an open-addressing table of its own (`entry_count` is then the OWNER's).  Level 1 scatters a rank's
// tuples into bins (owner, coarse slab of the owner's table), straight into the send buffer: one segment per owner,
// [header: tuples per (coarse slab, XCD) | the sub-slabs], every segment the same size whatever the data -- the
// exchange is one all-to-all with equal splits and no host round trip.  The owner runs level 2 over the sub-slabs
// of ALL ranks' segments (`nsrc` sources), then passes 3 and 4 as on one GPU.  A rank exchanges tuples, not partial
// tables: no local 200 M-entry table is built, scanned and re-inserted (measured in round 3: partition 530 ms +
// re-insert 4.5 ms against a 2.9 ms shard scan, profiles/r03_multi_gpu_floor_before.json).
#pragma once
#include "scan_agg_baseline_fast.h"
#include "watch.h"

namespace hdk {

constexpr int kPartBlock = 512;                  // scatter passes (sweep at C5: 256x8 3.7 ms, 512x4 2.9 ms, 1024x2 3.9 ms for pass 1)
constexpr int kPartAggBlock = 1024;              // aggregation pass: 2 blocks x 64 KiB LDS per CU, all 32 wave slots busy
#ifndef HDK_PART_NT_STORES
#define HDK_PART_NT_STORES 0  // measured: non-temporal copy-out and write-back stores are SLOWER (3.04 + 2.85 + 1.89 ms against 2.73 + 2.59 + 1.70): partial lines are merged in L2
#endif
constexpr bool kPartNtStores = HDK_PART_NT_STORES != 0;
constexpr int kPartVR = 4;                       // tuples per thread and scatter batch (16- and 24-byte tuples)
constexpr int kPartVRNarrow = 8;                 // ... 8-byte tuples: same bytes per batch, so a run stays ~128 bytes
constexpr int kPartTile = kPartBlock * kPartVR;  // tuples per scatter batch
constexpr int kPartTileNarrow = kPartBlock * kPartVRNarrow;
__host__ __device__ constexpr int part_tile(bool one_word) { return one_word ? kPartTileNarrow : kPartTile; }
constexpr int kPartMaxSrc = 16;                  // ranks a level-2 pass reads sub-slabs from (1 on a single GPU)
constexpr int kPartMaxBins = 256;                // bins a scatter pass distinguishes (P1 <= 256, P2 <= 256)
constexpr int kPartMaxArgs = 2;                  // argument columns carried in a tuple (at most)
constexpr int kPartMaxTW = 1 + kPartMaxArgs;     // tuple words: 1-2 keys + argument columns, 3 in all (LDS staging)
constexpr uint32_t kPartLdsBytes = 60 * 1024;    // LDS image of a region (two 1024-thread blocks per CU; above 64 KiB per block only one was resident)
constexpr uint32_t kPartCursorStride = 32;       // uint32 cursors one per 128-byte line
// pass-2 blocks per coarse slab.  Same box, C5 at 1 B rows: 8 -> 19.45 ms, 16 -> 18.3-18.7, 32 -> 18.2-18.3, 64 -> 18.0-18.2;
// at 256 M rows 16 -> 4.69 ms, 32 -> 4.55, 64 -> 4.50 (more, shorter blocks even out the tail of the pass)
constexpr int kPartG2X = 64;
constexpr int kPartXcds = 8;                     // L2 domains: a line of a slab should only ever be written from one of them
constexpr uint32_t kPartSpillSeg = 64;           // spilled tuples a region keeps in its own segment before it takes the shared list

struct PartArgs {
  const hdk_hip_plan* plan;
  KernParams kp;
  uint32_t entry_count;
  uint64_t total_rows;   // upper bound on the rows of the launch (hdk_hip_kernel_options::total_rows)
  uint32_t fine_count;   // PF = ceil(entry_count / S) regions
  uint32_t p2_log2;      // regions per coarse partition = 1 << p2_log2
  uint32_t p1;           // coarse partitions = ceil(PF / P2)
  uint32_t mod_magic, mod_shift;  // h % entry_count without a division (fastmod_u32)
  uint32_t slots;                 // S: entries per region (the LDS image of a region holds S rows)
  uint32_t reg_magic, reg_shift;  // home / S
  int64_t pad_key[2];             // padding keys (second key word 0), from two different coarse partitions
  uint32_t pad_coarse[2];
  int32_t tw;            // tuple words
  int32_t g_log2;        // flush granule G = 1 << g_log2 tuples (whole 128-byte lines)
  int32_t key_buf_idx, key_width, key_kind;
  int32_t nkeys;         // 1 or 2 key columns: tuple = [key0, (key1), arguments...]
  int32_t key2_buf_idx, key2_width, key2_kind;
  int32_t nargs;
  int32_t all_wide;      // key and argument columns are all 8 bytes wide (int64 / double): pass 1 reads full tiles with 16-byte loads
  BaseFastTarget arg[kPartMaxArgs];  // argument columns (buf_idx / width / kind); .target unused
  int32_t ntargets;
  int32_t tgt_index[HDK_HIP_MAX_TARGETS];  // plan target index
  int32_t tgt_arg[HDK_HIP_MAX_TARGETS];    // tuple word of its argument (>= nkeys), or 0 for none
  uint64_t cap1, cap2, cap_ovf, cap_spill; // capacities in tuples (cap1, cap2: multiples of G)
  uint64_t sub1;         // a coarse slab is kPartXcds sub-slabs of sub1 tuples, one per XCD that writes (cap1 = 8 * sub1)
  int64_t* slab1;        // [p1][cap1][tw]            (pass 3 reuses this memory as the shared spill list)
  int64_t* slab2;        // [fine_count][cap2][tw]
  int64_t* ovf;          // [cap_ovf][tw]
  uint32_t* fill1;       // [p1][kPartXcds] x kPartCursorStride
  uint32_t* fill2;       // [fine_count]
  uint32_t* fill_ovf;    // [1]
  uint32_t* fill_spill;  // [1]: the shared spill list (in slab1's memory), used once a region's own segment is full
  uint32_t* nspill;      // [fine_count]: tuples in the region's own spill segment
  int64_t* spill_seg;    // [fine_count][kPartSpillSeg][tw]
  uint32_t* fallback;    // [1]: set when the overflow area is exhausted -> the atomics kernel takes over
  int32_t init_output;   // HDK_HIP_LAUNCH_INIT_OUTPUT: the table holds nothing yet, pass 3 writes every region's image
  // hdk_part_aggregate_simple: rows of [key quad | one 8-byte integer slot], one aggregate
  int32_t simple_agg;          // hdk_hip_agg of that slot (SUM / MIN / MAX / COUNT), -1: the shape does not apply
  int32_t simple_has_arg;      // the aggregate reads tuple word 1
  int32_t simple_skip;         // *_skip_val: NULL arguments are skipped and the slot starts at its sentinel
  int32_t simple_arg_nullable;
  int64_t simple_null;         // the slot's sentinel
  int64_t simple_arg_null;     // in-band NULL of the argument column
  int32_t nquals;        // plain filters, applied in pass 1
  ProjFastQual q[kMaxPlainQuals];
  // narrow tuples: ONE word [argument as int32 : key as int32] (tw == 1 while the readers see [key, argument])
  int32_t soa;               // pass 3 keeps the region as [sums | keys | bits] (hdk_part_aggregate_soa); regions of kPartSoaSlots entries
  int32_t narrow;
  int32_t narrow_null;       // the argument column may hold NULLs: INT32_MIN stands for its in-band NULL
  int64_t narrow_arg_null;   // ... which the readers put back
  int32_t narrow_null_is_stale;  // statistics say "no NULLs" but the argument's type is nullable: a value equal to the in-band
                                 // NULL contradicts them (it fits 32 bits for a narrow column, so the range test cannot see it)
  int32_t narrow_pad_;
  // level-2 input: the sub-slabs of `nsrc` level-1 outputs (single GPU: this launch's own; owner: one per rank)
  uint32_t nsrc;
  uint32_t src_fill_stride;              // uint32 words between the cursors of neighbouring sub-slabs
  const int64_t* src_slab[kPartMaxSrc];  // [p1][kPartXcds][sub1][tw]
  const uint32_t* src_fill[kPartMaxSrc]; // [p1][kPartXcds] x src_fill_stride
  // multi-GPU tuple exchange: level 1 writes owner segments
  uint32_t owners;           // != 0: bins of level 1 are (owner, coarse slab of the owner's table)
  int8_t* send;              // [owners] segments of seg_bytes: [header | p1 x kPartXcds sub-slabs of sub1 tuples]
  uint64_t seg_bytes;
  uint64_t seg_header_bytes; // header: uint32 tuples[p1 * kPartXcds], then uint32 flag (0 = complete)
};

// h % d for any 32-bit h: unsigned division by an invariant divisor, round-up method in its branch-free form
// (33-bit magic number, low 32 bits in `magic`): one v_mul_hi_u32 + one v_mul_lo_u32 instead of a division
HDK_DEV uint32_t fastmod_u32(uint32_t h, uint32_t magic, uint32_t shift, uint32_t d) {
  const uint32_t t = __umulhi(magic, h);
  const uint32_t q = (((h - t) >> 1) + t) >> shift;
  return h - q * d;
}

// key_hash (QE/GroupByRuntime.cpp:24-29) of a tuple's key
template <typename K, int TW = kPartMaxTW>
HDK_DEV uint32_t part_hash_of_key(const PartArgs& a, const int64_t* tup) {
  const K k[2] = {static_cast<K>(tup[0]), TW > 1 ? static_cast<K>(tup[TW > 1 ? 1 : 0]) : K(0)};
  return (TW > 1 && a.nkeys == 2) ? key_hash_dev<K>(k, 2) : key_hash_dev<K>(k, 1);  // constant trip counts unroll
}
// the reference's first probe position of a tuple's key: key_hash % entry_count
template <typename K, int TW = kPartMaxTW>
HDK_DEV uint32_t part_home_of_key(const PartArgs& a, const int64_t* tup) {
  return fastmod_u32(part_hash_of_key<K, TW>(a, tup), a.mod_magic, a.mod_shift, a.entry_count);
}
// 4-byte keys leave the upper half of a tuple's first word free: pass 1 stores the home there, so that the hash
// (MurmurHash3 over the key: ~25 instructions, a third of them quarter-rate multiplies) and the modulo are worked
// out once per row instead of once per pass.  Every reader takes the key as static_cast<K>(word 0).  (Narrow tuples
// keep the argument there instead and recompute the home.)
template <typename K>
HDK_DEV int64_t part_pack_home(int64_t word0, uint32_t home) {
  return sizeof(K) == 4 ? static_cast<int64_t>((static_cast<uint64_t>(home) << 32) | static_cast<uint32_t>(word0)) : word0;
}
// NARROW: 1 / 0 known at compile time, -1 read from the arguments
template <typename K, int TW = kPartMaxTW, int NARROW = -1>
HDK_DEV uint32_t part_home(const PartArgs& a, const int64_t* tup) {
  const bool narrow = NARROW >= 0 ? NARROW != 0 : a.narrow != 0;
  if (sizeof(K) == 4 && !narrow) {
    return static_cast<uint32_t>(static_cast<uint64_t>(tup[0]) >> 32);
  }
  return part_home_of_key<K, TW>(a, tup);
}
// narrow tuples: the argument as the column holds it (int64, the column's own NULL sentinel)
HDK_DEV int64_t part_narrow_arg(const PartArgs& a, int64_t word) {
  const int32_t v = static_cast<int32_t>(static_cast<uint64_t>(word) >> 32);
  return (a.narrow_null && v == INT32_MIN) ? a.narrow_arg_null : static_cast<int64_t>(v);
}

// region (fine partition) of a home: home / S
HDK_DEV uint32_t part_region_of_home(const PartArgs& a, uint32_t home) {
  const uint32_t t = __umulhi(a.reg_magic, home);
  return (((home - t) >> 1) + t) >> a.reg_shift;
}
template <typename K, int TW = kPartMaxTW, int NARROW = -1>
HDK_DEV uint32_t part_region(const PartArgs& a, const int64_t* tup) {
  return part_region_of_home(a, part_home<K, TW, NARROW>(a, tup));
}

// The padding that fills a run up to whole 128-byte lines: a tuple whose KEY belongs to another coarse slab (and so
// to another region) than the slab it is written to.  Readers recompute the partition of every tuple anyway (pass
// 2: its bin; pass 3: its home) and drop what does not belong -- padding needs no flag, no count and no hole map.
// The host picks two keys from different coarse partitions; `region` is the first region of the slab written to.
HDK_DEV int64_t part_padding(const PartArgs& a, uint32_t region) {
  return (region >> a.p2_log2) != a.pad_coarse[0] ? a.pad_key[0] : a.pad_key[1];
}

// staging capacity of a scatter batch: the tuples plus at most G - 1 padding slots per bin (multiple of 8)
__host__ __device__ inline uint32_t part_stage_tuples(uint32_t nbins, uint32_t gmask, uint32_t tile) {
  return (tile + nbins * gmask + 7u) & ~7u;
}
__host__ inline size_t part_scatter_lds_bytes(uint32_t nbins, uint32_t gmask, int tw, bool narrow) {
  const size_t cs = part_stage_tuples(nbins, gmask, part_tile(narrow));
  return cs * tw * 8 + cs + 16;
}

// ---- scatter: LEVEL 1 reads the columns, LEVEL 2 reads a coarse slab ------------------------------------
// dynamic LDS: [cap_stage][tw] staging | uint8 bin of every staging slot [cap_stage]
// NARROW: one-word tuples [argument as int32 : key as int32] (K = int32_t, TW = 1), twice the tuples per batch
#ifndef HDK_PART_L1_WAVES
#define HDK_PART_L1_WAVES 0  // waves per SIMD the narrow level-1 kernel is held to (0: the compiler's choice)
#endif
template <int LEVEL, typename K, int TW, bool NARROW = false>
__global__ __launch_bounds__(kPartBlock, (NARROW && LEVEL == 1) ? HDK_PART_L1_WAVES : 0) void hdk_part_scatter(PartArgs a) {
  __shared__ uint32_t s_cnt[kPartMaxBins];     // tuples of the bin in this batch; rank source
  // per bin and batch, read as one 16-byte word by the copy-out: .x start of the run in the staging area, .y slots
  // of the run (tuples rounded up to whole lines), .z first slab position claimed for it, .w tuples | kind << 16:
  // kind 0 slab, 2 slab + overflow area, 3 slab + dropped (fallback armed)
  __shared__ uint4 s_run[kPartMaxBins];
  __shared__ uint32_t s_nfit[kPartMaxBins];    // kinds 2, 3: how many slots still fit the slab
  __shared__ uint32_t s_obase[kPartMaxBins];   // kind 2: overflow-area position of the tuples that do not
  __shared__ uint64_t s_binbase[kPartMaxBins]; // byte offset of the bin's slab (this block's sub-slab of it) from `out`
  __shared__ uint32_t s_sub_before[kPartMaxSrc * kPartXcds + 1];  // level 2: tiles in front of every sub-slab
  __shared__ uint32_t s_sub_n[kPartMaxSrc * kPartXcds];
  __shared__ uint32_t s_total, s_stop;
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  constexpr int VR = NARROW ? kPartVRNarrow : kPartVR;
  constexpr int kTile = kPartBlock * VR;
  constexpr int tw = TW;  // tuple words, compile time: the tuples of a batch live in registers
  static_assert(!NARROW || (TW == 1 && sizeof(K) == 4), "narrow tuples: one word, 4-byte key");
  const int tid = threadIdx.x;
  const bool dist = LEVEL == 1 && a.owners != 0;
  const uint32_t nbins = LEVEL == 1 ? (dist ? a.owners * a.p1 : a.p1) : (1u << a.p2_log2);
  const uint32_t gmask = (1u << a.g_log2) - 1;
  const uint32_t cap_stage = part_stage_tuples(nbins, gmask, kTile);
  int64_t* s_stage = s_dyn;
  uint8_t* s_binof = reinterpret_cast<uint8_t*>(s_dyn + static_cast<size_t>(cap_stage) * tw);
  // Partial lines at the ends of a run are completed by whoever claims the neighbouring run; the two halves merge in
  // L2 only if both writers sit behind the SAME L2 (scripts/microbench/partition.hip: one cursor and sub-slab per
  // (bin, XCD) 1.9 ms per 256 M tuples, shared cursors 2.45).  Level 1: a coarse slab is eight sub-slabs, a block
  // appends to the one of its XCD (block b runs on XCD b % 8 -- affinity for speed only, any mapping is correct).
  // Level 2: all blocks of a coarse slab are launched on one XCD (see the id -> (c, member) mapping below), so the
  // fine slabs need no split.
  // (level 1 asks the hardware which XCD it runs on -- HW_REG_XCC_ID, bits 3:0; level 2 needs the launch-order mapping)
  const uint32_t xcd = LEVEL == 1 ? (static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & (kPartXcds - 1))
                                  : (blockIdx.x & (kPartXcds - 1));
  const uint32_t l2_c = LEVEL == 2 ? (blockIdx.x / (kPartXcds * kPartG2X)) * kPartXcds + xcd : 0;  // coarse slab of this block
  const uint32_t l2_m = LEVEL == 2 ? (blockIdx.x / kPartXcds) % kPartG2X : 0;                        // its member number
  if (LEVEL == 2 && l2_c >= a.p1) {
    return;
  }
  const uint64_t cap = LEVEL == 1 ? a.sub1 : a.cap2;  // what a cursor may hand out
  const uint32_t cstride = LEVEL == 1 ? kPartCursorStride * kPartXcds : 1;
  const uint32_t bin0 = LEVEL == 1 ? 0 : (l2_c << a.p2_log2);  // first region of the coarse slab
  uint32_t* fill = (LEVEL == 1 ? a.fill1 + static_cast<size_t>(xcd) * kPartCursorStride : a.fill2) + static_cast<size_t>(bin0) * cstride;
  int8_t* out = LEVEL == 1 ? (dist ? a.send : reinterpret_cast<int8_t*>(a.slab1)) : reinterpret_cast<int8_t*>(a.slab2);
  for (int i = tid; i < kPartMaxBins; i += kPartBlock) {
    s_cnt[i] = 0;
    // where the bin's tuples go: level 1 -- sub-slab `xcd` of coarse slab i (in owner i / p1's segment of the send
    // buffer when the launch scatters to owners); level 2 -- fine slab bin0 + i
    uint64_t off;
    if (LEVEL == 2) {
      off = static_cast<uint64_t>(bin0 + i) * a.cap2 * tw * 8;
    } else if (dist) {
      const uint32_t o = static_cast<uint32_t>(i) / a.p1, c = static_cast<uint32_t>(i) - o * a.p1;
      off = static_cast<uint64_t>(o) * a.seg_bytes + a.seg_header_bytes + (static_cast<uint64_t>(c) * kPartXcds + xcd) * a.sub1 * tw * 8;
    } else {
      off = (static_cast<uint64_t>(i) * kPartXcds + xcd) * a.sub1 * tw * 8;
    }
    s_binbase[i] = off;
  }
  __syncthreads();

  auto do_batch = [&](const bool (&live)[VR], int64_t (&tup)[VR][TW], const uint32_t (&bin)[VR]) {
    // 1. histogram + rank inside the bin
    uint32_t rank[VR];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      rank[r] = 0;
      if (live[r]) {
        rank[r] = atomicAdd(&s_cnt[bin[r]], 1u);
      }
    }
    __syncthreads();
    // 2. one slab claim per bin, in whole 128-byte lines; run starts in the staging area (exclusive scan by wave 0)
    uint32_t my_n = 0, my_np = 0;
    if (tid < kPartMaxBins) {
      const uint32_t n = tid < static_cast<int>(nbins) ? s_cnt[tid] : 0;
      const uint32_t np = (n + gmask) & ~gmask;  // the run's slots: what is not a tuple is padding (see part_padding)
      uint32_t kind = 0, base = 0, nfit = np, obase = 0;
      if (n) {
        base = atomicAdd(fill + static_cast<size_t>(tid) * cstride, np);
        // what does not fit the slab any more (heavy hitter) goes to the overflow area, without padding -- split at
        // the slab's end, a multiple of G like `base`: readers take the slab as [0, min(cursor, cap)), every slot written
        nfit = static_cast<uint64_t>(base) >= cap ? 0u : static_cast<uint32_t>(min(static_cast<uint64_t>(np), cap - base));
        if (nfit < n) {
          if (dist) {
            kind = 3;  // (no overflow area travels with an owner segment: the exchange is flagged incomplete)
            atomicMax(a.fallback, 1u);
          } else {
            obase = atomicAdd(a.fill_ovf, n - nfit);
            kind = 2;
            if (static_cast<uint64_t>(obase) + (n - nfit) > a.cap_ovf) {
              kind = 3;
              atomicMax(a.fallback, 1u);  // too skewed for slabs: hand the launch to the atomics kernel (2 = interrupted sticks)
            }
          }
        }
      }
      s_run[tid].y = np;
      s_run[tid].z = base;
      s_run[tid].w = n | (kind << 16);
      s_nfit[tid] = nfit;
      s_obase[tid] = obase;
      my_n = n;
      my_np = np;
    }
    if (tid < kWave) {
      uint32_t carry = 0;
      for (int c0 = 0; c0 < kPartMaxBins; c0 += kWave) {
        const uint32_t n = (s_cnt[c0 + tid] + gmask) & ~gmask;  // padded length of the run
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const uint32_t v = __shfl_up(incl, d, kWave);
          if (tid >= d) {
            incl += v;
          }
        }
        s_run[c0 + tid].x = carry + incl - n;
        carry += __shfl(incl, kWave - 1, kWave);
      }
      if (tid == 0) {
        s_total = carry;
      }
    }
    __syncthreads();
    // 3. stage the tuples ordered by bin; the padding slots only get their bin
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      if (live[r]) {
        const uint32_t si = s_run[bin[r]].x + rank[r];
        s_binof[si] = static_cast<uint8_t>(bin[r]);
        if (TW == 2) {
          bf_i64x2 v;
          v.x = tup[r][0];
          v.y = tup[r][TW - 1];
          reinterpret_cast<bf_i64x2*>(s_stage)[si] = v;
        } else {
#pragma unroll
          for (int w = 0; w < TW; ++w) {
            s_stage[static_cast<size_t>(si) * tw + w] = tup[r][w];
          }
        }
      }
    }
    if (tid < kPartMaxBins) {
      const uint32_t lp = s_run[tid].x;
      for (uint32_t j = my_n; j < my_np; ++j) {
        s_binof[lp + j] = static_cast<uint8_t>(tid);
      }
      s_cnt[tid] = 0;
    }
    __syncthreads();
    // 4. copy out whole lines: consecutive staging slots of a bin go to consecutive slab positions
    const uint32_t total = s_total;
    for (uint32_t i = tid; i < total; i += kPartBlock) {
      const uint32_t b = s_binof[i];
      const uint4 run = s_run[b];
      const uint32_t r = i - run.x;
      const uint32_t n = run.w & 0xffffu, kind = run.w >> 16;
      const bool pad = r >= n;
      uint32_t nfit = run.y;
      if (kind >= 2) {
        nfit = s_nfit[b];
        if (r >= nfit && (pad || kind == 3)) {
          continue;
        }
      }
      int64_t* q = r >= nfit ? a.ovf + (static_cast<size_t>(s_obase[b]) + (r - nfit)) * tw
                             : reinterpret_cast<int64_t*>(out + s_binbase[b]) + static_cast<size_t>(run.z + r) * tw;
      int64_t t[TW];
      if (TW == 2) {
        const bf_i64x2 v = reinterpret_cast<const bf_i64x2*>(s_stage)[i];
        t[0] = pad ? 0 : v.x;
        t[TW - 1] = pad ? 0 : v.y;
      } else {
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          t[w] = pad ? 0 : s_stage[static_cast<size_t>(i) * tw + w];
        }
      }
      if (pad) {
        t[0] = part_padding(a, LEVEL == 1 ? b << a.p2_log2 : bin0 + b);
      }
      if (TW == 2) {
        bf_i64x2 v;
        v.x = t[0];
        v.y = t[1];
        if (kPartNtStores) {
          __builtin_nontemporal_store(v, reinterpret_cast<bf_i64x2*>(q));
        } else {
          *reinterpret_cast<bf_i64x2*>(q) = v;
        }
      } else {
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          q[w] = t[w];
        }
      }
    }
    __syncthreads();
  };

  if (LEVEL == 2) {
    // Block-uniform, through LDS: another block of THIS kernel may raise the flag (a fine slab and the overflow area
    // full) between the reads of two waves.  A wave that returns on its own takes its share of the batch's steps with
    // it -- wave 0 gone means no run starts and no total: the others then copy out with whatever LDS held (found by the
    // round-3 soak as a 1-in-5 memory fault on a zipf-skewed two-key shape).
    if (tid == 0) {
      s_stop = __hip_atomic_load(a.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_stop) {
      return;
    }
  }
  if (LEVEL == 1) {
    const uint64_t nfrag = *a.kp.num_fragments;
    const uint32_t ntab = *a.kp.num_tables;
    const Watch watch = watch_begin(a.kp);
    // level-1 bin of a row from its key hash; 4-byte keys of wide tuples get the home packed into word 0
    auto bin_of = [&](int64_t (&t)[TW]) -> uint32_t {
      const uint32_t h = part_hash_of_key<K, TW>(a, t);
      const uint32_t home = fastmod_u32(h, a.mod_magic, a.mod_shift, a.entry_count);
      if (sizeof(K) == 4 && !NARROW) {
        t[0] = part_pack_home<K>(t[0], home);
      }
      const uint32_t c = part_region_of_home(a, home) >> a.p2_log2;
      return dist ? static_cast<uint32_t>((static_cast<uint64_t>(h) * a.owners) >> 32) * a.p1 + c : c;
    };
    // narrow tuples: [argument as int32 : key as int32]; a value the statistics did not announce hands the launch to
    // the atomics kernel (which reads the columns at their full width)
    bool stale = false;
    auto pack_narrow = [&](int64_t key, int64_t arg) -> int64_t {
      const bool is_null = a.narrow_null && arg == a.narrow_arg_null;
      const int32_t v32 = is_null ? INT32_MIN : static_cast<int32_t>(arg);
      stale |= static_cast<int64_t>(static_cast<int32_t>(key)) != key ||
               (!is_null && (static_cast<int64_t>(v32) != arg || (a.narrow_null && v32 == INT32_MIN))) ||
               (a.narrow_null_is_stale && arg == a.narrow_arg_null);
      return static_cast<int64_t>((static_cast<uint64_t>(static_cast<uint32_t>(v32)) << 32) | static_cast<uint32_t>(key));
    };
    int64_t tile = blockIdx.x;
    int64_t frag_tile_begin = 0;
    for (uint64_t f = 0; f < nfrag; ++f) {
      const int64_t nrows = a.kp.num_rows[f * ntab];
      const int64_t ntiles = (nrows + kTile - 1) / kTile;
      const int8_t* const* cols = a.kp.col_buffers[f];
      for (; tile < frag_tile_begin + ntiles; tile += gridDim.x) {
        // block-uniform exit: thread 0 samples the flag, everyone agrees before the batch's barriers
        if (tid == 0) {
          uint32_t stop = __hip_atomic_load(a.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (watch.flags) {
            if (const int32_t w_ = watch_poll(watch)) {  // interrupt / watchdog: the launch ends with that error
              record_error(a.kp.error_code, w_);
              atomicMax(a.fallback, 2u);  // the later passes return at once; the armed atomics kernel runs only for 1
              stop = 2;
            }
          }
          s_stop = stop;
        }
        __syncthreads();
        if (s_stop) {
          return;
        }
        const int64_t row0 = (tile - frag_tile_begin) * kTile + tid;
        bool live[VR];
        int64_t tup[VR][TW];
        uint32_t bin[VR];
        if (a.all_wide && !a.nquals && (tile - frag_tile_begin + 1) * kTile <= nrows) {
          // full tile, every tuple column 8 bytes wide, no filter: rows dealt in adjacent pairs, one 16-byte
          // non-temporal load per lane and pair, no bounds tests (which rows a lane takes does not matter to a scatter)
          const uint64_t tile_byte0 = static_cast<uint64_t>(tile - frag_tile_begin) * kTile * 8;
          constexpr int NCOL = NARROW ? 2 : TW;  // columns read: a narrow tuple packs key and argument into one word
          int64_t colv[VR][NCOL];
#pragma unroll
          for (int w = 0; w < NCOL; ++w) {
            const int bi = w == 0 ? a.key_buf_idx : (w < a.nkeys ? a.key2_buf_idx : a.arg[w - a.nkeys > 0 ? 1 : 0].buf_idx);
            const uint64_t b = reinterpret_cast<uintptr_t>(cols[bi]) + tile_byte0;
            const uint32_t b_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b));
            const uint32_t b_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b >> 32));
            const __attribute__((address_space(1))) int8_t* base =
                reinterpret_cast<const __attribute__((address_space(1))) int8_t*>((static_cast<uint64_t>(b_hi) << 32) | b_lo);
#pragma unroll
            for (int u = 0; u < VR / 2; ++u) {
              const bf_i64x2 v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(
                  base + static_cast<uint32_t>(u * kPartBlock + tid) * 16u));
              colv[2 * u][w] = v.x;
              colv[2 * u + 1][w] = v.y;
            }
          }
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            live[r] = true;
            if (NARROW) {
              tup[r][0] = pack_narrow(colv[r][0], colv[r][NCOL - 1]);
            } else {
#pragma unroll
              for (int w = 0; w < TW; ++w) {
                tup[r][w] = colv[r][w < NCOL ? w : 0];
              }
            }
            bin[r] = bin_of(tup[r]);
          }
          do_batch(live, tup, bin);
          continue;
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          live[r] = row0 + static_cast<int64_t>(r) * kPartBlock < nrows;
        }
        if (a.nquals) {
          int64_t rows[VR];
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            rows[r] = row0 + static_cast<int64_t>(r) * kPartBlock;
          }
          plain_quals_pass<VR>(a.q, a.nquals, cols, rows, live, true);
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
          tup[r][0] = live[r] ? decode_col_g(cols[a.key_buf_idx], a.key_width, a.key_kind, row, true) : 0;
        }
        if (NARROW) {
          const BaseFastTarget c = a.arg[0];
#pragma unroll
          for (int r = 0; r < VR; ++r) {
            const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
            const int64_t v = live[r] ? decode_col_g(cols[c.buf_idx], c.width, c.kind, row, true) : 0;
            tup[r][0] = pack_narrow(tup[r][0], v);
          }
        }
        const int nk = a.nkeys;
#pragma unroll
        for (int w = 1; w < TW; ++w) {
          if (w < nk) {  // second key column
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
              tup[r][w] = live[r] ? decode_col_g(cols[a.key2_buf_idx], a.key2_width, a.key2_kind, row, true) : 0;
            }
          } else {
            const BaseFastTarget c = a.arg[w - nk > 0 ? 1 : 0];
#pragma unroll
            for (int r = 0; r < VR; ++r) {
              const int64_t row = row0 + static_cast<int64_t>(r) * kPartBlock;
              tup[r][w] = live[r] ? decode_col_g(cols[c.buf_idx], c.width, c.kind, row, true) : 0;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          bin[r] = bin_of(tup[r]);
        }
        do_batch(live, tup, bin);
      }
      frag_tile_begin += ntiles;
    }
    if (NARROW && __any(stale) && (threadIdx.x & (kWave - 1)) == 0) {
      atomicMax(a.fallback, 1u);
    }
  } else {
    // coarse slab c = the sub-slabs of every source (one per source and XCD), each [0, min(cursor, sub1)); a slot
    // that does not belong to c is padding.  The block walks their tiles as one sequence, member m taking tiles
    // m, m + G, ...
    const uint32_t c = l2_c;
    const uint32_t nsub = a.nsrc * kPartXcds;
    for (uint32_t j = tid; j < nsub; j += kPartBlock) {
      const uint32_t src = j / kPartXcds, x = j % kPartXcds;
      s_sub_n[j] = static_cast<uint32_t>(min(static_cast<uint64_t>(a.src_fill[src][(static_cast<size_t>(c) * kPartXcds + x) * a.src_fill_stride]), a.sub1));
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t before = 0;
      for (uint32_t j = 0; j < nsub; ++j) {
        s_sub_before[j] = before;
        before += (s_sub_n[j] + kTile - 1) / kTile;
      }
      s_sub_before[nsub] = before;
    }
    __syncthreads();
    const uint32_t ntiles = s_sub_before[nsub];
    for (uint32_t vt = l2_m; vt < ntiles; vt += kPartG2X) {
      uint32_t sj = 0;
      for (uint32_t j = 1; j < nsub; ++j) {  // (uniform: scalar loop over at most 127 LDS words)
        sj += vt >= s_sub_before[j] ? 1u : 0u;
      }
      sj = __builtin_amdgcn_readfirstlane(sj);
      const uint64_t n = s_sub_n[sj];
      const uint64_t t0 = static_cast<uint64_t>(vt - s_sub_before[sj]) * kTile;
      const int64_t* in = a.src_slab[sj / kPartXcds] + (static_cast<size_t>(c) * kPartXcds + (sj % kPartXcds)) * a.sub1 * tw;
      bool live[VR];
      int64_t tup[VR][TW];
      uint32_t bin[VR];
      if (TW == 2 && t0 + kTile <= n) {  // full tile of 16-byte tuples: unconditional loads
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const uint64_t i = t0 + static_cast<uint64_t>(r) * kPartBlock + tid;
          const bf_i64x2 v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(
              reinterpret_cast<uintptr_t>(in + i * 2)));
          tup[r][0] = v.x;
          tup[r][TW - 1] = v.y;
          const uint32_t f = part_region<K, TW, NARROW>(a, tup[r]);
          live[r] = (f >> a.p2_log2) == c;
          bin[r] = f & (nbins - 1);
        }
        do_batch(live, tup, bin);
        continue;
      }
      if (TW == 1 && t0 + kTile <= n) {  // full tile of 8-byte tuples: adjacent pairs, one 16-byte load per pair
#pragma unroll
        for (int u = 0; u < VR / 2; ++u) {
          const uint64_t i = t0 + (static_cast<uint64_t>(u) * kPartBlock + tid) * 2;
          const bf_i64x2 v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(
              reinterpret_cast<uintptr_t>(in + i)));
          tup[2 * u][0] = v.x;
          tup[2 * u + 1][0] = v.y;
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
          const uint32_t f = part_region<K, TW, NARROW>(a, tup[r]);
          live[r] = (f >> a.p2_log2) == c;
          bin[r] = f & (nbins - 1);
        }
        do_batch(live, tup, bin);
        continue;
      }
#pragma unroll
      for (int r = 0; r < VR; ++r) {
        const uint64_t i = t0 + static_cast<uint64_t>(r) * kPartBlock + tid;
        live[r] = i < n;
        if (TW == 2) {
          bf_i64x2 v;
          v.x = 0;
          v.y = 0;
          if (live[r]) {
            v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(
                reinterpret_cast<uintptr_t>(in + i * 2)));
          }
          tup[r][0] = v.x;
          tup[r][TW - 1] = v.y;
        } else {
#pragma unroll
          for (int w = 0; w < TW; ++w) {
            tup[r][w] = live[r] ? __builtin_nontemporal_load(in + i * tw + w) : 0;
          }
        }
        const uint32_t f = part_region<K, TW, NARROW>(a, tup[r]);
        live[r] = live[r] && (f >> a.p2_log2) == c;
        bin[r] = f & (nbins - 1);
      }
      do_batch(live, tup, bin);
    }
  }
}

// level 1 of a scatter to owners: the header of every owner segment -- tuples per (coarse slab, XCD) sub-slab, then the
// flag word (0: complete; else the exchange cannot be used: a sub-slab overflowed, the statistics were stale, or the
// launch was interrupted)
// what a segment says about its own shape: the owner lays the inbox out with ITS shape, and column statistics (which
// decide the tuple width) are per rank
HDK_DEV uint32_t part_segment_tag(const PartArgs& a) { return static_cast<uint32_t>(a.tw * 8) | (a.p1 << 8); }

__global__ void hdk_part_publish(PartArgs a) {
  const uint32_t o = blockIdx.x;
  uint32_t* header = reinterpret_cast<uint32_t*>(a.send + static_cast<uint64_t>(o) * a.seg_bytes);
  const uint32_t nsub = a.p1 * kPartXcds;
  for (uint32_t j = threadIdx.x; j < nsub; j += blockDim.x) {
    const uint32_t n = a.fill1[(static_cast<size_t>(o) * nsub + j) * kPartCursorStride];
    header[j] = static_cast<uint32_t>(min(static_cast<uint64_t>(n), a.sub1));
  }
  if (threadIdx.x == 0) {
    header[nsub] = *a.fallback;
    header[nsub + 1] = part_segment_tag(a);
  }
}

// owner side: any source whose segment is flagged makes the whole exchange unusable -- the owner's passes return at
// once (fallback != 0) and the step ends with HDK_HIP_ERR_EXCHANGE_INCOMPLETE, which the caller answers with the
// table exchange (partial tables, hdk_hip_partition_baseline) or a retry
__global__ void hdk_part_collect_flags(PartArgs a) {
  const uint32_t nsub = a.p1 * kPartXcds;
  uint32_t bad = 0;
  for (uint32_t r = threadIdx.x; r < a.nsrc; r += blockDim.x) {
    bad |= a.src_fill[r][nsub];
    bad |= a.src_fill[r][nsub + 1] != part_segment_tag(a) ? 1u : 0u;  // a sender with another tuple width / geometry
  }
  if (bad) {
    atomicMax(a.fallback, 3u);
    record_error(a.kp.error_code, HDK_HIP_ERR_EXCHANGE_INCOMPLETE);
  }
}

// What a tuple needs to know about a target, gathered once per block into LDS: reading the plan (global
// memory) per tuple put ~25 dependent global loads on every tuple's path (10.7 ms for the C5 shape).
struct PartTarget {
  int32_t agg, has_arg, skip_null, arg_is_fp;
  int32_t slot_width, slot2_width, slot_off, slot2_off;
  int32_t arg_word, arg_fp, arg_nullable, pad_;
  int64_t null_val;      // skip value of the slot
  int64_t arg_null_val;  // in-band NULL of the argument
};

HDK_DEV void part_load_targets(const PartArgs& a, const hdk_hip_plan* p, PartTarget* s_tg) {
  if (static_cast<int>(threadIdx.x) < a.ntargets) {
    const int t = threadIdx.x;
    const hdk_hip_target& tg = p->targets[a.tgt_index[t]];
    PartTarget d;
    d.agg = tg.agg;
    d.has_arg = tg.has_arg;
    d.skip_null = tg.skip_null;
    d.arg_is_fp = tg.arg_is_fp;
    d.slot_width = tg.slot_width;
    d.slot2_width = tg.slot2_width;
    d.slot_off = tg.slot_off;
    d.slot2_off = tg.slot2_off;
    d.arg_word = a.tgt_arg[t];
    const int32_t kind = d.arg_word ? a.arg[d.arg_word - a.nkeys].kind : HDK_COL_INT;
    d.arg_fp = kind == HDK_COL_FLOAT || kind == HDK_COL_DOUBLE;
    d.arg_nullable = tg.arg.nullable;
    d.pad_ = 0;
    d.null_val = tg.null_val;
    d.arg_null_val = tg.arg.null_val;
    s_tg[t] = d;
  }
}

// the aggregates of one tuple onto its group's row (`rowb`: in an LDS image or in the table itself)
HDK_DEV void part_apply_targets(const PartTarget* s_tg, int ntargets, int8_t* rowb, const int64_t* tup) {
  for (int t = 0; t < ntargets; ++t) {
    const PartTarget tg = s_tg[t];
    int8_t* s1 = rowb + tg.slot_off;
    int8_t* s2 = rowb + tg.slot2_off;
    int64_t v = tg.arg_word == 1 ? tup[1] : (tg.arg_word == 2 ? tup[2] : 0);
    bool is_null = false;
    if (tg.has_arg) {  // eval_target_arg for a plain column argument (device_common.h)
      const bool arg_fp = tg.arg_fp != 0;
      if (tg.skip_null && is_null_val(v, tg.arg_null_val, tg.arg_nullable, arg_fp)) {
        is_null = true;
      } else {
        if (tg.arg_is_fp && !arg_fp) {
          v = double_to_bits(static_cast<double>(v));
        }
        if (tg.skip_null) {
          is_null = tg.arg_is_fp ? (bits_to_double(v) == bits_to_double(tg.null_val)) : (v == tg.null_val);
        }
      }
    }
    if (is_null) {
      continue;
    }
    if (tg.agg == HDK_AGG_COUNT) {
      g_count(s1, tg.slot_width);
      continue;
    }
    if (tg.agg == HDK_AGG_AVG) {
      g_count(s2, tg.slot2_width);
    }
    if (tg.slot_width == 4) {
      g_agg32(tg.agg, tg.skip_null, static_cast<int32_t>(tg.null_val), reinterpret_cast<int32_t*>(s1), static_cast<int32_t>(v));
    } else {
      g_agg64(tg.agg, tg.arg_is_fp, tg.skip_null, tg.null_val, reinterpret_cast<int64_t*>(s1), v);
    }
  }
}

// ---- pass 3: one block per region -------------------------------------------------------------------------
// (second launch bound = waves per SIMD: two 1024-thread blocks per CU need 8, i.e. <= 64 VGPRs and <= 80 SGPRs --
// at 98 SGPRs only one block was resident and the pass took 3.2 ms instead of 2.3)
template <typename K>
__global__ __launch_bounds__(kPartAggBlock, 8) void hdk_part_aggregate(PartArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds_table[];
  __shared__ PartTarget s_tg[HDK_HIP_MAX_TARGETS];
  __shared__ uint32_t s_nspill;
  const hdk_hip_plan* __restrict__ p = a.plan;
  const int tid = threadIdx.x;
  const uint32_t f = blockIdx.x;
  if (tid == 0) {
    s_nspill = 0;
  }
  const TableShape shape = table_shape(p);
  const uint32_t rq = shape.row_quads;
  const int ntargets = a.ntargets;
  const uint32_t first = f * a.slots;                        // first table entry of the region
  const uint32_t slots = min(a.slots, a.entry_count - first);  // (the last region may be short)
  part_load_targets(a, p, s_tg);
  const uint32_t words = slots * rq;
  int64_t* region = a.kp.groupby_buf[0] + static_cast<size_t>(first) * rq;
  const uint64_t n = min(static_cast<uint64_t>(a.fill2[f]), a.cap2);
  const bool idle = n == 0 || *a.fallback;  // nothing to aggregate here (or the atomics kernel will redo the launch)
  if (a.init_output) {
    // the table is uninitialised memory: build the region's image here -- the bytes the init kernel writes -- and
    // store it whatever happens next (an idle region still has to read as empty; so does the fallback's table)
    const uint32_t keys_quads = (static_cast<uint32_t>(p->key_count) * static_cast<uint32_t>(p->key_width) + 7) / 8;
    for (uint32_t i = tid; i < words; i += kPartAggBlock) {
      const int64_t v = init_row_quad(i % rq, keys_quads, p->key_count, p->key_width, a.kp.init_agg_vals);
      lds_table[i] = v;
      if (idle) {
        region[i] = v;
      }
    }
    if (idle) {
      return;
    }
  } else {
    if (idle) {
      return;  // the region keeps its initialised (empty) image
    }
    for (uint32_t i = tid; i < words; i += kPartAggBlock) {
      lds_table[i] = region[i];  // the init kernel's image (or an earlier launch's groups), whatever the layout
    }
  }
  __syncthreads();
  const int tw = a.tw;
  const int64_t* in = a.slab2 + static_cast<size_t>(f) * a.cap2 * tw;
  int64_t* spill = a.slab1;  // dead since pass 2 and large enough for every tuple of the launch
  // Two tuples per trip, the next pair's loads issued before the current pair is applied: the LDS
  // claim/aggregate chain (ds_* ops, lgkmcnt) of one pair hides the HBM latency (vmcnt) of the next.
  // Plain scalars on purpose -- no per-thread tuple arrays that could end up in scratch.
  int64_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
  auto fetch = [&](uint64_t i, int64_t& t0, int64_t& t1, int64_t& t2) {
    if (i < n) {
      const int64_t* q = in + i * tw;
      t0 = __builtin_nontemporal_load(q);
      t1 = tw > 1 ? __builtin_nontemporal_load(q + 1) : 0;
      t2 = tw > 2 ? __builtin_nontemporal_load(q + 2) : 0;
    }
  };
  auto apply = [&](const int64_t (&raw)[kPartMaxTW]) {
    // (a narrow tuple is one word; the aggregate functions see [key, argument] as the columns hold them)
    const int64_t tup[kPartMaxTW] = {raw[0], a.narrow ? part_narrow_arg(a, raw[0]) : raw[1], raw[2]};
    const K key[2] = {static_cast<K>(tup[0]), static_cast<K>(tup[1])};  // (word 1 is only read as a key when key_count == 2)
    const uint32_t local = part_home<K>(a, tup) - first;
    if (local >= slots) {
      return;  // padding of the scatter passes: a key of another region (part_padding)
    }
    bool fresh;
    const int64_t e = find_or_claim_from<K, false>(shape, lds_table, slots, local, key, &fresh);
    if (e < 0) {  // the group lives past the end of this region: pass 4 places it with the whole-table probe
      // (a counter in LDS and a segment of the region's own: one shared cursor for every region took a returning
      // global atomic on ONE address per spilled tuple -- ~1e5 of them per 256 M rows, ~1 ms)
      const uint32_t k = atomicAdd(&s_nspill, 1u);
      int64_t* q;
      if (k < kPartSpillSeg) {
        q = a.spill_seg + (static_cast<size_t>(f) * kPartSpillSeg + k) * tw;
      } else {
        const uint32_t o = atomicAdd(a.fill_spill, 1u);
        if (o >= a.cap_spill) {
          // one GPU: cannot happen, the shared list is as large as the launch's input.  The owner of a tuple exchange
          // has a list of its own, sized for the spill of a table that is not full: loud, not silent
          record_error(a.kp.error_code, HDK_HIP_ERR_EXCHANGE_INCOMPLETE);
          return;
        }
        q = spill + static_cast<size_t>(o) * tw;
      }
      for (int w = 0; w < tw; ++w) {
        q[w] = tup[w];
      }
      return;
    }
    part_apply_targets(s_tg, ntargets, reinterpret_cast<int8_t*>(lds_table + static_cast<size_t>(e) * rq), tup);
  };
  fetch(tid, a0, a1, a2);
  fetch(static_cast<uint64_t>(tid) + kPartAggBlock, b0, b1, b2);
  for (uint64_t i = tid; i < n; i += 2 * kPartAggBlock) {
    const int64_t ta[kPartMaxTW] = {a0, a1, a2};
    const int64_t tb[kPartMaxTW] = {b0, b1, b2};
    const bool has_b = i + kPartAggBlock < n;
    fetch(i + 2 * kPartAggBlock, a0, a1, a2);
    fetch(i + 3 * kPartAggBlock, b0, b1, b2);
    apply(ta);
    if (has_b) {
      apply(tb);
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < words; i += kPartAggBlock) {
    region[i] = lds_table[i];
  }
  if (tid == 0) {
    a.nspill[f] = min(s_nspill, kPartSpillSeg);
  }
}

// ---- pass 3 for the common shape: one key, one 8-byte integer aggregate (C5: GROUP BY key, SUM(val)) ---------------
// Same contract as hdk_part_aggregate -- image, bounded probe, spill segment -- with everything the general kernel
// looks up per tuple (table shape, target descriptors, aggregate kind) fixed or block-uniform: the general kernel
// spends ~210 vector + ~220 scalar instructions per tuple, most of them on its per-target interpreter.
// TW_T / AGG_T / SKIP_T >= 0 fix the tuple width, the aggregate and its NULL handling at compile time (C5: two words,
// SUM); -1 reads them from the arguments
// NARROW: one-word tuples [argument as int32 : key as int32], read in adjacent pairs (16 bytes per lane and load)
template <typename K, int TW_T = -1, int AGG_T = -1, int SKIP_T = -1, bool NARROW = false>
__global__ __launch_bounds__(kPartAggBlock, 8) void hdk_part_aggregate_simple(PartArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds_table[];
  __shared__ uint32_t s_nspill;
  const hdk_hip_plan* __restrict__ p = a.plan;
  const int tid = threadIdx.x;
  const uint32_t f = blockIdx.x;
  if (tid == 0) {
    s_nspill = 0;
  }
  constexpr uint32_t rq = 2;
  const uint32_t first = f * a.slots;
  const uint32_t slots = min(a.slots, a.entry_count - first);
  int64_t* region = a.kp.groupby_buf[0] + static_cast<size_t>(first) * rq;
  const uint32_t n = static_cast<uint32_t>(min(static_cast<uint64_t>(a.fill2[f]), a.cap2));  // (cursors are 32-bit)
  const bool idle = n == 0 || *a.fallback;
  bf_i64x2* lds_rows = reinterpret_cast<bf_i64x2*>(lds_table);
  // (first * 16 bytes: aligned; the table is hipMalloc'ed global memory -- as a generic pointer every access is a flat_*)
  __attribute__((address_space(1))) bf_i64x2* region_rows =
      reinterpret_cast<__attribute__((address_space(1))) bf_i64x2*>(reinterpret_cast<uintptr_t>(region));
  if (a.init_output) {
    bf_i64x2 row;
    row.x = init_row_quad(0, 1, 1, p->key_width, a.kp.init_agg_vals);
    row.y = a.kp.init_agg_vals[0];
    for (uint32_t i = tid; i < slots; i += kPartAggBlock) {
      lds_rows[i] = row;
      if (idle) {
        region_rows[i] = row;
      }
    }
    if (idle) {
      return;
    }
  } else {
    if (idle) {
      return;
    }
    for (uint32_t i = tid; i < slots; i += kPartAggBlock) {
      lds_rows[i] = region_rows[i];
    }
  }
  __syncthreads();
  const int agg = AGG_T >= 0 ? AGG_T : a.simple_agg;
  const bool has_arg = a.simple_has_arg != 0;
  const bool skip = SKIP_T >= 0 ? SKIP_T != 0 : a.simple_skip != 0;
  const bool arg_nullable = a.simple_arg_nullable != 0;
  const int64_t slot_null = a.simple_null;
  const int64_t arg_null = a.simple_arg_null;
  const int tw = TW_T > 0 ? TW_T : a.tw;
  const int64_t* in = a.slab2 + static_cast<size_t>(f) * a.cap2 * tw;
  const K ek = empty_key<K>();
  auto apply = [&](int64_t t0, int64_t t1) {
    const int64_t tup[2] = {t0, t1};
    const uint32_t local = part_home<K, 2, NARROW ? 1 : 0>(a, tup) - first;
    if (local >= slots) {
      return;  // padding of the scatter passes
    }
    const K key = static_cast<K>(t0);
    uint32_t pos = local;
    int64_t seen = 0;  // the slot as it was when the row was looked at
    for (;;) {
      K* kp = reinterpret_cast<K*>(lds_rows + pos);
      // Look before claiming (a slot never returns to EMPTY), key and slot in ONE 16-byte LDS read.  The two halves may
      // be from different moments: harmless -- a key that reads EMPTY goes through the CAS, a slot that reads as the
      // sentinel goes through the CAS of g_agg64_seen, and neither ever returns to that state.  (A `volatile` read here
      // lost the LDS address space and compiled to a system-coherent flat_load + full waitcnt per probe step.)
      const bf_i64x2 row = lds_rows[pos];
      K old = static_cast<K>(row.x);
      seen = row.y;
      if (old == ek) {
        if constexpr (sizeof(K) == 8) {
          old = static_cast<K>(atomicCAS(reinterpret_cast<unsigned long long*>(kp), static_cast<unsigned long long>(ek),
                                         static_cast<unsigned long long>(key)));
        } else {
          old = static_cast<K>(atomicCAS(reinterpret_cast<unsigned int*>(kp), static_cast<unsigned int>(ek),
                                         static_cast<unsigned int>(key)));
        }
        if (old == ek) {
          break;  // claimed
        }
      }
      if (old == key) {
        break;
      }
      if (++pos == slots) {
        break;
      }
    }
    if (pos == slots) {  // the group lives past the end of this region: pass 4 places it (kept out of the probe loop)
      const uint32_t k = atomicAdd(&s_nspill, 1u);
      int64_t* q;
      if (k < kPartSpillSeg) {
        q = a.spill_seg + (static_cast<size_t>(f) * kPartSpillSeg + k) * tw;
      } else {
        const uint32_t o = atomicAdd(a.fill_spill, 1u);
        if (o >= a.cap_spill) {
          record_error(a.kp.error_code, HDK_HIP_ERR_EXCHANGE_INCOMPLETE);  // (see hdk_part_aggregate)
          return;
        }
        q = a.slab1 + static_cast<size_t>(o) * tw;
      }
      q[0] = t0;
      if (tw > 1) {
        q[1] = t1;
      }
      return;
    }
    int64_t* slot = reinterpret_cast<int64_t*>(lds_rows + pos) + 1;
    // NULL argument, or a value that collides with the skip value (`val != skip_val`): as part_apply_targets
    if (has_arg && skip && ((arg_nullable && t1 == arg_null) || t1 == slot_null)) {
      return;
    }
    if (agg == HDK_AGG_COUNT) {
      atomicAdd(reinterpret_cast<unsigned long long*>(slot), 1ull);
    } else {
      g_agg64_seen(agg, false, skip, slot_null, slot, t1, seen);
    }
  };
  // two tuples per trip, the next pair's loads issued before the current pair is applied (see hdk_part_aggregate)
  int64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
  const __attribute__((address_space(1))) int8_t* in_bytes =
      reinterpret_cast<const __attribute__((address_space(1))) int8_t*>(reinterpret_cast<uintptr_t>(in));
  if (NARROW) {
    // lane-adjacent pairs: tuples 2j and 2j + 1 come in with one 16-byte load (cap2 is even, so every slab starts
    // 16-byte aligned); (a0, b0) hold the pair being applied, (a1, b1) the pair in flight
    const uint32_t npairs = (n + 1) / 2;
    auto fetch_pair = [&](uint32_t j, int64_t& lo, int64_t& hi) {
      if (j < npairs) {
        const bf_i64x2 v = __builtin_nontemporal_load(
            reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(in_bytes + static_cast<uint64_t>(j) * 16));
        lo = v.x;
        hi = v.y;
      }
    };
    fetch_pair(tid, a1, b1);
    for (uint32_t j = tid; j < npairs; j += kPartAggBlock) {
      a0 = a1;
      b0 = b1;
      fetch_pair(j + kPartAggBlock, a1, b1);
      apply(a0, part_narrow_arg(a, a0));
      if (2 * j + 1 < n) {
        apply(b0, part_narrow_arg(a, b0));
      }
    }
  } else {
    auto fetch = [&](uint32_t i, int64_t& t0, int64_t& t1) {  // 32-bit indices: cap2 <= 0xFFFF0000 (match_partitioned)
      if (i < n) {
        if (tw == 2) {
          const bf_i64x2 v = __builtin_nontemporal_load(
              reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(in_bytes + static_cast<uint64_t>(i) * 16));
          t0 = v.x;
          t1 = v.y;
        } else {
          t0 = __builtin_nontemporal_load(in + i);
          t1 = 0;
        }
      }
    };
    fetch(tid, a0, a1);
    fetch(static_cast<uint32_t>(tid) + kPartAggBlock, b0, b1);
    // (i + 3 * kPartAggBlock cannot wrap: n <= cap2 <= 0xFFFF0000)
    for (uint32_t i = tid; i < n; i += 2 * kPartAggBlock) {
      const int64_t ta0 = a0, ta1 = a1, tb0 = b0, tb1 = b1;
      const bool has_b = i + kPartAggBlock < n;
      fetch(i + 2 * kPartAggBlock, a0, a1);
      fetch(i + 3 * kPartAggBlock, b0, b1);
      apply(ta0, ta1);
      if (has_b) {
        apply(tb0, tb1);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < slots; i += kPartAggBlock) {
    if (kPartNtStores) {
      __builtin_nontemporal_store(lds_rows[i], region_rows + i);
    } else {
      region_rows[i] = lds_rows[i];
    }
  }
  if (tid == 0) {
    a.nspill[f] = min(s_nspill, kPartSpillSeg);
  }
}

// ---- pass 3, structure-of-arrays image: GROUP BY one 4-byte table key, SUM over one integer column ------------------
// What bounds hdk_part_aggregate_simple once the tuples are 8 bytes is the LDS, not memory (3.8 ms per 1 B tuples for
// 11 GB): a probe reads key AND slot as one ds_read_b128 at a random address -- four banks per lane, ~3 conflicts per
// access -- to learn whether the slot still holds its NULL sentinel.  Here the image is [sums int64 | keys int32 |
// "saw a non-NULL value" bits]: a probe is a 4-byte read, an update one non-returning ds_add_u64 (plus one ds_or_b32
// on the bit when NULL arguments are possible), the sentinel rule of agg_sum_skip_val (QE/RuntimeFunctions.cpp:
// 612-625: the first non-NULL value replaces the sentinel, later ones add) is applied once per entry when the image
// is written back: slot = saw a value ? sum : sentinel.  12 bytes per entry instead of 16 also make a region a third
// larger (kPartSoaSlots entries), i.e. a quarter fewer regions for the scatter passes to separate.
// NULLS = false: no tuple carries a NULL argument (narrow tuples whose column statistics say so, checked in pass 1)
// and the table is fresh, so "claimed" implies "saw a value" and the bits are not kept.
constexpr uint32_t kPartSoaSlots = 5056;  // 5056 x 12 B + 5056 / 8 B = 61 304 B of LDS (two blocks per CU)
__host__ __device__ constexpr uint32_t part_soa_lds_bytes(uint32_t slots) { return slots * 12u + ((slots + 31u) / 32u) * 4u; }

template <bool NARROW, bool NULLS>
__global__ __launch_bounds__(kPartAggBlock, 8) void hdk_part_aggregate_soa(PartArgs a) {
  extern __shared__ __attribute__((aligned(16))) int64_t lds_table[];
  __shared__ uint32_t s_nspill;
  using K = int32_t;
  const int tid = threadIdx.x;
  const uint32_t f = blockIdx.x;
  if (tid == 0) {
    s_nspill = 0;
  }
  const uint32_t first = f * a.slots;
  const uint32_t slots = min(a.slots, a.entry_count - first);
  int64_t* s_sum = lds_table;
  K* s_key = reinterpret_cast<K*>(lds_table + a.slots);
  uint32_t* s_mask = reinterpret_cast<uint32_t*>(s_key + a.slots);
  __attribute__((address_space(1))) bf_i64x2* region_rows = reinterpret_cast<__attribute__((address_space(1))) bf_i64x2*>(
      reinterpret_cast<uintptr_t>(a.kp.groupby_buf[0] + static_cast<size_t>(first) * 2));
  const uint32_t n = static_cast<uint32_t>(min(static_cast<uint64_t>(a.fill2[f]), a.cap2));
  const bool idle = n == 0 || *a.fallback;
  const K ek = empty_key<K>();
  const bool skip = a.simple_skip != 0;
  const int64_t slot_null = a.simple_null;
  const int64_t slot_init = a.kp.init_agg_vals[0];
  if (a.init_output) {
    if (idle) {  // nothing to aggregate here: the region still has to read as empty
      bf_i64x2 row;
      row.x = static_cast<int64_t>(static_cast<uint32_t>(ek));
      row.y = slot_init;
      for (uint32_t i = tid; i < slots; i += kPartAggBlock) {
        region_rows[i] = row;
      }
      return;
    }
    for (uint32_t i = tid; i < slots; i += kPartAggBlock) {
      s_sum[i] = 0;
      s_key[i] = ek;
    }
    if (NULLS) {
      for (uint32_t i = tid; i < (slots + 31) / 32; i += kPartAggBlock) {
        s_mask[i] = 0;
      }
    }
  } else {
    if (idle) {
      return;
    }
    // an earlier launch's groups: a slot at its sentinel has seen no value yet
    for (uint32_t i = tid; i < (slots + 31) / 32; i += kPartAggBlock) {
      s_mask[i] = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < slots; i += kPartAggBlock) {
      const bf_i64x2 row = region_rows[i];
      const K k = static_cast<K>(row.x);
      const bool has = k != ek && !(skip && row.y == slot_null);
      s_key[i] = k;
      s_sum[i] = k == ek ? 0 : (has ? row.y : 0) - (skip ? 0 : slot_init);  // (without the sentinel rule the slot is init + sum)
      if (has) {
        atomicOr(&s_mask[i >> 5], 1u << (i & 31));
      }
    }
  }
  __syncthreads();
  const bool arg_nullable = a.simple_arg_nullable != 0;
  const int64_t arg_null = a.simple_arg_null;
  const int tw = NARROW ? 1 : 2;
  const int64_t* in = a.slab2 + static_cast<size_t>(f) * a.cap2 * tw;
  auto apply = [&](int64_t t0, int64_t t1, bool is_null) {
    const int64_t tup[2] = {t0, t1};
    const uint32_t local = part_home<K, 2, NARROW ? 1 : 0>(a, tup) - first;
    if (local >= slots) {
      return;  // padding of the scatter passes
    }
    const K key = static_cast<K>(t0);
    uint32_t pos = local;
    for (;;) {
      K old = s_key[pos];  // look before claiming (a key never returns to EMPTY)
      if (old == ek) {
        old = static_cast<K>(atomicCAS(reinterpret_cast<unsigned int*>(s_key + pos), static_cast<unsigned int>(ek),
                                       static_cast<unsigned int>(key)));
        if (old == ek) {
          break;  // claimed
        }
      }
      if (old == key) {
        break;
      }
      if (++pos == slots) {
        break;
      }
    }
    if (pos == slots) {  // the group lives past the end of this region: pass 4 places it (kept out of the probe loop)
      const uint32_t k = atomicAdd(&s_nspill, 1u);
      int64_t* q;
      if (k < kPartSpillSeg) {
        q = a.spill_seg + (static_cast<size_t>(f) * kPartSpillSeg + k) * tw;
      } else {
        const uint32_t o = atomicAdd(a.fill_spill, 1u);
        if (o >= a.cap_spill) {
          record_error(a.kp.error_code, HDK_HIP_ERR_EXCHANGE_INCOMPLETE);  // (see hdk_part_aggregate)
          return;
        }
        q = a.slab1 + static_cast<size_t>(o) * tw;
      }
      q[0] = t0;
      if (tw > 1) {
        q[1] = t1;
      }
      return;
    }
    if (is_null) {
      return;
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(s_sum + pos), static_cast<unsigned long long>(t1));
    if (NULLS) {
      atomicOr(&s_mask[pos >> 5], 1u << (pos & 31));
    }
  };
  const __attribute__((address_space(1))) int8_t* in_bytes =
      reinterpret_cast<const __attribute__((address_space(1))) int8_t*>(reinterpret_cast<uintptr_t>(in));
  if (NARROW) {
    // lane-adjacent pairs of one-word tuples, one 16-byte load per pair, the next pair in flight while this one is applied
    const uint32_t npairs = (n + 1) / 2;
    int64_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
    auto fetch_pair = [&](uint32_t j, int64_t& lo, int64_t& hi) {
      if (j < npairs) {
        const bf_i64x2 v = __builtin_nontemporal_load(
            reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(in_bytes + static_cast<uint64_t>(j) * 16));
        lo = v.x;
        hi = v.y;
      }
    };
    auto apply_word = [&](int64_t w) {
      const int32_t v32 = static_cast<int32_t>(static_cast<uint64_t>(w) >> 32);
      apply(w, static_cast<int64_t>(v32), NULLS && a.narrow_null && v32 == INT32_MIN);
    };
    fetch_pair(tid, a1, b1);
    for (uint32_t j = tid; j < npairs; j += kPartAggBlock) {
      a0 = a1;
      b0 = b1;
      fetch_pair(j + kPartAggBlock, a1, b1);
      apply_word(a0);
      if (2 * j + 1 < n) {
        apply_word(b0);
      }
    }
  } else {
    int64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    auto fetch = [&](uint32_t i, int64_t& t0, int64_t& t1) {
      if (i < n) {
        const bf_i64x2 v = __builtin_nontemporal_load(
            reinterpret_cast<const __attribute__((address_space(1))) bf_i64x2*>(in_bytes + static_cast<uint64_t>(i) * 16));
        t0 = v.x;
        t1 = v.y;
      }
    };
    auto null_arg = [&](int64_t t1) { return skip && ((arg_nullable && t1 == arg_null) || t1 == slot_null); };
    fetch(tid, a0, a1);
    fetch(static_cast<uint32_t>(tid) + kPartAggBlock, b0, b1);
    for (uint32_t i = tid; i < n; i += 2 * kPartAggBlock) {
      const int64_t ta0 = a0, ta1 = a1, tb0 = b0, tb1 = b1;
      const bool has_b = i + kPartAggBlock < n;
      fetch(i + 2 * kPartAggBlock, a0, a1);
      fetch(i + 3 * kPartAggBlock, b0, b1);
      apply(ta0, ta1, null_arg(ta1));
      if (has_b) {
        apply(tb0, tb1, null_arg(tb1));
      }
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < slots; i += kPartAggBlock) {
    const K k = s_key[i];
    bf_i64x2 row;
    row.x = static_cast<int64_t>(static_cast<uint32_t>(k));  // [key | zero padding], as the init kernel leaves the quad
    if (k == ek) {
      row.y = slot_init;
    } else if (skip) {
      const bool saw = NULLS ? ((s_mask[i >> 5] >> (i & 31)) & 1u) != 0 : true;
      row.y = saw ? s_sum[i] : slot_null;
    } else {
      row.y = slot_init + s_sum[i];
    }
    region_rows[i] = row;
  }
  if (tid == 0) {
    a.nspill[f] = min(s_nspill, kPartSpillSeg);
  }
}

// one tuple straight onto the table in global memory with the reference's probe sequence (passes 4 and the owner's
// atomics kernel); false: the table is full
template <typename K>
HDK_DEV bool part_apply_global(const PartArgs& a, const TableShape& shape, int64_t* table, const PartTarget* s_tg, const int64_t* q) {
  const int tw = a.tw;
  int64_t tup[kPartMaxTW];
#pragma unroll
  for (int w = 0; w < kPartMaxTW; ++w) {
    tup[w] = w < tw ? q[w] : 0;
  }
  if (a.narrow) {
    tup[1] = part_narrow_arg(a, tup[0]);
  }
  const K key[2] = {static_cast<K>(tup[0]), static_cast<K>(tup[1])};
  bool fresh;
  const int64_t e = find_or_claim_from<K, true>(shape, table, a.entry_count, part_home<K>(a, tup), key, &fresh);
  if (e < 0) {
    return false;  // (the reference's get_group_value returns NULL)
  }
  part_apply_targets(s_tg, a.ntargets, reinterpret_cast<int8_t*>(table + static_cast<size_t>(e) * shape.row_quads), tup);
  return true;
}

// ---- pass 4: overflow and spilled tuples, straight onto the table with the reference's probe sequence --------
template <typename K>
__global__ __launch_bounds__(kPartBlock) void hdk_part_overflow(PartArgs a) {
  __shared__ PartTarget s_tg[HDK_HIP_MAX_TARGETS];
  const hdk_hip_plan* __restrict__ p = a.plan;
  if (*a.fallback) {
    return;
  }
  part_load_targets(a, p, s_tg);
  __syncthreads();
  const uint64_t n_ovf = min(static_cast<uint64_t>(*a.fill_ovf), a.cap_ovf);
  const uint64_t n_spill = min(static_cast<uint64_t>(*a.fill_spill), a.cap_spill);
  const uint64_t n_seg = static_cast<uint64_t>(a.fine_count) * kPartSpillSeg;  // slots of the regions' own segments
  const TableShape shape = table_shape(p);
  const int tw = a.tw;
  int64_t* table = a.kp.groupby_buf[0];
  int32_t err = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kPartBlock + threadIdx.x; i < n_ovf + n_spill + n_seg;
       i += static_cast<uint64_t>(gridDim.x) * kPartBlock) {
    const int64_t* q;
    if (i < n_seg) {
      if (static_cast<uint32_t>(i % kPartSpillSeg) >= a.nspill[i / kPartSpillSeg]) {
        continue;
      }
      q = a.spill_seg + i * tw;
    } else {
      q = i - n_seg < n_ovf ? a.ovf + (i - n_seg) * tw : a.slab1 + (i - n_seg - n_ovf) * tw;
    }
    if (!part_apply_global<K>(a, shape, table, s_tg, q)) {
      err = HDK_HIP_ERR_OUT_OF_SLOTS;  // the table is full
    }
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

// ---- the owner of a tuple exchange whose inbox is too skewed for fine slabs (a heavy hitter: every rank's tuples of
// the key meet in one slab and its overflow area): every tuple of every source segment straight onto the owner's
// table, which pass 3 has left empty.  Armed like the one-GPU path's atomics kernel: it runs only when level 2 has
// set *fallback to 1 (3: a SENDER could not carry its share -- the exchange is incomplete; 2: interrupted).
template <typename K>
__global__ __launch_bounds__(kPartBlock) void hdk_part_owner_fallback(PartArgs a) {
  __shared__ PartTarget s_tg[HDK_HIP_MAX_TARGETS];
  const hdk_hip_plan* __restrict__ p = a.plan;
  if (*a.fallback != 1) {
    return;
  }
  part_load_targets(a, p, s_tg);
  __syncthreads();
  const TableShape shape = table_shape(p);
  const int tw = a.tw;
  int64_t* table = a.kp.groupby_buf[0];
  const uint32_t nsub = a.p1 * kPartXcds;
  int32_t err = 0;
  for (uint32_t sj = blockIdx.x; sj < a.nsrc * nsub; sj += gridDim.x) {
    const uint32_t src = sj / nsub, j = sj - src * nsub;
    const uint64_t n = min(static_cast<uint64_t>(a.src_fill[src][static_cast<size_t>(j) * a.src_fill_stride]), a.sub1);
    const int64_t* in = a.src_slab[src] + static_cast<size_t>(j) * a.sub1 * tw;
    for (uint64_t i = threadIdx.x; i < n; i += kPartBlock) {
      if (!part_apply_global<K>(a, shape, table, s_tg, in + i * tw)) {
        err = HDK_HIP_ERR_OUT_OF_SLOTS;
      }
    }
  }
  if (err) {
    record_error(a.kp.error_code, err);
  }
}

}  // namespace hdk
