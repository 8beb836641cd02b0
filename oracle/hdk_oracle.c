/*
 * hdk_oracle.c -- CPU restatement of HDK's per-row runtime for the hot path (see hdk_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into or called by the product library.
 * Parity: PINNED against oracle/_ref (the reference's RuntimeFunctions.cpp compiled in place) and
 * the golden vectors under tests/golden/.
 *
 * Plain C11, scalar, row-at-a-time, in the reference's own order of operations.
 */
#define _POSIX_C_SOURCE 200809L /* clock_gettime for the timed baseline loop */
#define _GNU_SOURCE /* sched_setaffinity, CPU_* */
#include "hdk_oracle.h"

#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sched.h>
#include <sys/mman.h>
#include <stdio.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================================================
 * Hashes
 * ========================================================================================== */

static inline uint32_t rotl32(uint32_t x, int8_t r) {
  return (x << r) | (x >> (32 - r));
}

/* QE/MurmurHash3Inl.h:11-76 (MurmurHash3_x86_32). */
uint32_t orc_murmur_hash3(const void* key, int len, uint32_t seed) {
  const uint8_t* data = (const uint8_t*)key;
  const int nblocks = len / 4;
  uint32_t h1 = seed;
  const uint32_t c1 = 0xcc9e2d51;
  const uint32_t c2 = 0x1b873593;
  for (int i = 0; i < nblocks; i++) {
    uint32_t k1;
    memcpy(&k1, data + 4 * i, 4);
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3:
      k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2:
      k1 ^= (uint32_t)tail[1] << 8; /* fallthrough */
    case 1:
      k1 ^= tail[0];
      k1 *= c1;
      k1 = rotl32(k1, 15);
      k1 *= c2;
      h1 ^= k1;
  }
  h1 ^= (uint32_t)len;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6b;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35;
  h1 ^= h1 >> 16;
  return h1;
}

/* QE/MurmurHash1Inl.h:6-52. */
uint32_t orc_murmur_hash1(const void* key, int len, uint32_t seed) {
  const uint32_t m = 0xc6a4a793;
  const int r = 16;
  uint32_t h = seed ^ ((uint32_t)len * m);
  const unsigned char* data = (const unsigned char*)key;
  while (len >= 4) {
    uint32_t k;
    memcpy(&k, data, 4);
    h += k;
    h *= m;
    h ^= h >> 16;
    data += 4;
    len -= 4;
  }
  switch (len) {
    case 3:
      h += (uint32_t)data[2] << 16; /* fallthrough */
    case 2:
      h += (uint32_t)data[1] << 8; /* fallthrough */
    case 1:
      h += data[0];
      h *= m;
      h ^= h >> r;
  }
  h *= m;
  h ^= h >> 10;
  h *= m;
  h ^= h >> 17;
  return h;
}

/* QE/MurmurHash1Inl.h:54-104 (MurmurHash64A). */
uint64_t orc_murmur_hash64a(const void* key, int len, uint64_t seed) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  uint64_t h = seed ^ ((uint64_t)len * m);
  const uint8_t* p = (const uint8_t*)key;
  const uint8_t* end = p + (len / 8) * 8;
  while (p != end) {
    uint64_t k;
    memcpy(&k, p, 8);
    p += 8;
    k *= m;
    k ^= k >> r;
    k *= m;
    h ^= k;
    h *= m;
  }
  switch (len & 7) {
    case 7:
      h ^= (uint64_t)p[6] << 48; /* fallthrough */
    case 6:
      h ^= (uint64_t)p[5] << 40; /* fallthrough */
    case 5:
      h ^= (uint64_t)p[4] << 32; /* fallthrough */
    case 4:
      h ^= (uint64_t)p[3] << 24; /* fallthrough */
    case 3:
      h ^= (uint64_t)p[2] << 16; /* fallthrough */
    case 2:
      h ^= (uint64_t)p[1] << 8; /* fallthrough */
    case 1:
      h ^= (uint64_t)p[0];
      h *= m;
  }
  h ^= h >> r;
  h *= m;
  h ^= h >> r;
  return h;
}

/* QE/GroupByRuntime.cpp:24-29. */
uint32_t orc_key_hash(const int64_t* key, uint32_t key_count, uint32_t key_byte_width) {
  return orc_murmur_hash3(key, (int)(key_byte_width * key_count), 0);
}

/* ============================================================================================
 * Decoders -- QE/DecodersImpl.h:30-150
 * ========================================================================================== */

int64_t orc_fixed_width_int_decode(const int8_t* byte_stream, int32_t byte_width, int64_t pos) {
  switch (byte_width) {
    case 1:
      return (int64_t)byte_stream[pos * byte_width];
    case 2: {
      int16_t v;
      memcpy(&v, &byte_stream[pos * byte_width], 2);
      return v;
    }
    case 4: {
      int32_t v;
      memcpy(&v, &byte_stream[pos * byte_width], 4);
      return v;
    }
    case 8: {
      int64_t v;
      memcpy(&v, &byte_stream[pos * byte_width], 8);
      return v;
    }
    default:
      return INT64_MIN + 1;
  }
}

int64_t orc_fixed_width_unsigned_decode(const int8_t* byte_stream, int32_t byte_width, int64_t pos) {
  switch (byte_width) {
    case 1:
      return ((const uint8_t*)byte_stream)[pos * byte_width];
    case 2: {
      uint16_t v;
      memcpy(&v, &byte_stream[pos * byte_width], 2);
      return v;
    }
    case 4: {
      uint32_t v;
      memcpy(&v, &byte_stream[pos * byte_width], 4);
      return v;
    }
    case 8: {
      uint64_t v;
      memcpy(&v, &byte_stream[pos * byte_width], 8);
      return (int64_t)v;
    }
    default:
      return INT64_MIN + 1;
  }
}

/* fixed_width_small_date_decode: QE/DecodersImpl.h:151-159 */
int64_t orc_fixed_width_small_date_decode(const int8_t* byte_stream, int32_t byte_width, int32_t null_val,
                                          int64_t ret_null_val, int64_t pos) {
  const int64_t val = orc_fixed_width_int_decode(byte_stream, byte_width, pos);
  return val == null_val ? ret_null_val : val * 86400;
}

float orc_fixed_width_float_decode(const int8_t* byte_stream, int64_t pos) {
  float v;
  memcpy(&v, &byte_stream[pos * sizeof(float)], sizeof(float));
  return v;
}

double orc_fixed_width_double_decode(const int8_t* byte_stream, int64_t pos) {
  double v;
  memcpy(&v, &byte_stream[pos * sizeof(double)], sizeof(double));
  return v;
}

/* ============================================================================================
 * Group lookup
 * ========================================================================================== */

static inline int8_t* align_to_int64_ptr(int8_t* p) { /* Shared/BufferCompaction.h:44-47 */
  uintptr_t a = (uintptr_t)p;
  a += sizeof(int64_t) - 1;
  return (int8_t*)((a >> 3) << 3);
}
static inline size_t align_to_int64_sz(size_t a) {
  return ((a + 7) >> 3) << 3;
}

/* QE/RuntimeFunctions.cpp:1209-1253 (template + key_width switch). */
int64_t* orc_get_matching_group_value(int64_t* groups_buffer, uint32_t h, const int64_t* key,
                                      uint32_t key_count, uint32_t key_width,
                                      uint32_t row_size_quad) {
  uint64_t off = (uint64_t)h * row_size_quad;
  if (key_width == 4) {
    int32_t* row_ptr = (int32_t*)(groups_buffer + off);
    const int32_t* key32 = (const int32_t*)key;
    if (*row_ptr == HDK_EMPTY_KEY_32) {
      memcpy(row_ptr, key32, key_count * sizeof(int32_t));
      return (int64_t*)align_to_int64_ptr((int8_t*)(row_ptr + key_count));
    }
    if (memcmp(row_ptr, key32, key_count * sizeof(int32_t)) == 0) {
      return (int64_t*)align_to_int64_ptr((int8_t*)(row_ptr + key_count));
    }
    return NULL;
  } else if (key_width == 8) {
    int64_t* row_ptr = groups_buffer + off;
    if (*row_ptr == HDK_EMPTY_KEY_64) {
      memcpy(row_ptr, key, key_count * sizeof(int64_t));
      return row_ptr + key_count;
    }
    if (memcmp(row_ptr, key, key_count * sizeof(int64_t)) == 0) {
      return row_ptr + key_count;
    }
    return NULL;
  }
  return NULL;
}

/* QE/GroupByRuntime.cpp:31-55. */
int64_t* orc_get_group_value(int64_t* groups_buffer, uint32_t groups_buffer_entry_count,
                             const int64_t* key, uint32_t key_count, uint32_t key_width,
                             uint32_t row_size_quad) {
  uint32_t h = orc_key_hash(key, key_count, key_width) % groups_buffer_entry_count;
  int64_t* matching_group =
      orc_get_matching_group_value(groups_buffer, h, key, key_count, key_width, row_size_quad);
  if (matching_group) {
    return matching_group;
  }
  uint32_t h_probe = (h + 1) % groups_buffer_entry_count;
  while (h_probe != h) {
    matching_group = orc_get_matching_group_value(
        groups_buffer, h_probe, key, key_count, key_width, row_size_quad);
    if (matching_group) {
      return matching_group;
    }
    h_probe = (h_probe + 1) % groups_buffer_entry_count;
  }
  return NULL;
}

/* QE/RuntimeFunctions.cpp:1255-1305. */
int32_t orc_get_matching_group_value_columnar_slot(int64_t* groups_buffer, uint32_t entry_count,
                                                   uint32_t h, const int64_t* key, uint32_t key_count,
                                                   uint32_t key_width) {
  if (key_width == 4) {
    int32_t* key_buffer = (int32_t*)groups_buffer;
    const int32_t* key32 = (const int32_t*)key;
    uint32_t off = h;
    if (key_buffer[off] == HDK_EMPTY_KEY_32) {
      for (size_t i = 0; i < key_count; ++i) {
        key_buffer[off] = key32[i];
        off += entry_count;
      }
      return (int32_t)h;
    }
    off = h;
    for (size_t i = 0; i < key_count; ++i) {
      if (key_buffer[off] != key32[i]) {
        return -1;
      }
      off += entry_count;
    }
    return (int32_t)h;
  } else if (key_width == 8) {
    uint32_t off = h;
    if (groups_buffer[off] == HDK_EMPTY_KEY_64) {
      for (size_t i = 0; i < key_count; ++i) {
        groups_buffer[off] = key[i];
        off += entry_count;
      }
      return (int32_t)h;
    }
    off = h;
    for (size_t i = 0; i < key_count; ++i) {
      if (groups_buffer[off] != key[i]) {
        return -1;
      }
      off += entry_count;
    }
    return (int32_t)h;
  }
  return -1;
}

/* QE/GroupByRuntime.cpp:90-113. */
int32_t orc_get_group_value_columnar_slot(int64_t* groups_buffer, uint32_t groups_buffer_entry_count,
                                          const int64_t* key, uint32_t key_count,
                                          uint32_t key_width) {
  uint32_t h = orc_key_hash(key, key_count, key_width) % groups_buffer_entry_count;
  int32_t matching_slot = orc_get_matching_group_value_columnar_slot(
      groups_buffer, groups_buffer_entry_count, h, key, key_count, key_width);
  if (matching_slot != -1) {
    return (int32_t)h;
  }
  uint32_t h_probe = (h + 1) % groups_buffer_entry_count;
  while (h_probe != h) {
    matching_slot = orc_get_matching_group_value_columnar_slot(
        groups_buffer, groups_buffer_entry_count, h_probe, key, key_count, key_width);
    if (matching_slot != -1) {
      return (int32_t)h_probe;
    }
    h_probe = (h_probe + 1) % groups_buffer_entry_count;
  }
  return -1;
}

/* QE/RuntimeFunctions.cpp:1307-1328 + QE/GroupByRuntime.cpp:145-166. */
static int64_t* orc_get_matching_group_value_columnar(int64_t* groups_buffer, uint32_t h,
                                                      const int64_t* key, uint32_t key_qw_count,
                                                      size_t entry_count) {
  size_t off = h;
  if (groups_buffer[off] == HDK_EMPTY_KEY_64) {
    for (size_t i = 0; i < key_qw_count; ++i) {
      groups_buffer[off] = key[i];
      off += entry_count;
    }
    return &groups_buffer[off];
  }
  off = h;
  for (size_t i = 0; i < key_qw_count; ++i) {
    if (groups_buffer[off] != key[i]) {
      return NULL;
    }
    off += entry_count;
  }
  return &groups_buffer[off];
}

int64_t* orc_get_group_value_columnar(int64_t* groups_buffer, uint32_t groups_buffer_entry_count,
                                      const int64_t* key, uint32_t key_qw_count) {
  uint32_t h = orc_key_hash(key, key_qw_count, sizeof(int64_t)) % groups_buffer_entry_count;
  int64_t* matching_group = orc_get_matching_group_value_columnar(
      groups_buffer, h, key, key_qw_count, groups_buffer_entry_count);
  if (matching_group) {
    return matching_group;
  }
  uint32_t h_probe = (h + 1) % groups_buffer_entry_count;
  while (h_probe != h) {
    matching_group = orc_get_matching_group_value_columnar(
        groups_buffer, h_probe, key, key_qw_count, groups_buffer_entry_count);
    if (matching_group) {
      return matching_group;
    }
    h_probe = (h_probe + 1) % groups_buffer_entry_count;
  }
  return NULL;
}

/* QE/GroupByRuntime.cpp:198-213. */
int64_t* orc_get_group_value_fast(int64_t* groups_buffer, int64_t key, int64_t min_key,
                                  int64_t bucket, uint32_t row_size_quad) {
  int64_t key_diff = key - min_key;
  if (bucket) {
    key_diff /= bucket;
  }
  int64_t off = key_diff * row_size_quad;
  if (groups_buffer[off] == HDK_EMPTY_KEY_64) {
    groups_buffer[off] = key;
  }
  return groups_buffer + off + 1;
}

/* QE/RuntimeFunctions.cpp:1387-1394. */
int64_t* orc_get_group_value_fast_keyless(int64_t* groups_buffer, int64_t key, int64_t min_key,
                                          int64_t bucket, uint32_t row_size_quad) {
  (void)bucket;
  return groups_buffer + row_size_quad * (key - min_key);
}

/* QE/GroupByRuntime.cpp:232-246. */
uint32_t orc_get_columnar_group_bin_offset(int64_t* key_base_ptr, int64_t key, int64_t min_key,
                                           int64_t bucket) {
  int64_t off = key - min_key;
  if (bucket) {
    off /= bucket;
  }
  if (key_base_ptr[off] == HDK_EMPTY_KEY_64) {
    key_base_ptr[off] = key;
  }
  return (uint32_t)off;
}

/* QE/RuntimeFunctions.cpp:1341-1354. */
int64_t* orc_get_matching_group_value_perfect_hash(int64_t* groups_buffer, uint32_t hashed_index,
                                                   const int64_t* key, uint32_t key_count,
                                                   uint32_t row_size_quad) {
  uint32_t off = hashed_index * row_size_quad;
  if (groups_buffer[off] == HDK_EMPTY_KEY_64) {
    for (uint32_t i = 0; i < key_count; ++i) {
      groups_buffer[off + i] = key[i];
    }
  }
  return groups_buffer + off + key_count;
}

/* QE/RuntimeFunctions.cpp:1362-1367. */
int64_t* orc_get_matching_group_value_perfect_hash_keyless(int64_t* groups_buffer,
                                                           uint32_t hashed_index,
                                                           uint32_t row_size_quad) {
  return groups_buffer + row_size_quad * hashed_index;
}

/* QE/RuntimeFunctions.cpp:1373-1384. */
void orc_set_matching_group_value_perfect_hash_columnar(int64_t* groups_buffer, uint32_t hashed_index,
                                                        const int64_t* key, uint32_t key_count,
                                                        uint32_t entry_count) {
  if (groups_buffer[hashed_index] == HDK_EMPTY_KEY_64) {
    for (uint32_t i = 0; i < key_count; i++) {
      groups_buffer[i * entry_count + hashed_index] = key[i];
    }
  }
}

/* ============================================================================================
 * Aggregates -- QE/RuntimeFunctions.cpp:387-875
 * ========================================================================================== */

static inline double bits_to_double(int64_t b) {
  double d;
  memcpy(&d, &b, 8);
  return d;
}
static inline int64_t double_to_bits(double d) {
  int64_t b;
  memcpy(&b, &d, 8);
  return b;
}
static inline float bits_to_float(int32_t b) {
  float f;
  memcpy(&f, &b, 4);
  return f;
}
static inline int32_t float_to_bits(float f) {
  int32_t b;
  memcpy(&b, &f, 4);
  return b;
}

uint64_t orc_agg_count(uint64_t* agg, int64_t val) { /* :387-391 */
  (void)val;
  return (*agg)++;
}
uint32_t orc_agg_count_int32(uint32_t* agg, int32_t val) { /* :528-531 */
  (void)val;
  return (*agg)++;
}
int64_t orc_agg_sum(int64_t* agg, int64_t val) { /* :456-461; wraps like the reference (no check) */
  const int64_t old = *agg;
  *agg = (int64_t)((uint64_t)*agg + (uint64_t)val);
  return old;
}
int32_t orc_agg_sum_int32(int32_t* agg, int32_t val) { /* :533-538 */
  const int32_t old = *agg;
  *agg = (int32_t)((uint32_t)*agg + (uint32_t)val);
  return old;
}
void orc_agg_max(int64_t* agg, int64_t val) { /* :463-466 */
  *agg = *agg > val ? *agg : val;
}
void orc_agg_min(int64_t* agg, int64_t val) { /* :468-471 */
  *agg = *agg < val ? *agg : val;
}
void orc_agg_max_int32(int32_t* agg, int32_t val) { /* :540-549 */
  *agg = *agg > val ? *agg : val;
}
void orc_agg_min_int32(int32_t* agg, int32_t val) { /* :551-560 */
  *agg = *agg < val ? *agg : val;
}

int64_t orc_agg_sum_skip_val(int64_t* agg, int64_t val, int64_t skip_val) { /* :612-625 */
  const int64_t old = *agg;
  if (val != skip_val) {
    if (old != skip_val) {
      return orc_agg_sum(agg, val);
    } else {
      *agg = val;
    }
  }
  return old;
}
int32_t orc_agg_sum_int32_skip_val(int32_t* agg, int32_t val, int32_t skip_val) { /* :627-640 */
  const int32_t old = *agg;
  if (val != skip_val) {
    if (old != skip_val) {
      return orc_agg_sum_int32(agg, val);
    } else {
      *agg = val;
    }
  }
  return old;
}
uint64_t orc_agg_count_skip_val(uint64_t* agg, int64_t val, int64_t skip_val) { /* :642-650 */
  if (val != skip_val) {
    return orc_agg_count(agg, val);
  }
  return *agg;
}
uint32_t orc_agg_count_int32_skip_val(uint32_t* agg, int32_t val, int32_t skip_val) { /* :652-660 */
  if (val != skip_val) {
    return orc_agg_count_int32(agg, val);
  }
  return *agg;
}
/* DEF_SKIP_AGG, :670-706 */
void orc_agg_max_skip_val(int64_t* agg, int64_t val, int64_t skip_val) {
  if (val != skip_val) {
    const int64_t old_agg = *agg;
    if (old_agg != skip_val) {
      orc_agg_max(agg, val);
    } else {
      *agg = val;
    }
  }
}
void orc_agg_min_skip_val(int64_t* agg, int64_t val, int64_t skip_val) {
  if (val != skip_val) {
    const int64_t old_agg = *agg;
    if (old_agg != skip_val) {
      orc_agg_min(agg, val);
    } else {
      *agg = val;
    }
  }
}
void orc_agg_max_int32_skip_val(int32_t* agg, int32_t val, int32_t skip_val) {
  if (val != skip_val) {
    const int32_t old_agg = *agg;
    if (old_agg != skip_val) {
      orc_agg_max_int32(agg, val);
    } else {
      *agg = val;
    }
  }
}
void orc_agg_min_int32_skip_val(int32_t* agg, int32_t val, int32_t skip_val) {
  if (val != skip_val) {
    const int32_t old_agg = *agg;
    if (old_agg != skip_val) {
      orc_agg_min_int32(agg, val);
    } else {
      *agg = val;
    }
  }
}

uint64_t orc_agg_count_double(uint64_t* agg, double val) { /* :710-713 */
  (void)val;
  return (*agg)++;
}
void orc_agg_sum_double(int64_t* agg, double val) { /* :715-720 */
  const double r = bits_to_double(*agg) + val;
  *agg = double_to_bits(r);
}
void orc_agg_max_double(int64_t* agg, double val) { /* :722-729; std::max(a, b) = (a < b) ? b : a */
  const double a = bits_to_double(*agg);
  const double r = (a < val) ? val : a;
  *agg = double_to_bits(r);
}
void orc_agg_min_double(int64_t* agg, double val) { /* :731-738; std::min(a, b) = (b < a) ? b : a */
  const double a = bits_to_double(*agg);
  const double r = (val < a) ? val : a;
  *agg = double_to_bits(r);
}
uint64_t orc_agg_count_double_skip_val(uint64_t* agg, double val, double skip_val) { /* :823-831 */
  if (val != skip_val) {
    return orc_agg_count_double(agg, val);
  }
  return *agg;
}
/* DEF_SKIP_AGG (fp), :851-875: value compare on `val`, bit compare on the accumulator */
void orc_agg_sum_double_skip_val(int64_t* agg, double val, double skip_val) {
  if (val != skip_val) {
    const int64_t old_agg = *agg;
    if (old_agg != double_to_bits(skip_val)) {
      orc_agg_sum_double(agg, val);
    } else {
      *agg = double_to_bits(val);
    }
  }
}
void orc_agg_max_double_skip_val(int64_t* agg, double val, double skip_val) {
  if (val != skip_val) {
    const int64_t old_agg = *agg;
    if (old_agg != double_to_bits(skip_val)) {
      orc_agg_max_double(agg, val);
    } else {
      *agg = double_to_bits(val);
    }
  }
}
void orc_agg_min_double_skip_val(int64_t* agg, double val, double skip_val) {
  if (val != skip_val) {
    const int64_t old_agg = *agg;
    if (old_agg != double_to_bits(skip_val)) {
      orc_agg_min_double(agg, val);
    } else {
      *agg = double_to_bits(val);
    }
  }
}
void orc_agg_sum_float(int32_t* agg, float val) { /* :773-778 */
  const float r = bits_to_float(*agg) + val;
  *agg = float_to_bits(r);
}
void orc_agg_sum_float_skip_val(int32_t* agg, float val, float skip_val) {
  if (val != skip_val) {
    const int32_t old_agg = *agg;
    if (old_agg != float_to_bits(skip_val)) {
      orc_agg_sum_float(agg, val);
    } else {
      *agg = float_to_bits(val);
    }
  }
}

void orc_agg_max_float(int32_t* agg, float val) { /* :777-784, std::max(*agg, val) */
  const float a = bits_to_float(*agg);
  *agg = float_to_bits(a < val ? val : a);
}
void orc_agg_min_float(int32_t* agg, float val) { /* :786-793, std::min(*agg, val) */
  const float a = bits_to_float(*agg);
  *agg = float_to_bits(val < a ? val : a);
}
void orc_agg_max_float_skip_val(int32_t* agg, float val, float skip_val) { /* DEF_SKIP_AGG :855-875 */
  if (val != skip_val) {
    if (*agg != float_to_bits(skip_val)) orc_agg_max_float(agg, val);
    else *agg = float_to_bits(val);
  }
}
void orc_agg_min_float_skip_val(int32_t* agg, float val, float skip_val) {
  if (val != skip_val) {
    if (*agg != float_to_bits(skip_val)) orc_agg_min_float(agg, val);
    else *agg = float_to_bits(val);
  }
}

/* ============================================================================================
 * Scalar helpers
 * ========================================================================================== */

int64_t orc_scale_decimal_down_nullable(int64_t operand, int64_t scale, int64_t null_val) {
  /* QE/RuntimeFunctions.cpp:240-252: rounded scale-down */
  if (operand == null_val) {
    return null_val;
  }
  int64_t tmp = scale >> 1;
  tmp = operand >= 0 ? operand + tmp : operand - tmp;
  return tmp / scale;
}
int64_t orc_scale_decimal_down_not_nullable(int64_t operand, int64_t scale, int64_t null_val) {
  (void)null_val; /* :254-261 */
  int64_t tmp = scale >> 1;
  tmp = operand >= 0 ? operand + tmp : operand - tmp;
  return tmp / scale;
}
int64_t orc_floor_div_lhs(int64_t dividend, int64_t divisor) { /* :265-268 */
  return (dividend < 0 ? dividend - (divisor - 1) : dividend) / divisor;
}
int64_t orc_floor_div_nullable_lhs(int64_t dividend, int64_t divisor, int64_t null_val) { /* :272-277 */
  return dividend == null_val ? null_val : orc_floor_div_lhs(dividend, divisor);
}

/* omniscidb/Utils/ExtractFromTime.cpp:156-162 (fast path), :260-272 (general). */
int64_t orc_extract_year(int64_t timeval) {
  const uint32_t kEpochOffsetYear1900 = 2208988800u;
  const uint32_t kSecsJanToMar1900 = 5097600u;
  const uint32_t kSecondsPer4YearCycle = 126230400u;
  const uint32_t kUSecsPerDay = 86400u;
  const uint32_t kSecondsPerNonLeapYear = 31536000u;
  if (timeval >= 0LL && timeval <= (int64_t)(UINT32_MAX - kEpochOffsetYear1900)) {
    const uint32_t seconds_1900 = (uint32_t)timeval + kEpochOffsetYear1900;
    const uint32_t leap_years = (seconds_1900 - kSecsJanToMar1900) / kSecondsPer4YearCycle;
    const uint32_t year =
        (seconds_1900 - leap_years * kUSecsPerDay) / kSecondsPerNonLeapYear + 1900;
    return (int32_t)year;
  }
  const int64_t kSecsPerDay = 86400;
  const int64_t kEpochAdjustedDays = 11017;
  const int64_t kDaysPer400Years = 146097;
  const unsigned MARJAN = 31 + 30 + 31 + 30 + 31 + 31 + 30 + 31 + 30 + 31;
  const int64_t day = orc_floor_div_lhs(timeval, kSecsPerDay);
  const int64_t era = orc_floor_div_lhs(day - kEpochAdjustedDays, kDaysPer400Years);
  const unsigned doe = (unsigned)(day - kEpochAdjustedDays - era * kDaysPer400Years);
  const unsigned yoe = (doe - doe / 1460 + doe / 36524 - (doe == 146096)) / 365;
  const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  return 2000 + era * 400 + yoe + (MARJAN <= doy);
}

int8_t orc_logical_and(int8_t lhs, int8_t rhs, int8_t null_val) { /* :361-372 */
  if (lhs == null_val) {
    return rhs == 0 ? rhs : null_val;
  }
  if (rhs == null_val) {
    return lhs == 0 ? lhs : null_val;
  }
  return (lhs && rhs) ? 1 : 0;
}
int8_t orc_logical_or(int8_t lhs, int8_t rhs, int8_t null_val) { /* :374-384 */
  if (lhs == null_val) {
    return rhs == 0 ? null_val : rhs;
  }
  if (rhs == null_val) {
    return lhs == 0 ? null_val : lhs;
  }
  return (lhs || rhs) ? 1 : 0;
}
int8_t orc_logical_not(int8_t operand, int8_t null_val) { /* :355-358 */
  return operand == null_val ? operand : (operand ? 0 : 1);
}

/* ============================================================================================
 * Join probe -- QE/GroupByRuntime.cpp:274-366, JoinHashImpl.h:84-97
 * ========================================================================================== */

int64_t orc_hash_join_idx(const int32_t* hash_buff, int64_t key, int64_t min_key, int64_t max_key) {
  if (key >= min_key && key <= max_key) {
    return hash_buff[key - min_key];
  }
  return -1;
}
int64_t orc_bucketized_hash_join_idx(const int32_t* hash_buff, int64_t key, int64_t min_key,
                                     int64_t max_key, int64_t bucket_normalization) {
  if (key >= min_key && key <= max_key) {
    return hash_buff[(key - min_key) / bucket_normalization];
  }
  return -1;
}
int64_t orc_hash_join_idx_nullable(const int32_t* hash_buff, int64_t key, int64_t min_key,
                                   int64_t max_key, int64_t null_val) {
  return key != null_val ? orc_hash_join_idx(hash_buff, key, min_key, max_key) : -1;
}
int64_t orc_bucketized_hash_join_idx_nullable(const int32_t* hash_buff, int64_t key, int64_t min_key, int64_t max_key,
                                              int64_t null_val, int64_t bucket_normalization) { /* :312-323 */
  return key != null_val ? orc_bucketized_hash_join_idx(hash_buff, key, min_key, max_key, bucket_normalization) : -1;
}
int64_t orc_bucketized_hash_join_idx_bitwise(const int32_t* hash_buff, int64_t key, int64_t min_key, int64_t max_key,
                                             int64_t null_val, int64_t translated_val,
                                             int64_t bucket_normalization) { /* :334-350 */
  return key != null_val
             ? orc_bucketized_hash_join_idx(hash_buff, key, min_key, max_key, bucket_normalization)
             : orc_bucketized_hash_join_idx(hash_buff, translated_val, min_key, translated_val, bucket_normalization);
}
int64_t orc_hash_join_idx_bitwise(const int32_t* hash_buff, int64_t key, int64_t min_key,
                                  int64_t max_key, int64_t null_val, int64_t translated_val) {
  return key != null_val ? orc_hash_join_idx(hash_buff, key, min_key, max_key)
                         : orc_hash_join_idx(hash_buff, translated_val, min_key, translated_val);
}

/* ============================================================================================
 * Join build -- QE/JoinHashTable/Runtime/HashJoinRuntime.cpp
 * ========================================================================================== */

void orc_init_hash_join_buff(int32_t* buff, int64_t entry_count, int32_t invalid_slot_val) {
  for (int64_t i = 0; i < entry_count; ++i) { /* :127-147 */
    buff[i] = invalid_slot_val;
  }
}

/* JoinColumnTyped element decode: QE/JoinHashTable/Runtime/JoinColumnIterator.h (getElementAt):
 * SmallDate -> fixed_width_small_date_decode, Signed -> int decode, Unsigned -> unsigned decode. */
static inline int64_t join_elem(const hdk_hip_join_chunk* c, size_t i,
                                const hdk_hip_join_column_type_info* ti) {
  switch (ti->column_type) {
    case HDK_JC_SMALL_DATE: {
      int64_t v = orc_fixed_width_int_decode(c->col_buff, (int32_t)ti->elem_sz, (int64_t)i);
      return v == (int32_t)ti->null_val ? ti->null_val : v * 86400; /* DecodersImpl.h:152-160 */
    }
    case HDK_JC_UNSIGNED:
      return orc_fixed_width_unsigned_decode(c->col_buff, (int32_t)ti->elem_sz, (int64_t)i);
    default:
      return orc_fixed_width_int_decode(c->col_buff, (int32_t)ti->elem_sz, (int64_t)i);
  }
}

static inline int32_t* join_slot(int32_t* buff, int64_t elem, int64_t min_val, int64_t bucket) {
  /* JoinHashImpl.h:84-97 */
  if (bucket > 1) {
    return buff + (elem - min_val) / bucket;
  }
  return buff + (elem - min_val);
}

/* fill_hash_join_buff[_bucketized]: HashJoinRuntime.cpp:197-293; slot fill JoinHashImpl.h:55-80. */
int orc_fill_hash_join_buff(int32_t* buff, int32_t invalid_slot_val, int32_t for_semi_join,
                            const hdk_hip_join_chunk* chunks, size_t num_chunks,
                            const hdk_hip_join_column_type_info* ti, int64_t bucket_normalization) {
  size_t index = 0;
  for (size_t c = 0; c < num_chunks; ++c) {
    for (size_t i = 0; i < chunks[c].num_elems; ++i, ++index) {
      int64_t elem = join_elem(&chunks[c], i, ti);
      if (elem == ti->null_val) {
        if (ti->uses_bw_eq) {
          elem = ti->translated_null_val;
        } else {
          continue;
        }
      }
      int32_t* entry_ptr = join_slot(buff, elem, ti->min_val, bucket_normalization);
      if (*entry_ptr == invalid_slot_val) { /* CAS(invalid -> idx) */
        *entry_ptr = (int32_t)index;
      } else if (!for_semi_join) {
        return -1; /* slot is full */
      }
    }
  }
  return 0;
}

/* fill_one_to_many_hash_table_impl: HashJoinRuntime.cpp:1140-1190 with count_matches :589-636 and
 * fill_row_ids :770-822.  buff = [pos | count | ids]. */
void orc_fill_one_to_many_hash_table(int32_t* buff, int64_t hash_entry_count, int32_t invalid_slot_val,
                                     const hdk_hip_join_chunk* chunks, size_t num_chunks,
                                     const hdk_hip_join_column_type_info* ti,
                                     int64_t bucket_normalization) {
  (void)invalid_slot_val;
  int32_t* pos_buff = buff;
  int32_t* count_buff = buff + hash_entry_count;
  int32_t* id_buff = count_buff + hash_entry_count;
  memset(count_buff, 0, (size_t)hash_entry_count * sizeof(int32_t));
  /* count_matches */
  for (size_t c = 0; c < num_chunks; ++c) {
    for (size_t i = 0; i < chunks[c].num_elems; ++i) {
      int64_t elem = join_elem(&chunks[c], i, ti);
      if (elem == ti->null_val) {
        if (ti->uses_bw_eq) {
          elem = ti->translated_null_val;
        } else {
          continue;
        }
      }
      (*join_slot(count_buff, elem, ti->min_val, bucket_normalization))++;
    }
  }
  /* count_copy[0] = 0; count_copy[1..] = count[0..n-2]; inclusive scan; pos where count != 0 */
  int32_t* count_copy = (int32_t*)calloc((size_t)hash_entry_count, sizeof(int32_t));
  memcpy(count_copy + 1, count_buff, (size_t)(hash_entry_count - 1) * sizeof(int32_t));
  int32_t sum = 0;
  for (int64_t i = 0; i < hash_entry_count; ++i) {
    sum += count_copy[i];
    count_copy[i] = sum;
  }
  for (int64_t i = 0; i < hash_entry_count; ++i) {
    if (count_buff[i]) {
      pos_buff[i] = count_copy[i];
    }
  }
  free(count_copy);
  memset(count_buff, 0, (size_t)hash_entry_count * sizeof(int32_t));
  /* fill_row_ids */
  size_t index = 0;
  for (size_t c = 0; c < num_chunks; ++c) {
    for (size_t i = 0; i < chunks[c].num_elems; ++i, ++index) {
      int64_t elem = join_elem(&chunks[c], i, ti);
      if (elem == ti->null_val) {
        if (ti->uses_bw_eq) {
          elem = ti->translated_null_val;
        } else {
          continue;
        }
      }
      int32_t* pos_ptr = join_slot(pos_buff, elem, ti->min_val, bucket_normalization);
      const int64_t bin_idx = pos_ptr - pos_buff;
      const int32_t id_buff_idx = count_buff[bin_idx]++ + *pos_ptr;
      id_buff[id_buff_idx] = (int32_t)index;
    }
  }
}

/* ============================================================================================
 * Output buffer init -- QE/GpuInitGroups.cu:17-166
 * ========================================================================================== */

void orc_init_group_by_buffer(int64_t* groups_buffer, const int64_t* init_vals,
                              uint32_t groups_buffer_entry_count, uint32_t key_count,
                              uint32_t key_width, uint32_t row_size_quad, int32_t keyless,
                              int8_t warp_size) {
  if (keyless) { /* :127-135 */
    const int64_t n = (int64_t)groups_buffer_entry_count * row_size_quad * (int32_t)warp_size;
    for (int64_t i = 0; i < n; ++i) {
      groups_buffer[i] = init_vals[i % row_size_quad];
    }
    return;
  }
  for (uint32_t i = 0; i < groups_buffer_entry_count; ++i) { /* :137-151 */
    int64_t* keys_ptr = groups_buffer + (size_t)i * row_size_quad;
    if (key_width == 4) {
      int32_t* k = (int32_t*)keys_ptr;
      for (uint32_t j = 0; j < key_count; ++j) {
        k[j] = HDK_EMPTY_KEY_32;
      }
    } else if (key_width == 8) {
      for (uint32_t j = 0; j < key_count; ++j) {
        keys_ptr[j] = HDK_EMPTY_KEY_64;
      }
    }
  }
  const uint32_t values_off_quad = (uint32_t)(align_to_int64_sz(key_count * key_width) / 8); /* :153 */
  for (uint32_t i = 0; i < groups_buffer_entry_count; ++i) {
    int64_t* vals_ptr = groups_buffer + (size_t)i * row_size_quad + values_off_quad;
    const uint32_t val_count = row_size_quad - values_off_quad;
    for (uint32_t j = 0; j < val_count; ++j) {
      vals_ptr[j] = init_vals[j];
    }
  }
}

void orc_init_columnar_group_by_buffer(int64_t* groups_buffer, const int64_t* init_vals,
                                       uint32_t entry_count, uint32_t key_count,
                                       uint32_t agg_col_count, const int8_t* col_sizes,
                                       int32_t need_padding, int32_t keyless, int8_t key_size) {
  int8_t* buffer_ptr = (int8_t*)groups_buffer; /* :17-108 */
  if (!keyless) {
    for (uint32_t i = 0; i < key_count; ++i) {
      switch (key_size) {
        case 1:
          for (uint32_t e = 0; e < entry_count; ++e) ((int8_t*)buffer_ptr)[e] = HDK_EMPTY_KEY_8;
          buffer_ptr += entry_count;
          break;
        case 2:
          for (uint32_t e = 0; e < entry_count; ++e) ((int16_t*)buffer_ptr)[e] = HDK_EMPTY_KEY_16;
          buffer_ptr += (size_t)entry_count * 2;
          break;
        case 4:
          for (uint32_t e = 0; e < entry_count; ++e) ((int32_t*)buffer_ptr)[e] = HDK_EMPTY_KEY_32;
          buffer_ptr += (size_t)entry_count * 4;
          break;
        case 8:
          for (uint32_t e = 0; e < entry_count; ++e) ((int64_t*)buffer_ptr)[e] = HDK_EMPTY_KEY_64;
          buffer_ptr += (size_t)entry_count * 8;
          break;
        default:
          break;
      }
      buffer_ptr = align_to_int64_ptr(buffer_ptr);
    }
  }
  int32_t init_idx = 0;
  for (uint32_t i = 0; i < agg_col_count; ++i) {
    if (need_padding) {
      buffer_ptr = align_to_int64_ptr(buffer_ptr);
    }
    switch (col_sizes[i]) {
      case 1:
        for (uint32_t e = 0; e < entry_count; ++e) ((int8_t*)buffer_ptr)[e] = (int8_t)init_vals[init_idx];
        init_idx++;
        buffer_ptr += entry_count;
        break;
      case 2:
        for (uint32_t e = 0; e < entry_count; ++e) ((int16_t*)buffer_ptr)[e] = (int16_t)init_vals[init_idx];
        init_idx++;
        buffer_ptr += (size_t)entry_count * 2;
        break;
      case 4:
        for (uint32_t e = 0; e < entry_count; ++e) ((int32_t*)buffer_ptr)[e] = (int32_t)init_vals[init_idx];
        init_idx++;
        buffer_ptr += (size_t)entry_count * 4;
        break;
      case 8:
        for (uint32_t e = 0; e < entry_count; ++e) ((int64_t*)buffer_ptr)[e] = init_vals[init_idx];
        init_idx++;
        buffer_ptr += (size_t)entry_count * 8;
        break;
      case 0:
        continue;
      default:
        break;
    }
  }
}

/* ============================================================================================
 * Keyed ("baseline") join tables: composite or wide keys, open addressing with MurmurHash1.
 * Probe: QE/JoinHashTable/Runtime/JoinHashTableQueryRuntime.cpp:25-98 (one-to-one),
 * :130-172 (get_composite_key_index, one-to-many).  Build: HashJoinRuntime.cpp:357-507 (one-to-one),
 * :723-768 + :889-950 (count_matches_baseline / fill_row_ids_baseline) with the key handler of
 * HashJoinKeyHandlers.h:36-100 (rows with a NULL component are skipped).
 * ========================================================================================== */
static int compare_to_key(const int8_t* entry, const int8_t* key, size_t key_bytes) {
  return memcmp(entry, key, key_bytes) == 0;
}

#define ORC_BASELINE_IDX(NAME, T, INVALID)                                                            \
  int64_t NAME(const int8_t* hash_buff, const int8_t* key, size_t key_bytes, size_t entry_count) {    \
    if (!entry_count) return -1; /* kNoMatch */                                                       \
    const uint32_t h = orc_murmur_hash1(key, (int)key_bytes, 0) % entry_count;                        \
    uint32_t hp = h;                                                                                  \
    do {                                                                                              \
      const int8_t* e = hash_buff + (size_t)hp * (key_bytes + sizeof(T));                             \
      if (compare_to_key(e, key, key_bytes)) return *(const T*)(e + key_bytes);                       \
      if (*(const T*)e == (INVALID)) return -2; /* kNotPresent */                                     \
      hp = (hp + 1) % entry_count;                                                                    \
    } while (hp != h);                                                                                \
    return -1;                                                                                        \
  }
ORC_BASELINE_IDX(orc_baseline_hash_join_idx_32, int32_t, INT32_MAX)
ORC_BASELINE_IDX(orc_baseline_hash_join_idx_64, int64_t, INT64_MAX)

#define ORC_COMPOSITE_IDX(NAME, T, INVALID)                                                           \
  int64_t NAME(const T* key, size_t key_component_count, const T* dict, size_t entry_count) {         \
    const uint32_t h = orc_murmur_hash1(key, (int)(key_component_count * sizeof(T)), 0) % entry_count; \
    uint32_t off = h * key_component_count;                                                           \
    if (memcmp(&dict[off], key, key_component_count * sizeof(T)) == 0) return h;                      \
    uint32_t hp = (h + 1) % entry_count;                                                              \
    while (hp != h) {                                                                                 \
      off = hp * key_component_count;                                                                 \
      if (memcmp(&dict[off], key, key_component_count * sizeof(T)) == 0) return hp;                   \
      if (dict[off] == (INVALID)) return -1;                                                          \
      hp = (hp + 1) % entry_count;                                                                    \
    }                                                                                                 \
    return -1;                                                                                        \
  }
ORC_COMPOSITE_IDX(orc_get_composite_key_index_32, int32_t, INT32_MAX)
ORC_COMPOSITE_IDX(orc_get_composite_key_index_64, int64_t, INT64_MAX)

/* init_baseline_hash_join_buff (HashJoinRuntime.cpp:296-349): keys = invalid key, payload = invalid slot */
void orc_init_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count, size_t key_component_count,
                                      int32_t key_component_width, int32_t with_val_slot,
                                      int32_t invalid_slot_val) {
  const size_t comps = key_component_count + (with_val_slot ? 1 : 0);
  for (int64_t e = 0; e < entry_count; ++e) {
    for (size_t i = 0; i < comps; ++i) {
      const int is_val = with_val_slot && i == key_component_count;
      if (key_component_width == 4) {
        ((int32_t*)hash_buff)[e * comps + i] = is_val ? invalid_slot_val : INT32_MAX;
      } else {
        ((int64_t*)hash_buff)[e * comps + i] = is_val ? (int64_t)invalid_slot_val : INT64_MAX;
      }
    }
  }
}

/* one row's composite key through the GenericKeyHandler; returns 0 when the row is skipped (NULL) */
static int composite_key_of_row(const hdk_hip_join_column* cols, const hdk_hip_join_column_type_info* ti,
                                size_t ncols, size_t row, int64_t* key) {
  for (size_t k = 0; k < ncols; ++k) {
    /* locate the chunk holding `row` (JoinColumnIterator, HashJoinRuntime.h:126-205) */
    const hdk_hip_join_chunk* chunks = (const hdk_hip_join_chunk*)cols[k].col_chunks_buff;
    size_t r = row;
    size_t ci = 0;
    while (r >= chunks[ci].num_elems) {
      r -= chunks[ci].num_elems;
      ++ci;
    }
    const int64_t elem = join_elem(&chunks[ci], r, &ti[k]);
    if (elem == ti[k].null_val && !ti[k].uses_bw_eq) {
      return 0;
    }
    key[k] = elem;
  }
  return 1;
}

/* find-or-insert of write_baseline_hash_slot / get_matching_baseline_hash_slot_at (sequential) */
static int64_t keyed_slot_for_insert(int8_t* hash_buff, int64_t entry_count, const int64_t* key, size_t kc,
                                     int32_t w, size_t entry_bytes) {
  int32_t k32[HDK_HIP_MAX_JOIN_KEYS];
  for (size_t i = 0; i < kc; ++i) k32[i] = (int32_t)key[i];
  const void* kb = w == 4 ? (const void*)k32 : (const void*)key;
  const size_t key_bytes = kc * (size_t)w;
  const uint32_t h = orc_murmur_hash1(kb, (int)key_bytes, 0) % entry_count;
  uint32_t hp = h;
  do {
    int8_t* e = hash_buff + (size_t)hp * entry_bytes;
    const int empty = w == 4 ? *(int32_t*)e == INT32_MAX : *(int64_t*)e == INT64_MAX;
    if (empty) {
      memcpy(e, kb, key_bytes);
      return hp;
    }
    if (memcmp(e, kb, key_bytes) == 0) {
      return hp;
    }
    hp = (hp + 1) % entry_count;
  } while (hp != h);
  return -1;
}

/* fill_baseline_hash_join_buff (HashJoinRuntime.cpp:509-573): returns 0, -1 (duplicate key: the caller
 * falls back to one-to-many) or -2 (table full) */
int orc_fill_baseline_hash_join_buff(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                     size_t key_component_count, int32_t key_component_width,
                                     const hdk_hip_join_column* cols, const hdk_hip_join_column_type_info* ti) {
  return orc_fill_baseline_hash_join_buff_semi(hash_buff, entry_count, invalid_slot_val, 0, key_component_count,
                                               key_component_width, cols, ti);
}

/* for_semi_join: write_baseline_hash_slot_for_semi_join (HashJoinRuntime.cpp:480-507) -- the first row of a key
 * keeps the slot, later rows of the key are not an error */
int orc_fill_baseline_hash_join_buff_semi(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                          int32_t for_semi_join, size_t key_component_count,
                                          int32_t key_component_width, const hdk_hip_join_column* cols,
                                          const hdk_hip_join_column_type_info* ti) {
  const size_t entry_bytes = (key_component_count + 1) * (size_t)key_component_width;
  const size_t n = cols[0].num_elems;
  for (size_t row = 0; row < n; ++row) {
    int64_t key[HDK_HIP_MAX_JOIN_KEYS];
    if (!composite_key_of_row(cols, ti, key_component_count, row, key)) continue;
    const int64_t slot = keyed_slot_for_insert(hash_buff, entry_count, key, key_component_count,
                                               key_component_width, entry_bytes);
    if (slot < 0) return -2;
    int8_t* val = hash_buff + (size_t)slot * entry_bytes + key_component_count * (size_t)key_component_width;
    if (key_component_width == 4) {
      if (*(int32_t*)val != invalid_slot_val) {
        if (for_semi_join) continue;
        return -1;
      }
      *(int32_t*)val = (int32_t)row;
    } else {
      if (*(int64_t*)val != (int64_t)invalid_slot_val) {
        if (for_semi_join) continue;
        return -1;
      }
      *(int64_t*)val = (int64_t)row;
    }
  }
  return 0;
}

/* keyed one-to-many: composite key dictionary + [offsets | counts | row ids]
 * (BaselineJoinHashTableBuilder; count_matches_baseline, inclusive scan, fill_row_ids_baseline) */
int orc_fill_one_to_many_baseline_hash_table(int8_t* hash_buff, int64_t entry_count, int32_t invalid_slot_val,
                                             size_t key_component_count, int32_t key_component_width,
                                             const hdk_hip_join_column* cols,
                                             const hdk_hip_join_column_type_info* ti) {
  const size_t entry_bytes = key_component_count * (size_t)key_component_width;
  int32_t* pos_buff = (int32_t*)(hash_buff + (size_t)entry_count * entry_bytes);
  int32_t* count_buff = pos_buff + entry_count;
  int32_t* id_buff = count_buff + entry_count;
  const size_t n = cols[0].num_elems;
  for (int64_t e = 0; e < entry_count; ++e) {
    pos_buff[e] = invalid_slot_val;
    count_buff[e] = 0;
  }
  for (size_t row = 0; row < n; ++row) {
    int64_t key[HDK_HIP_MAX_JOIN_KEYS];
    if (!composite_key_of_row(cols, ti, key_component_count, row, key)) continue;
    const int64_t slot = keyed_slot_for_insert(hash_buff, entry_count, key, key_component_count,
                                               key_component_width, entry_bytes);
    if (slot < 0) return -2;
    count_buff[slot]++;
  }
  int32_t running = 0;
  for (int64_t e = 0; e < entry_count; ++e) { /* set_valid_pos over the exclusive scan of the counts */
    if (count_buff[e]) {
      pos_buff[e] = running;
      running += count_buff[e];
      count_buff[e] = 0;
    }
  }
  for (size_t row = 0; row < n; ++row) {
    int64_t key[HDK_HIP_MAX_JOIN_KEYS];
    if (!composite_key_of_row(cols, ti, key_component_count, row, key)) continue;
    const int64_t slot = keyed_slot_for_insert(hash_buff, entry_count, key, key_component_count,
                                               key_component_width, entry_bytes);
    id_buff[pos_buff[slot] + count_buff[slot]++] = (int32_t)row;
  }
  return 0;
}

/* ============================================================================================
 * The row function for a plan
 * ========================================================================================== */

typedef struct {
  int64_t v; /* int64 value, or double bits */
} orc_val;

typedef struct {
  const hdk_hip_plan* plan;
  const int8_t* const* cols; /* col_buffers[frag] */
  int64_t pos;               /* outer row */
  int64_t join_row[HDK_HIP_MAX_JOINS];
} orc_row_ctx;

static inline int64_t load_col(const orc_row_ctx* c, int32_t col_idx) {
  const hdk_hip_col* col = &c->plan->cols[col_idx];
  const int64_t row = col->table == 0 ? c->pos : c->join_row[col->table - 1];
  const int8_t* buf = c->cols[col->buf_idx];
  switch (col->kind) {
    case HDK_COL_SMALL_DATE: /* FixedWidthSmallDate::codegenDecode, QE/Codec.cpp:86-102 */
      return orc_fixed_width_small_date_decode(buf, col->width, col->width == 4 ? INT32_MIN : INT16_MIN, INT64_MIN,
                                               row);
    case HDK_COL_UNSIGNED:
      return orc_fixed_width_unsigned_decode(buf, col->width, row);
    case HDK_COL_FLOAT:
      return double_to_bits((double)orc_fixed_width_float_decode(buf, row));
    case HDK_COL_DOUBLE:
      return double_to_bits(orc_fixed_width_double_decode(buf, row));
    default:
      return orc_fixed_width_int_decode(buf, col->width, row);
  }
}

static inline int64_t load_leaf(const orc_row_ctx* c, const hdk_hip_leaf* l) {
  switch (l->kind) {
    case HDK_LEAF_COL: {
      /* LEFT join without a match: every column of the inner table is NULL
       * (codegenOuterJoinNullPlaceholder, QE/ColumnIR.cpp) */
      const int32_t tb = c->plan->cols[l->col].table;
      if (tb > 0 && c->join_row[tb - 1] < 0) {
        return l->null_val;
      }
      return load_col(c, l->col);
    }
    case HDK_LEAF_INT:
    case HDK_LEAF_FP:
      return l->ival;
    default:
      return 0;
  }
}

static inline int leaf_is_fp(const hdk_hip_plan* p, const hdk_hip_leaf* l) {
  if (l->kind == HDK_LEAF_FP) {
    return 1;
  }
  if (l->kind == HDK_LEAF_COL) {
    return p->cols[l->col].kind == HDK_COL_FLOAT || p->cols[l->col].kind == HDK_COL_DOUBLE;
  }
  return 0;
}

/* In-band NULL test.  Integers compare by value; fp compare as the reference's `lhs != null_val`
 * on doubles does (value compare: NULL_DOUBLE = DBL_MIN is an ordinary number). */
static inline int is_null_i(int64_t v, int64_t null_val, int nullable) {
  return nullable && v == null_val;
}
static inline int is_null_f(int64_t v, int64_t null_val, int nullable) {
  return nullable && bits_to_double(v) == bits_to_double(null_val);
}

/* Evaluate an expression chain; *err receives ERR_DIV_BY_ZERO when the reference's
 * div-by-zero check (QE/ArithmeticIR.cpp codegenDivZeroCheck) would fire. */
/* + - * with the overflow check the reference generates around them (QE/ArithmeticIR.cpp: codegenAdd :277-337,
 * codegenSub :339-417, codegenMul :419-520): with chosen_max / chosen_min the limits of the operation's SQL type
 * (`width` bytes),
 *   add: (lhs > 0 && rhs > max - lhs) || (lhs < 0 && rhs < min - lhs)
 *   sub: the mirrored test;   mul: |lhs| > limit / |rhs| with limit = max (+1 when the signs differ)
 * i.e. exactly "the mathematical result does not fit the type".  NULL operands are not checked (the caller has
 * already handled them).  Values are carried as int64: for widths below 8 the exact result is compared with the
 * type's range, width 8 uses the overflow-detecting builtins.  Returns 1 on overflow; *r receives the wrapped result. */
static int checked_arith(int op, int64_t a, int64_t b, int32_t width, int64_t* r) {
  long long res;
  int ovf;
  switch (op) {
    case HDK_OP_ADD: ovf = __builtin_saddll_overflow(a, b, &res); break;
    case HDK_OP_SUB: ovf = __builtin_ssubll_overflow(a, b, &res); break;
    default: ovf = __builtin_smulll_overflow(a, b, &res); break;
  }
  *r = res;
  if (width <= 0) {
    return 0;
  }
  if (width < 8) {
    const int64_t lim = (int64_t)1 << (8 * width - 1);
    return ovf || res > lim - 1 || res < -lim;
  }
  return ovf;
}

static int64_t eval_expr(const orc_row_ctx* c, const hdk_hip_expr* e, int32_t* err) {
  const hdk_hip_plan* p = c->plan;
  int64_t acc = load_leaf(c, &e->leaf0);
  int acc_fp = leaf_is_fp(p, &e->leaf0);
  int64_t acc_null = e->leaf0.null_val;
  int acc_nullable = e->leaf0.nullable;
  for (int s = 0; s < e->nsteps; ++s) {
    const hdk_hip_step* st = &e->steps[s];
    const int lhs_null = acc_fp ? is_null_f(acc, acc_null, acc_nullable)
                                : is_null_i(acc, acc_null, acc_nullable);
    int64_t r = 0;
    int r_is_null = 0;
    switch (st->op) {
      case HDK_OP_ADD:
      case HDK_OP_SUB:
      case HDK_OP_MUL:
      case HDK_OP_DIV:
      case HDK_OP_MOD: {
        int64_t rhs = load_leaf(c, &st->rhs);
        const int rhs_fp = leaf_is_fp(p, &st->rhs);
        const int rhs_null = rhs_fp ? is_null_f(rhs, st->rhs.null_val, st->rhs.nullable)
                                    : is_null_i(rhs, st->rhs.null_val, st->rhs.nullable);
        if (lhs_null || rhs_null) { /* DEF_ARITH_NULLABLE, RuntimeFunctions.cpp:49-81 */
          r_is_null = 1;
          break;
        }
        if (st->out_class == HDK_VC_FP) {
          const double a = acc_fp ? bits_to_double(acc) : (double)acc;
          const double b = rhs_fp ? bits_to_double(rhs) : (double)rhs;
          double d = 0;
          switch (st->op) {
            case HDK_OP_ADD: d = a + b; break;
            case HDK_OP_SUB: d = a - b; break;
            case HDK_OP_MUL: d = a * b; break;
            case HDK_OP_DIV:
              if (b == 0.0) { *err = HDK_HIP_ERR_DIV_BY_ZERO; r_is_null = 1; }
              else d = a / b;
              break;
            default: r_is_null = 1; break;
          }
          r = double_to_bits(d);
        } else {
          const int64_t a = acc, b = rhs;
          switch (st->op) {
            case HDK_OP_ADD:
            case HDK_OP_SUB:
            case HDK_OP_MUL:
              if (checked_arith(st->op, a, b, st->check_width, &r)) {
                *err = HDK_HIP_ERR_OVERFLOW_OR_UNDERFLOW;
              }
              break;
            case HDK_OP_DIV:
              if (b == 0) { *err = HDK_HIP_ERR_DIV_BY_ZERO; r_is_null = 1; }
              else if (a == INT64_MIN && b == -1) r = INT64_MIN;
              else r = a / b;
              break;
            case HDK_OP_MOD:
              if (b == 0) { *err = HDK_HIP_ERR_DIV_BY_ZERO; r_is_null = 1; }
              else if (b == -1) r = 0;
              else r = a % b;
              break;
          }
        }
        break;
      }
      case HDK_OP_EXTRACT_YEAR: /* ExtractFromTime, nullable: DateTimeIR.cpp codegen null check */
        if (lhs_null) r_is_null = 1; else r = orc_extract_year(acc);
        break;
      case HDK_OP_SCALE_DOWN:
        if (lhs_null) r_is_null = 1;
        else r = orc_scale_decimal_down_not_nullable(acc, st->rhs.ival, 0);
        break;
      case HDK_OP_FLOOR_DIV:
        if (lhs_null) r_is_null = 1; else r = orc_floor_div_lhs(acc, st->rhs.ival);
        break;
      case HDK_OP_CAST_INT_TO_FP: /* cast_int64_t_to_double_nullable */
        if (lhs_null) r_is_null = 1; else r = double_to_bits((double)acc);
        break;
      case HDK_OP_CAST_FP_TO_INT: { /* DEF_ROUND_NULLABLE(double, int64_t) */
        if (lhs_null) { r_is_null = 1; break; }
        const double d = bits_to_double(acc);
        r = (int64_t)(d + (d < 0.0 ? -0.5 : 0.5));
        break;
      }
      default:
        r_is_null = 1;
        break;
    }
    acc = r_is_null ? st->null_out : r;
    acc_fp = st->out_class == HDK_VC_FP;
    acc_null = st->null_out;
    acc_nullable = 1;
  }
  return acc;
}

/* one filter conjunct -> {0, 1, NULL_BOOLEAN}: DEF_CMP_NULLABLE, RuntimeFunctions.cpp:83-117 */
static int8_t eval_qual(const orc_row_ctx* c, const hdk_hip_qual* q, int32_t* err) {
  const int64_t lhs = eval_expr(c, &q->lhs, err);
  const int64_t rhs = load_leaf(c, &q->rhs);
  const int lhs_fp = q->lhs.vclass == HDK_VC_FP;
  const int rhs_fp = leaf_is_fp(c->plan, &q->rhs);
  const int lnull = lhs_fp ? is_null_f(lhs, q->lhs.null_val, q->lhs.nullable)
                           : is_null_i(lhs, q->lhs.null_val, q->lhs.nullable);
  const int rnull = rhs_fp ? is_null_f(rhs, q->rhs.null_val, q->rhs.nullable)
                           : is_null_i(rhs, q->rhs.null_val, q->rhs.nullable);
  if (lnull || rnull) {
    return INT8_MIN; /* NULL_BOOLEAN */
  }
  if (lhs_fp || rhs_fp) {
    const double a = lhs_fp ? bits_to_double(lhs) : (double)lhs;
    const double b = rhs_fp ? bits_to_double(rhs) : (double)rhs;
    switch (q->cmp) {
      case HDK_CMP_EQ: return a == b;
      case HDK_CMP_NE: return a != b;
      case HDK_CMP_LT: return a < b;
      case HDK_CMP_GT: return a > b;
      case HDK_CMP_LE: return a <= b;
      case HDK_CMP_GE: return a >= b;
    }
  } else {
    switch (q->cmp) {
      case HDK_CMP_EQ: return lhs == rhs;
      case HDK_CMP_NE: return lhs != rhs;
      case HDK_CMP_LT: return lhs < rhs;
      case HDK_CMP_GT: return lhs > rhs;
      case HDK_CMP_LE: return lhs <= rhs;
      case HDK_CMP_GE: return lhs >= rhs;
    }
  }
  return 0;
}

/* The filter of a plan at one stage (0: before the joins, 1: after them): the conjunction of the stage's quals, or --
 * with a filter program -- the postfix expression over all quals, combined with the reference's three-valued
 * logical_and / logical_or / logical_not (orc_logical_*, pinned against RuntimeFunctions.cpp:357-384). */
static int filter_pass(const orc_row_ctx* c, int stage, int32_t* err) {
  const hdk_hip_plan* p = c->plan;
  if (p->num_filter_ops) {
    if ((p->filter_after_joins != 0) != (stage != 0)) {
      return 1;
    }
    int8_t stack[HDK_HIP_MAX_FILTER_OPS + 1];
    int sp = 0;
    for (int i = 0; i < p->num_filter_ops; ++i) {
      const uint8_t op = p->filter_ops[i];
      if (op < HDK_F_AND) {
        stack[sp++] = eval_qual(c, &p->quals[op], err);
      } else if (op == HDK_F_NOT) {
        stack[sp - 1] = orc_logical_not(stack[sp - 1], INT8_MIN);
      } else {
        const int8_t b = stack[--sp], a = stack[sp - 1];
        stack[sp - 1] = op == HDK_F_AND ? orc_logical_and(a, b, INT8_MIN) : orc_logical_or(a, b, INT8_MIN);
      }
    }
    return sp == 1 && stack[0] == 1;
  }
  for (int q = 0; q < p->num_quals; ++q) {
    if ((p->quals[q].after_joins != 0) == (stage != 0) && eval_qual(c, &p->quals[q], err) != 1) {
      return 0;
    }
  }
  return 1;
}

static inline size_t columnar_keys_bytes(const hdk_hip_plan* p, uint32_t entry_count) {
  if (p->query_kind == HDK_Q_PROJECTION) { /* row-position column (QueryMemoryDescriptor.cpp:457-478) */
    return align_to_int64_sz((size_t)entry_count * 8);
  }
  if (p->keyless) {
    return 0;
  }
  return (size_t)p->key_count * align_to_int64_sz((size_t)entry_count * (size_t)p->key_width);
}

/* byte offset of linear slot `slot` in a columnar buffer of `entry_count` entries
 * (RS/QueryMemoryDescriptor.cpp getColOffInBytes: key columns then 8-aligned slot columns) */
static size_t columnar_slot_off(const hdk_hip_plan* p, uint32_t entry_count, int slot) {
  size_t off = columnar_keys_bytes(p, entry_count);
  int s = 0;
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target* tg = &p->targets[t];
    off = align_to_int64_sz(off);
    if (s == slot) {
      return off;
    }
    off += (size_t)entry_count * (size_t)tg->slot_width;
    ++s;
    if (tg->agg == HDK_AGG_AVG) {
      off = align_to_int64_sz(off);
      if (s == slot) {
        return off;
      }
      off += (size_t)entry_count * (size_t)tg->slot2_width;
      ++s;
    }
  }
  return off;
}

static inline int first_slot_of_target(const hdk_hip_plan* p, int t) {
  int s = 0;
  for (int u = 0; u < t; ++u) {
    s += p->targets[u].agg == HDK_AGG_AVG ? 2 : 1;
  }
  return s;
}

/* 8- and 16-bit MIN / MAX slots (agg_{min,max}_int{8,16}[_skip_val], QE/RuntimeFunctions.cpp:540-560,670-704; used by
 * reduceOneSlot through AGGREGATE_ONE_NULLABLE_VALUE_SMALL, QE/ResultSetReduction.cpp:1136-1172 -- "8/16-bit kMin and
 * kMax only").  `w` = 1 or 2. */
static void small_min_max(int agg, int skip, int w, int8_t* slot, int64_t val, int64_t nullv) {
  if (w == 2) {
    int16_t* s = (int16_t*)slot;
    const int16_t v = (int16_t)val, n = (int16_t)nullv;
    if (skip && v == n) return;              /* DEF_SKIP_AGG: val != skip_val */
    if (skip && *s == n) { *s = v; return; } /* ... first value replaces the sentinel */
    *s = agg == HDK_AGG_MIN ? (*s < v ? *s : v) : (*s > v ? *s : v);
  } else {
    int8_t* s = slot;
    const int8_t v = (int8_t)val, n = (int8_t)nullv;
    if (skip && v == n) return;
    if (skip && *s == n) { *s = v; return; }
    *s = agg == HDK_AGG_MIN ? (*s < v ? *s : v) : (*s > v ? *s : v);
  }
}

/* checked_single_agg_id and its typed forms: QE/RuntimeFunctions.cpp:489-506 (int64), :567-583 (DEF_CHECKED_SINGLE_AGG_ID_INT,
 * here n = 32), :743-760 (double: NULL test by value, slot compare by bit pattern), :799-816 (float).  0, or 15 =
 * Executor::ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES. */
int32_t orc_checked_single_agg_id(int64_t* agg, int64_t val, int64_t null_val) {
  if (val == null_val) return 0;
  if (*agg == val) return 0;
  if (*agg == null_val) {
    *agg = val;
    return 0;
  }
  return 15;
}
int32_t orc_checked_single_agg_id_int32(int32_t* agg, int32_t val, int32_t null_val) {
  if (val == null_val) return 0;
  if (*agg == val) return 0;
  if (*agg == null_val) {
    *agg = val;
    return 0;
  }
  return 15;
}
int32_t orc_checked_single_agg_id_double(int64_t* agg, double val, double null_val) {
  if (val == null_val) return 0;
  int64_t vbits, nbits;
  memcpy(&vbits, &val, 8);
  memcpy(&nbits, &null_val, 8);
  if (*agg == vbits) return 0;
  if (*agg == nbits) {
    *agg = vbits;
    return 0;
  }
  return 15;
}
int32_t orc_checked_single_agg_id_float(int32_t* agg, float val, float null_val) {
  if (val == null_val) return 0;
  int32_t vbits, nbits;
  memcpy(&vbits, &val, 4);
  memcpy(&nbits, &null_val, 4);
  if (*agg == vbits) return 0;
  if (*agg == nbits) {
    *agg = vbits;
    return 0;
  }
  return 15;
}

/* Apply one target's aggregate(s) to its slot(s): the call the JIT emits per target
 * (QE/TargetExprBuilder.cpp:341-460: name = agg_<kind>[_int32|_double|_float][_skip_val]). */
static int32_t apply_target_single_value(const hdk_hip_target* tg, int8_t* slot1, int64_t val) {
  const int64_t nullv = tg->null_val;
  if (tg->arg_is_fp == HDK_FP_SLOT_FLOAT) {
    return orc_checked_single_agg_id_float((int32_t*)slot1, (float)bits_to_double(val), (float)bits_to_double(nullv));
  }
  if (tg->arg_is_fp) {
    return orc_checked_single_agg_id_double((int64_t*)slot1, bits_to_double(val), bits_to_double(nullv));
  }
  if (tg->slot_width == 4) {
    return orc_checked_single_agg_id_int32((int32_t*)slot1, (int32_t)val, (int32_t)nullv);
  }
  return orc_checked_single_agg_id((int64_t*)slot1, val, nullv);
}

static void apply_target(const hdk_hip_target* tg, int8_t* slot1, int8_t* slot2, int64_t val) {
  const int skip = tg->skip_null;
  const int64_t nullv = tg->null_val;
  if (tg->agg == HDK_AGG_ID) { /* agg_id / agg_id_int32: RuntimeFunctions.cpp:473-476, 562-571 */
    if (tg->slot_width == 0) return; /* projected key of a baseline table: no slot (target_groupby_indices) */
    if (tg->slot_width == 4) *(int32_t*)slot1 = (int32_t)val;
    else if (tg->slot_width == 2) *(int16_t*)slot1 = (int16_t)val;
    else if (tg->slot_width == 1) *(int8_t*)slot1 = (int8_t)val;
    else *(int64_t*)slot1 = val;
    return;
  }
  /* the count part of COUNT / AVG */
  int8_t* count_slot = tg->agg == HDK_AGG_COUNT ? slot1 : (tg->agg == HDK_AGG_AVG ? slot2 : NULL);
  const int count_w = tg->agg == HDK_AGG_COUNT ? tg->slot_width : tg->slot2_width;
  if (count_slot) {
    if (!tg->has_arg || !skip) {
      if (count_w == 4) orc_agg_count_int32((uint32_t*)count_slot, 0);
      else orc_agg_count((uint64_t*)count_slot, 0);
    } else if (tg->arg_is_fp) {
      /* agg_count_double_skip_val: value compare */
      if (count_w == 4) {
        if (bits_to_double(val) != bits_to_double(nullv)) orc_agg_count_int32((uint32_t*)count_slot, 0);
      } else {
        orc_agg_count_double_skip_val((uint64_t*)count_slot, bits_to_double(val), bits_to_double(nullv));
      }
    } else {
      if (count_w == 4) orc_agg_count_int32_skip_val((uint32_t*)count_slot, (int32_t)val, (int32_t)nullv);
      else orc_agg_count_skip_val((uint64_t*)count_slot, val, nullv);
    }
    if (tg->agg == HDK_AGG_COUNT) {
      return;
    }
  }
  const int w = tg->slot_width;
  if (tg->arg_is_fp) {
    const double d = bits_to_double(val);
    const double dn = bits_to_double(nullv);
    if (tg->arg_is_fp == HDK_FP_SLOT_FLOAT) {
      /* takes_float_argument: agg_chosen_bytes = sizeof(float), the value is cast to float and the *_float runtime
       * function works on the slot's low 4 bytes (QE/TargetExprBuilder.cpp:361-384); nullv is the sentinel widened to double */
      const float f = (float)d, fn = (float)dn;
      int32_t* s = (int32_t*)slot1;
      switch (tg->agg) {
        case HDK_AGG_SUM:
        case HDK_AGG_AVG:
          if (skip) orc_agg_sum_float_skip_val(s, f, fn); else orc_agg_sum_float(s, f);
          break;
        case HDK_AGG_MIN:
          if (skip) orc_agg_min_float_skip_val(s, f, fn); else orc_agg_min_float(s, f);
          break;
        case HDK_AGG_MAX:
          if (skip) orc_agg_max_float_skip_val(s, f, fn); else orc_agg_max_float(s, f);
          break;
      }
      return;
    }
    switch (tg->agg) {
      case HDK_AGG_SUM:
      case HDK_AGG_AVG:
        if (skip) orc_agg_sum_double_skip_val((int64_t*)slot1, d, dn);
        else orc_agg_sum_double((int64_t*)slot1, d);
        break;
      case HDK_AGG_MIN:
        if (skip) orc_agg_min_double_skip_val((int64_t*)slot1, d, dn);
        else orc_agg_min_double((int64_t*)slot1, d);
        break;
      case HDK_AGG_MAX:
        if (skip) orc_agg_max_double_skip_val((int64_t*)slot1, d, dn);
        else orc_agg_max_double((int64_t*)slot1, d);
        break;
    }
    return;
  }
  if (w == 1 || w == 2) { /* logical-sized MIN / MAX slots of a columnar buffer */
    small_min_max(tg->agg, skip, w, slot1, val, nullv);
    return;
  }
  if (w == 4) {
    int32_t* s = (int32_t*)slot1;
    const int32_t v = (int32_t)val, n = (int32_t)nullv;
    switch (tg->agg) {
      case HDK_AGG_SUM:
      case HDK_AGG_AVG:
        if (skip) orc_agg_sum_int32_skip_val(s, v, n); else orc_agg_sum_int32(s, v);
        break;
      case HDK_AGG_MIN:
        if (skip) orc_agg_min_int32_skip_val(s, v, n); else orc_agg_min_int32(s, v);
        break;
      case HDK_AGG_MAX:
        if (skip) orc_agg_max_int32_skip_val(s, v, n); else orc_agg_max_int32(s, v);
        break;
    }
    return;
  }
  int64_t* s = (int64_t*)slot1;
  switch (tg->agg) {
    case HDK_AGG_SUM:
    case HDK_AGG_AVG:
      if (skip) orc_agg_sum_skip_val(s, val, nullv); else orc_agg_sum(s, val);
      break;
    case HDK_AGG_MIN:
      if (skip) orc_agg_min_skip_val(s, val, nullv); else orc_agg_min(s, val);
      break;
    case HDK_AGG_MAX:
      if (skip) orc_agg_max_skip_val(s, val, nullv); else orc_agg_max(s, val);
      break;
  }
}

/* Target argument with the arg-type NULL rewritten to the slot-type NULL
 * (RowFuncBuilder::convertNullIfAny, QE/RowFuncBuilder.cpp:803-860). */
static inline int64_t target_arg(const orc_row_ctx* c, const hdk_hip_target* tg, int32_t* err) {
  if (!tg->has_arg) {
    return 0;
  }
  int64_t v = eval_expr(c, &tg->arg, err);
  if (tg->agg == HDK_AGG_ID) {
    return v;
  }
  if (tg->skip_null) {
    const int isnull = tg->arg.vclass == HDK_VC_FP ? is_null_f(v, tg->arg.null_val, tg->arg.nullable)
                                                   : is_null_i(v, tg->arg.null_val, tg->arg.nullable);
    if (isnull) {
      return tg->null_val;
    }
  }
  if (tg->arg_is_fp && tg->arg.vclass != HDK_VC_FP) {
    return double_to_bits((double)v); /* Executor::castToFP */
  }
  return v;
}

/* ---- join probes -------------------------------------------------------------------------------
 * The matching set of one join level for the current outer row (HashJoin::codegenMatchingSet,
 * QE/JoinHashTable/HashJoin.cpp:149-197; BaselineJoinHashTable::codegenMatchingSet / codegenSlot,
 * QE/JoinHashTable/BaselineJoinHashTable.cpp:769-811).  `*single` receives the row id of a
 * one-to-one match; one-to-many sets are returned as a pointer into the table's row-id section. */
static inline const void* join_table_ptr(const hdk_hip_plan* p, const hdk_hip_join* jn,
                                         const int64_t* join_hash_tables) {
  return (const void*)(intptr_t)(p->num_joins == 1 && jn->table_idx == 0 ? (int64_t)(intptr_t)join_hash_tables
                                                                          : join_hash_tables[jn->table_idx]);
}

/* PerfectJoinHashTable::codegenSlot picks the function by name: "bucketized_" for a DATE key, then "_bitwise" or
 * "_nullable" (QE/JoinHashTable/PerfectJoinHashTable.cpp:1018-1031) */
static int64_t perfect_probe(const hdk_hip_join* jn, const int32_t* table, int64_t key) {
  if (jn->bucket > 1) {
    if (jn->null_mode == HDK_JOIN_NULL_BITWISE) {
      return orc_bucketized_hash_join_idx_bitwise(table, key, jn->min_key, jn->max_key, jn->null_val,
                                                  jn->translated_null, jn->bucket);
    }
    if (jn->null_mode == HDK_JOIN_NULL_NULLABLE) {
      return orc_bucketized_hash_join_idx_nullable(table, key, jn->min_key, jn->max_key, jn->null_val, jn->bucket);
    }
    return orc_bucketized_hash_join_idx(table, key, jn->min_key, jn->max_key, jn->bucket);
  }
  if (jn->null_mode == HDK_JOIN_NULL_NULLABLE) {
    return orc_hash_join_idx_nullable(table, key, jn->min_key, jn->max_key, jn->null_val);
  }
  if (jn->null_mode == HDK_JOIN_NULL_BITWISE) {
    return orc_hash_join_idx_bitwise(table, key, jn->min_key, jn->max_key, jn->null_val, jn->translated_null);
  }
  return orc_hash_join_idx(table, key, jn->min_key, jn->max_key);
}

static int64_t matching_set(const orc_row_ctx* c, const hdk_hip_join* jn, const int64_t* join_hash_tables,
                            const int32_t** ids, int64_t* single, int32_t* err) {
  const hdk_hip_plan* p = c->plan;
  const void* table = join_table_ptr(p, jn, join_hash_tables);
  *ids = NULL;
  *single = -1;
  if (jn->kind == HDK_JOIN_KEYED_ONE_TO_ONE || jn->kind == HDK_JOIN_KEYED_ONE_TO_MANY) {
    /* codegenKey: the outer-side components packed at the table's component width */
    int64_t k64[HDK_HIP_MAX_JOIN_KEYS];
    int32_t k32[HDK_HIP_MAX_JOIN_KEYS];
    const int kc = jn->key_component_count;
    for (int i = 0; i < kc; ++i) {
      const int64_t v = eval_expr(c, i == 0 ? &jn->outer_key : &jn->extra_keys[i - 1], err);
      k64[i] = v;
      k32[i] = (int32_t)v;
    }
    const int w = jn->key_component_width;
    const void* key = w == 4 ? (const void*)k32 : (const void*)k64;
    if (jn->kind == HDK_JOIN_KEYED_ONE_TO_ONE) {
      const int64_t idx = w == 4 ? orc_baseline_hash_join_idx_32((const int8_t*)table, (const int8_t*)key,
                                                                 (size_t)kc * 4, (size_t)jn->entry_count)
                                 : orc_baseline_hash_join_idx_64((const int8_t*)table, (const int8_t*)key,
                                                                 (size_t)kc * 8, (size_t)jn->entry_count);
      *single = idx;
      return idx >= 0 ? 1 : 0;
    }
    const int64_t slot = w == 4 ? orc_get_composite_key_index_32(k32, (size_t)kc, (const int32_t*)table,
                                                                 (size_t)jn->entry_count)
                                : orc_get_composite_key_index_64(k64, (size_t)kc, (const int64_t*)table,
                                                                 (size_t)jn->entry_count);
    if (slot < 0) {
      return 0;
    }
    const int32_t* otm = (const int32_t*)((const int8_t*)table + (size_t)jn->entry_count * kc * w);
    const int64_t pos = orc_hash_join_idx(otm, slot, 0, jn->entry_count - 1);
    if (pos < 0) {
      return 0;
    }
    *ids = otm + 2 * jn->entry_count + pos;
    return orc_hash_join_idx(otm + jn->entry_count, slot, 0, jn->entry_count - 1);
  }
  const int64_t key = eval_expr(c, &jn->outer_key, err);
  if (jn->kind == HDK_JOIN_ONE_TO_MANY) {
    const int32_t* t = (const int32_t*)table;
    const int64_t pos = perfect_probe(jn, t, key);
    if (pos < 0) {
      return 0;
    }
    *ids = t + 2 * jn->entry_count + pos;
    return perfect_probe(jn, t + jn->entry_count, key);
  }
  const int64_t idx = perfect_probe(jn, (const int32_t*)table, key);
  *single = idx;
  return idx >= 0 ? 1 : 0;
}

/* The join loop nest of one outer row (Executor::buildJoinLoops, QE/IRCodegen.cpp:497-667):
 * filters on the outer table first, then one loop level per join (Singleton or Set; a LEFT join
 * without a match continues once with the inner row "not found" = -1, whose columns read as NULL),
 * then the filters that read joined columns, then `body`. */
typedef int32_t (*orc_row_body)(orc_row_ctx* c, void* arg, int32_t* err);

static int32_t join_level(orc_row_ctx* c, const int64_t* join_hash_tables, int level, orc_row_body body,
                          void* arg, int32_t* err) {
  const hdk_hip_plan* p = c->plan;
  if (level == p->num_joins) {
    if (!filter_pass(c, 1, err)) {
      return 0;
    }
    return body(c, arg, err);
  }
  const hdk_hip_join* jn = &p->joins[level];
  const int32_t* ids;
  int64_t single;
  const int64_t n = matching_set(c, jn, join_hash_tables, &ids, &single, err);
  if (jn->type == HDK_JOIN_ANTI) {
    /* JoinLoop.cpp:258-262: the body runs when slot_lookup_result < 0 */
    if (n > 0) {
      return 0;
    }
    c->join_row[level] = -1;
    return join_level(c, join_hash_tables, level + 1, body, arg, err);
  }
  if (n <= 0) {
    if (jn->type != HDK_JOIN_LEFT) { /* INNER, SEMI (:254-257) */
      return 0;
    }
    c->join_row[level] = -1;
    return join_level(c, join_hash_tables, level + 1, body, arg, err);
  }
  for (int64_t m = 0; m < n; ++m) {
    c->join_row[level] = ids ? ids[m] : single;
    const int32_t rc = join_level(c, join_hash_tables, level + 1, body, arg, err);
    if (rc) {
      return rc;
    }
  }
  return 0;
}

/* returns HDK_HIP_ERR_OUT_OF_SLOTS / a projection's -pos when the scan must stop, else 0;
 * `*err` collects the row's soft error (division by zero) */
static int32_t drive_row(orc_row_ctx* c, const int64_t* join_hash_tables, orc_row_body body, void* arg,
                         int32_t* err) {
  if (!filter_pass(c, 0, err)) {
    return 0;
  }
  return join_level(c, join_hash_tables, 0, body, arg, err);
}

static int32_t aggregate_row(orc_row_ctx* c, void* arg, int32_t* errp) {
  const hdk_hip_plan* p = c->plan;
  int64_t* out = (int64_t*)arg;
  int32_t err = 0;
  /* group lookup */
  int8_t* row_base = NULL; /* row-wise: start of the entry's slots region base (row start) */
  uint32_t entry = 0;
  if (p->query_kind == HDK_Q_NON_GROUPED) {
    /* out[s] slots */
  } else {
    int64_t key[HDK_HIP_MAX_KEYS];
    for (int k = 0; k < p->key_count; ++k) {
      int64_t kv = eval_expr(c, &p->keys[k], &err);
      if (p->query_kind == HDK_Q_PERFECT_HASH && p->key_has_nulls[k] && p->keys[k].nullable &&
          kv == p->keys[k].null_val) {
        kv = p->key_null_translated[k]; /* translate_null_key_*, GroupByRuntime.cpp:368-382 */
      }
      key[k] = kv;
    }
    if (p->query_kind == HDK_Q_PERFECT_HASH) {
      if (p->key_count == 1) {
        if (p->output_columnar) {
          if (p->keyless) {
            int64_t off = key[0] - p->key_min[0];
            if (p->key_bucket[0]) off /= p->key_bucket[0];
            entry = (uint32_t)off;
          } else {
            entry = orc_get_columnar_group_bin_offset(out, key[0], p->key_min[0], p->key_bucket[0]);
          }
        } else if (p->keyless) {
          int64_t* s = orc_get_group_value_fast_keyless(out, key[0], p->key_min[0], p->key_bucket[0],
                                                        p->row_size_quad);
          entry = (uint32_t)((s - out) / p->row_size_quad);
        } else {
          int64_t* s = orc_get_group_value_fast(out, key[0], p->key_min[0], p->key_bucket[0],
                                                p->row_size_quad);
          entry = (uint32_t)((s - 1 - out) / p->row_size_quad);
        }
      } else {
        /* perfect_key_hash: QE/RowFuncBuilder.cpp:748-801 */
        int64_t h = 0;
        int64_t stride = 1;
        for (int k = 0; k < p->key_count; ++k) {
          int64_t term = key[k] - p->key_min[k];
          if (p->key_bucket[k]) term /= p->key_bucket[k];
          h += term * stride;
          stride *= p->key_card[k];
        }
        entry = (uint32_t)h;
        if (p->output_columnar) {
          if (!p->keyless) {
            orc_set_matching_group_value_perfect_hash_columnar(out, entry, key, (uint32_t)p->key_count,
                                                               p->entry_count);
          }
        } else if (!p->keyless) {
          orc_get_matching_group_value_perfect_hash(out, entry, key, (uint32_t)p->key_count,
                                                    p->row_size_quad);
        }
      }
    } else { /* baseline hash */
      int64_t packed[HDK_HIP_MAX_KEYS];
      const int64_t* kp = key;
      if (p->key_width == 4) {
        int32_t* k32 = (int32_t*)packed;
        for (int k = 0; k < p->key_count; ++k) k32[k] = (int32_t)key[k];
        kp = packed;
      }
      if (p->output_columnar) {
        const int32_t s = orc_get_group_value_columnar_slot(out, p->entry_count, kp,
                                                            (uint32_t)p->key_count, (uint32_t)p->key_width);
        if (s < 0) {
          return HDK_HIP_ERR_OUT_OF_SLOTS;
        }
        entry = (uint32_t)s;
      } else {
        int64_t* s = orc_get_group_value(out, p->entry_count, kp, (uint32_t)p->key_count,
                                         (uint32_t)p->key_width, p->row_size_quad);
        if (!s) {
          return HDK_HIP_ERR_OUT_OF_SLOTS;
        }
        const size_t key_quads = align_to_int64_sz((size_t)p->key_count * p->key_width) / 8;
        entry = (uint32_t)((s - key_quads - out) / p->row_size_quad);
      }
    }
    row_base = (int8_t*)(out + (size_t)entry * p->row_size_quad);
  }
  /* aggregates */
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target* tg = &p->targets[t];
    const int64_t v = target_arg(c, tg, &err);
    int8_t *s1, *s2 = NULL;
    if (p->query_kind == HDK_Q_NON_GROUPED) {
      const int fs = first_slot_of_target(p, t);
      s1 = (int8_t*)&out[fs];
      s2 = (int8_t*)&out[fs + 1];
    } else if (p->output_columnar) {
      const int fs = first_slot_of_target(p, t);
      s1 = (int8_t*)out + columnar_slot_off(p, p->entry_count, fs) + (size_t)entry * tg->slot_width;
      if (tg->agg == HDK_AGG_AVG) {
        s2 = (int8_t*)out + columnar_slot_off(p, p->entry_count, fs + 1) +
             (size_t)entry * tg->slot2_width;
      }
    } else {
      s1 = row_base + tg->slot_off;
      s2 = row_base + tg->slot2_off;
    }
    if (tg->agg == HDK_AGG_SINGLE_VALUE) { /* the row function returns the call's code (QE/RowFuncBuilder.cpp) */
      const int32_t e = apply_target_single_value(tg, s1, v);
      if (e && !err) err = e;
      continue;
    }
    apply_target(tg, s1, s2, v);
  }
  if (err && !*errp) {
    *errp = err;
  }
  return 0;
}

int32_t orc_run_plan_range(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                           uint64_t frag_begin, uint64_t frag_end, const int64_t* num_rows,
                           uint32_t num_tables, const int64_t* join_hash_tables, int64_t* out) {
  int32_t first_err = 0;
  orc_row_ctx c;
  memset(&c, 0, sizeof(c));
  c.plan = plan;
  for (uint64_t f = frag_begin; f < frag_end; ++f) { /* multifrag_query: RuntimeFunctions.cpp:1741-1768 */
    c.cols = col_buffers[f];
    const int64_t n = num_rows[f * num_tables];
    for (int64_t pos = 0; pos < n; ++pos) { /* query_group_by_template: pos_start=0, pos_step=1 */
      c.pos = pos;
      int32_t err = 0;
      const int32_t rc = drive_row(&c, join_hash_tables, aggregate_row, out, &err);
      if (rc) {
        return rc; /* out of slots stops the scan */
      }
      if (err && !first_err) {
        first_err = err; /* record_error_code: first error sticks (RuntimeFunctions.cpp:1123-1135) */
      }
    }
  }
  return first_err;
}

int32_t orc_run_plan(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                     uint64_t num_fragments, const int64_t* num_rows, uint32_t num_tables,
                     const int64_t* join_hash_tables, int64_t* out) {
  return orc_run_plan_range(plan, col_buffers, 0, num_fragments, num_rows, num_tables,
                            join_hash_tables, out);
}

/* ============================================================================================
 * Projection (filter/project): QE/RowFuncBuilder.cpp:162-215 claims an output row with
 * old = (*total_matched)++ and get_scan_output_slot / get_columnar_scan_output_offset
 * (QE/GroupByRuntime.cpp:248-272) writes the row position and returns the slot base; targets are
 * stored with agg_id.  Rows beyond max_matched report "out of slots" as -pos.
 * ========================================================================================== */
typedef struct {
  int64_t* out;
  int32_t max_matched;
  int32_t* total_matched;
  int32_t slots_err;
} orc_proj_arg;

static int32_t project_row(orc_row_ctx* c, void* arg, int32_t* err) {
  const hdk_hip_plan* p = c->plan;
  orc_proj_arg* a = (orc_proj_arg*)arg;
  int64_t* out = a->out;
  const uint32_t slot = (uint32_t)(*a->total_matched)++;
  if (slot >= (uint32_t)a->max_matched) {
    if (!a->slots_err) a->slots_err = -(int32_t)c->pos; /* out of slots: -pos (RowFuncBuilder.cpp:268-273) */
    return 0;
  }
  int8_t* row_base = NULL;
  if (p->output_columnar) {
    out[slot] = c->pos; /* get_columnar_scan_output_offset */
  } else {
    int64_t* rp = out + (size_t)slot * p->row_size_quad; /* get_scan_output_slot */
    rp[0] = c->pos;
    row_base = (int8_t*)rp;
  }
  for (int t = 0; t < p->num_targets; ++t) {
    const hdk_hip_target* tg = &p->targets[t];
    int64_t v = eval_expr(c, &tg->arg, err);
    int8_t* s1 = p->output_columnar
                     ? (int8_t*)out + columnar_slot_off(p, p->entry_count, t) + (size_t)slot * tg->slot_width
                     : row_base + tg->slot_off;
    apply_target(tg, s1, NULL, v);
  }
  return 0;
}

int32_t orc_run_projection(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                           uint64_t num_fragments, const int64_t* num_rows, uint32_t num_tables,
                           const int64_t* join_hash_tables, int64_t* out, int32_t max_matched,
                           int32_t* total_matched) {
  int32_t first_err = 0;
  orc_row_ctx c;
  memset(&c, 0, sizeof(c));
  c.plan = plan;
  orc_proj_arg a = {out, max_matched, total_matched, 0};
  for (uint64_t f = 0; f < num_fragments; ++f) {
    c.cols = col_buffers[f];
    const int64_t n = num_rows[f * num_tables];
    for (int64_t pos = 0; pos < n; ++pos) {
      c.pos = pos;
      int32_t err = 0;
      drive_row(&c, join_hash_tables, project_row, &a, &err);
      if (a.slots_err && !first_err) first_err = a.slots_err;
      if (err && !first_err) first_err = err;
    }
  }
  return first_err;
}

/* ============================================================================================
 * Reduction
 * ========================================================================================== */

static int64_t read_int_slot(const int8_t* p, int w) {
  if (w == 4) {
    int32_t v;
    memcpy(&v, p, 4);
    return v;
  }
  int64_t v;
  memcpy(&v, p, 8);
  return v;
}

/* RS/ResultSetStorage.cpp:439-521. */
int32_t orc_is_empty_entry(const hdk_hip_plan* p, const int64_t* buf, uint32_t entry_count,
                           uint32_t idx, const int64_t* init_vals) {
  if (p->query_kind == HDK_Q_NON_GROUPED) {
    return 0;
  }
  if (p->keyless) {
    /* keyless: the designated target (idx_target_as_key) still holds its init value. */
    const int ks = p->idx_target_as_key; /* a SLOT index */
    int kt = 0, second = 0;
    for (int t = 0, sidx = 0; t < p->num_targets; ++t) {
      const int n = p->targets[t].agg == HDK_AGG_AVG ? 2 : 1;
      if (ks >= sidx && ks < sidx + n) {
        kt = t;
        second = ks - sidx;
        break;
      }
      sidx += n;
    }
    const hdk_hip_target* tg = &p->targets[kt];
    const int w = second ? tg->slot2_width : tg->slot_width;
    const int8_t* s = p->output_columnar
                          ? (const int8_t*)buf + columnar_slot_off(p, entry_count, ks) + (size_t)idx * w
                          : (const int8_t*)(buf + (size_t)idx * p->row_size_quad) +
                                (second ? tg->slot2_off : tg->slot_off);
    int64_t iv = init_vals[ks];
    if (w == 4) iv = (int32_t)iv;
    return read_int_slot(s, w) == iv;
  }
  if (p->output_columnar) {
    if (p->key_width == 4) {
      return ((const int32_t*)buf)[idx] == HDK_EMPTY_KEY_32;
    }
    return buf[idx] == HDK_EMPTY_KEY_64;
  }
  const int64_t* keys_ptr = buf + (size_t)idx * p->row_size_quad;
  if (p->key_width == 4) {
    return *(const int32_t*)keys_ptr == HDK_EMPTY_KEY_32;
  }
  return *keys_ptr == HDK_EMPTY_KEY_64;
}

/* reduceOneSlot: QE/ResultSetReduction.cpp:1234-1330 with AGGREGATE_ONE_* :1026-1107.
 * `init_val` is the slot's init value == skip value for nullable targets. */
/* reduceOneSlotSingleValue (QE/ResultSetReduction.cpp:1186-1230): by slot width, against the slot's init value; the
 * reference throws "Multiple distinct values encountered", here the code the device path reports */
static int32_t reduce_single_value(const hdk_hip_target* tg, int8_t* this1, const int8_t* that1, int64_t init_val) {
  if (tg->slot_width == 4) {
    const int32_t l = *(const int32_t*)this1, r = *(const int32_t*)that1;
    if (r == (int32_t)init_val) return 0;
    if (l == (int32_t)init_val) {
      *(int32_t*)this1 = r;
      return 0;
    }
    return l != r ? 15 : 0;
  }
  const int64_t l = *(const int64_t*)this1, r = *(const int64_t*)that1;
  if (r == init_val) return 0;
  if (l == init_val) {
    *(int64_t*)this1 = r;
    return 0;
  }
  return l != r ? 15 : 0;
}

static void reduce_one_target(const hdk_hip_target* tg, int8_t* this1, int8_t* this2,
                              const int8_t* that1, const int8_t* that2, int64_t init_val) {
  if (tg->agg == HDK_AGG_ID) { /* non-agg projection: ResultSetReduction.cpp:1329-1385 */
    if (tg->slot_width == 0) { /* getTargetGroupbyIndex >= 0: nothing to reduce (:1248-1251) */
      return;
    }
    if (tg->slot_width == 2) {
      const int16_t rhs = *(const int16_t*)that1;
      if (rhs != (int16_t)init_val) *(int16_t*)this1 = rhs;
    } else if (tg->slot_width == 1) {
      if (*that1 != (int8_t)init_val) *this1 = *that1;
    } else if (tg->slot_width == 4) {
      const int32_t rhs = *(const int32_t*)that1;
      if (rhs != init_val) *(int32_t*)this1 = rhs;
    } else {
      const int64_t rhs = *(const int64_t*)that1;
      if (rhs != init_val) *(int64_t*)this1 = rhs;
    }
    return;
  }
  if (tg->agg == HDK_AGG_COUNT) { /* AGGREGATE_ONE_COUNT */
    if (tg->slot_width == 4) orc_agg_sum_int32((int32_t*)this1, *(const int32_t*)that1);
    else orc_agg_sum((int64_t*)this1, *(const int64_t*)that1);
    return;
  }
  if (tg->agg == HDK_AGG_AVG) {
    if (tg->slot2_width == 4) orc_agg_sum_int32((int32_t*)this2, *(const int32_t*)that2);
    else orc_agg_sum((int64_t*)this2, *(const int64_t*)that2);
  }
  const int w = tg->slot_width;
  const int skip = tg->skip_null;
  if (tg->arg_is_fp) {
    if (tg->arg_is_fp == HDK_FP_SLOT_FLOAT) { /* get_width_for_slot -> sizeof(float), ResultSetReduction.cpp:1176-1185 */
      const float o = bits_to_float(*(const int32_t*)that1);
      const float nv = bits_to_float((int32_t)init_val);
      int32_t* s = (int32_t*)this1;
      switch (tg->agg) {
        case HDK_AGG_SUM:
        case HDK_AGG_AVG:
          if (skip) orc_agg_sum_float_skip_val(s, o, nv); else orc_agg_sum_float(s, o);
          break;
        case HDK_AGG_MIN:
          if (skip) orc_agg_min_float_skip_val(s, o, nv); else orc_agg_min_float(s, o);
          break;
        case HDK_AGG_MAX:
          if (skip) orc_agg_max_float_skip_val(s, o, nv); else orc_agg_max_float(s, o);
          break;
      }
      return;
    }
    const double o = bits_to_double(*(const int64_t*)that1);
    const double nv = bits_to_double(init_val);
    switch (tg->agg) {
      case HDK_AGG_SUM:
      case HDK_AGG_AVG:
        if (skip) orc_agg_sum_double_skip_val((int64_t*)this1, o, nv);
        else orc_agg_sum_double((int64_t*)this1, o);
        break;
      case HDK_AGG_MIN:
        if (skip) orc_agg_min_double_skip_val((int64_t*)this1, o, nv);
        else orc_agg_min_double((int64_t*)this1, o);
        break;
      case HDK_AGG_MAX:
        if (skip) orc_agg_max_double_skip_val((int64_t*)this1, o, nv);
        else orc_agg_max_double((int64_t*)this1, o);
        break;
    }
    return;
  }
  if (w == 1 || w == 2) {
    small_min_max(tg->agg, skip, w, this1, w == 2 ? (int64_t)*(const int16_t*)that1 : (int64_t)*that1, init_val);
    return;
  }
  if (w == 4) {
    const int32_t o = *(const int32_t*)that1;
    const int32_t nv = (int32_t)init_val;
    switch (tg->agg) {
      case HDK_AGG_SUM:
      case HDK_AGG_AVG:
        if (skip) orc_agg_sum_int32_skip_val((int32_t*)this1, o, nv);
        else orc_agg_sum_int32((int32_t*)this1, o);
        break;
      case HDK_AGG_MIN:
        if (skip) orc_agg_min_int32_skip_val((int32_t*)this1, o, nv);
        else orc_agg_min_int32((int32_t*)this1, o);
        break;
      case HDK_AGG_MAX:
        if (skip) orc_agg_max_int32_skip_val((int32_t*)this1, o, nv);
        else orc_agg_max_int32((int32_t*)this1, o);
        break;
    }
    return;
  }
  const int64_t o = *(const int64_t*)that1;
  switch (tg->agg) {
    case HDK_AGG_SUM:
    case HDK_AGG_AVG:
      if (skip) orc_agg_sum_skip_val((int64_t*)this1, o, init_val);
      else orc_agg_sum((int64_t*)this1, o);
      break;
    case HDK_AGG_MIN:
      if (skip) orc_agg_min_skip_val((int64_t*)this1, o, init_val);
      else orc_agg_min((int64_t*)this1, o);
      break;
    case HDK_AGG_MAX:
      if (skip) orc_agg_max_skip_val((int64_t*)this1, o, init_val);
      else orc_agg_max((int64_t*)this1, o);
      break;
  }
}

static void slot_ptrs(const hdk_hip_plan* p, const int64_t* buf, uint32_t entry_count, uint32_t entry,
                      int t, int8_t** s1, int8_t** s2) {
  const hdk_hip_target* tg = &p->targets[t];
  const int fs = first_slot_of_target(p, t);
  if (p->query_kind == HDK_Q_NON_GROUPED) {
    *s1 = (int8_t*)&buf[fs];
    *s2 = (int8_t*)&buf[fs + 1];
  } else if (p->output_columnar) {
    *s1 = (int8_t*)buf + columnar_slot_off(p, entry_count, fs) + (size_t)entry * tg->slot_width;
    *s2 = tg->agg == HDK_AGG_AVG ? (int8_t*)buf + columnar_slot_off(p, entry_count, fs + 1) +
                                       (size_t)entry * tg->slot2_width
                                 : NULL;
  } else {
    int8_t* row = (int8_t*)(buf + (size_t)entry * p->row_size_quad);
    *s1 = row + tg->slot_off;
    *s2 = row + tg->slot2_off;
  }
}

int32_t orc_reduce(const hdk_hip_plan* p, int64_t* this_buf, uint32_t this_entry_count,
                   const int64_t* that_buf, uint32_t that_entry_count, const int64_t* init_vals) {
  int32_t err = 0; /* SINGLE_VALUE conflicts: every entry is still reduced, the code is returned at the end */
  if (p->query_kind == HDK_Q_NON_GROUPED || p->query_kind == HDK_Q_PERFECT_HASH) {
    /* reduceOneEntryNoCollisions / reduceEntriesNoCollisionsColWise (:262-330, :353-449) */
    const uint32_t n = p->query_kind == HDK_Q_NON_GROUPED ? 1 : this_entry_count;
    for (uint32_t e = 0; e < n; ++e) {
      if (orc_is_empty_entry(p, that_buf, that_entry_count, e, init_vals)) {
        continue;
      }
      if (p->query_kind == HDK_Q_PERFECT_HASH && !p->keyless) { /* copy the key from that */
        if (p->output_columnar) {
          for (int k = 0; k < p->key_count; ++k) {
            const size_t koff = (size_t)k * align_to_int64_sz((size_t)this_entry_count * 8) / 8;
            this_buf[koff + e] = that_buf[koff + e];
          }
        } else {
          memcpy(this_buf + (size_t)e * p->row_size_quad, that_buf + (size_t)e * p->row_size_quad,
                 (size_t)p->key_count * 8);
        }
      }
      int iv = 0;
      for (int t = 0; t < p->num_targets; ++t) {
        int8_t *a1, *a2, *b1, *b2;
        slot_ptrs(p, this_buf, this_entry_count, e, t, &a1, &a2);
        slot_ptrs(p, that_buf, that_entry_count, e, t, &b1, &b2);
        if (p->targets[t].agg == HDK_AGG_SINGLE_VALUE) {
          const int32_t e1 = reduce_single_value(&p->targets[t], a1, b1, init_vals[iv]);
          if (e1 && !err) err = e1;
        } else {
          reduce_one_target(&p->targets[t], a1, a2, b1, b2, init_vals[iv]);
        }
        iv += p->targets[t].agg == HDK_AGG_AVG ? 2 : 1;
      }
    }
    return err;
  }
  /* baseline: re-insert every non-empty entry of `that` (reduceOneEntryBaseline :694-731) */
  for (uint32_t e = 0; e < that_entry_count; ++e) {
    if (orc_is_empty_entry(p, that_buf, that_entry_count, e, init_vals)) {
      continue;
    }
    int64_t key[HDK_HIP_MAX_KEYS];
    uint32_t dst;
    int fresh;
    if (p->output_columnar) {
      if (p->key_width == 4) {
        int32_t* k32 = (int32_t*)key;
        for (int k = 0; k < p->key_count; ++k) {
          const size_t kcol = (size_t)k * align_to_int64_sz((size_t)that_entry_count * 4);
          k32[k] = *(const int32_t*)((const int8_t*)that_buf + kcol + (size_t)e * 4);
        }
      } else {
        for (int k = 0; k < p->key_count; ++k) {
          key[k] = that_buf[(size_t)k * that_entry_count + e];
        }
      }
      /* was the slot empty before the lookup? */
      const uint32_t h0 = orc_key_hash(key, (uint32_t)p->key_count, (uint32_t)p->key_width) % this_entry_count;
      (void)h0;
      /* find / claim */
      uint32_t h = orc_key_hash(key, (uint32_t)p->key_count, (uint32_t)p->key_width) % this_entry_count;
      uint32_t probe = h;
      int32_t found = -1;
      fresh = 0;
      do {
        const int was_empty = p->key_width == 4 ? ((const int32_t*)this_buf)[probe] == HDK_EMPTY_KEY_32
                                                : this_buf[probe] == HDK_EMPTY_KEY_64;
        const int32_t m = orc_get_matching_group_value_columnar_slot(
            this_buf, this_entry_count, probe, key, (uint32_t)p->key_count, (uint32_t)p->key_width);
        if (m != -1) {
          found = (int32_t)probe;
          fresh = was_empty;
          break;
        }
        probe = (probe + 1) % this_entry_count;
      } while (probe != h);
      if (found < 0) {
        return HDK_HIP_ERR_OUT_OF_SLOTS;
      }
      dst = (uint32_t)found;
    } else {
      const int64_t* src_row = that_buf + (size_t)e * p->row_size_quad;
      memcpy(key, src_row, (size_t)p->key_count * p->key_width);
      uint32_t h = orc_key_hash(key, (uint32_t)p->key_count, (uint32_t)p->key_width) % this_entry_count;
      uint32_t probe = h;
      int64_t* slots = NULL;
      fresh = 0;
      do {
        const int64_t* kp = this_buf + (size_t)probe * p->row_size_quad;
        const int was_empty =
            p->key_width == 4 ? *(const int32_t*)kp == HDK_EMPTY_KEY_32 : *kp == HDK_EMPTY_KEY_64;
        slots = orc_get_matching_group_value(this_buf, probe, key, (uint32_t)p->key_count,
                                             (uint32_t)p->key_width, p->row_size_quad);
        if (slots) {
          fresh = was_empty;
          break;
        }
        probe = (probe + 1) % this_entry_count;
      } while (probe != h);
      if (!slots) {
        return HDK_HIP_ERR_OUT_OF_SLOTS;
      }
      dst = probe;
    }
    int iv = 0;
    for (int t = 0; t < p->num_targets; ++t) {
      const hdk_hip_target* tg = &p->targets[t];
      int8_t *a1, *a2, *b1, *b2;
      slot_ptrs(p, this_buf, this_entry_count, dst, t, &a1, &a2);
      slot_ptrs(p, that_buf, that_entry_count, e, t, &b1, &b2);
      if (fresh) { /* fill_slots: plain copy into the new entry */
        memcpy(a1, b1, (size_t)tg->slot_width);
        if (tg->agg == HDK_AGG_AVG) {
          memcpy(a2, b2, (size_t)tg->slot2_width);
        }
      } else if (tg->agg == HDK_AGG_SINGLE_VALUE) {
        const int32_t e1 = reduce_single_value(tg, a1, b1, init_vals[iv]);
        if (e1 && !err) err = e1;
      } else {
        reduce_one_target(tg, a1, a2, b1, b2, init_vals[iv]);
      }
      iv += tg->agg == HDK_AGG_AVG ? 2 : 1;
    }
  }
  return err;
}

/* ============================================================================================
 * HDK-semantics CPU path: kernel per fragment on a thread pool + reduction of the partials
 * (QE/Execute.cpp:2776-2788, :1290-1317).  Used only for the timed cpu_baseline in bench.py.
 * ========================================================================================== */

size_t orc_sizeof_plan(void) { /* guards against a stale build after include/hdk_hip.h changed */
  return sizeof(hdk_hip_plan);
}

int32_t orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int32_t orc_run_plan_parallel(const hdk_hip_plan* plan, const int8_t* const* const* col_buffers,
                              uint64_t num_fragments, const int64_t* num_rows, uint32_t num_tables,
                              const int64_t* join_hash_tables, const int64_t* init_buffer,
                              size_t buffer_quads, const int64_t* init_vals, int32_t num_threads,
                              int64_t* out) {
  if (num_threads < 1) {
    num_threads = 1;
  }
  /* every kernel's private buffer on cache lines of its own (HDK allocates one ResultSet buffer per kernel; buffers laid end to
   * end in one malloc share their border lines, and two threads' read-modify-writes on one line cost a cross-core transfer
   * each: round 6 found the JIT-shaped leg 10 x slower for it) */
  const size_t pstride = (buffer_quads + 31) & ~(size_t)31; /* quads: multiples of 256 bytes */
  int64_t* partials = NULL;
  if (posix_memalign((void**)&partials, 256, (num_fragments ? num_fragments : 1) * pstride * sizeof(int64_t)) != 0) partials = NULL;
  if (!partials) {
    return HDK_HIP_ERR_RUNTIME;
  }
  int32_t err = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
#endif
  for (int64_t f = 0; f < (int64_t)num_fragments; ++f) { /* one ExecutionKernel per fragment */
    int64_t* mine = partials + (size_t)f * pstride;
    memcpy(mine, init_buffer, buffer_quads * sizeof(int64_t));
    const int32_t e = orc_run_plan_range(plan, col_buffers, (uint64_t)f, (uint64_t)f + 1, num_rows,
                                         num_tables, join_hash_tables, mine);
    if (e) {
#ifdef _OPENMP
#pragma omp critical
#endif
      {
        if (!err) err = e;
      }
    }
  }
  memcpy(out, init_buffer, buffer_quads * sizeof(int64_t));
  for (uint64_t f = 0; f < num_fragments && !err; ++f) { /* reduceMultiDeviceResultSets */
    const int32_t e = orc_reduce(plan, out, plan->entry_count, partials + (size_t)f * pstride,
                                 plan->entry_count, init_vals);
    if (e) {
      err = e;
    }
  }
  free(partials);
  return err;
}

/* ---- thread / page placement of the CPU baseline (bench.py: cpu_baseline) ----------------------------------------
 * The host of the GPU box is a multi-socket, many-NUMA-node machine; where the threads run and where their pages sit
 * decides what the row loop gets (round 3 measured 85 GB/s and 450 GB/s for the same code on two boxes).  Placement is
 * made explicit: thread t of T is pinned to the allowed CPU number t * (allowed / T) (spread over sockets and nodes, like
 * OMP_PLACES=cores OMP_PROC_BIND=spread would, but without depending on when libgomp read its environment), fragments
 * live in anonymous mmap'ed memory first touched by the thread that scans them. */
static int allowed_cpus(int* out, int cap) {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) != 0) return 0;
  int n = 0;
  for (int c = 0; c < CPU_SETSIZE && n < cap; ++c) {
    if (CPU_ISSET(c, &set)) out[n++] = c;
  }
  return n;
}

int32_t orc_allowed_cpu_count(void) {
  int cpus[CPU_SETSIZE];
  return allowed_cpus(cpus, CPU_SETSIZE);
}

static void pin_spread(int tid, int nthreads, const int* cpus, int ncpus) {
  if (ncpus <= 0 || nthreads <= 0) return;
  cpu_set_t one;
  CPU_ZERO(&one);
  CPU_SET(cpus[(int)(((long)tid * ncpus) / nthreads) % ncpus], &one);
  (void)sched_setaffinity(0, sizeof(one), &one);
}

static void unpin(const int* cpus, int ncpus) {
  cpu_set_t all;
  CPU_ZERO(&all);
  for (int i = 0; i < ncpus; ++i) CPU_SET(cpus[i], &all);
  (void)sched_setaffinity(0, sizeof(all), &all);
}

static void* map_pages(size_t bytes) {
  void* p = mmap(NULL, bytes ? bytes : 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  return p == MAP_FAILED ? NULL : p;
}

/* What the host's memory gives a kernel-per-thread streaming READ of two 8-byte columns (the access shape of the row
 * loop below, without its dependent read-modify-write): `bytes_per_thread` per column and thread, pages first touched
 * by their thread, best of `reps`; GB/s over all threads.  The figure the CPU baseline is held against. */
double orc_host_stream_read_gbps(int32_t num_threads, size_t bytes_per_thread, int32_t reps) {
  int cpus[CPU_SETSIZE];
  const int ncpus = allowed_cpus(cpus, CPU_SETSIZE);
  if (num_threads < 1) num_threads = 1;
  const size_t n = bytes_per_thread / 8;
  int64_t** a = (int64_t**)calloc((size_t)num_threads, sizeof(int64_t*));
  int64_t** b = (int64_t**)calloc((size_t)num_threads, sizeof(int64_t*));
  int64_t* sink = (int64_t*)calloc((size_t)num_threads, sizeof(int64_t));
  if (!a || !b || !sink) return -1.0;
  int bad = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
#endif
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
    const int T = omp_get_num_threads();
#else
    const int t = 0, T = 1;
#endif
    pin_spread(t, T, cpus, ncpus);
    a[t] = (int64_t*)map_pages(n * 8);
    b[t] = (int64_t*)map_pages(n * 8);
    if (!a[t] || !b[t]) {
      bad = 1;
    } else {
      for (size_t i = 0; i < n; ++i) {
        a[t][i] = (int64_t)i;
        b[t][i] = (int64_t)(i ^ 5);
      }
    }
  }
  double best = -1.0;
  for (int r = 0; r < reps && !bad; ++r) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
#endif
    {
#ifdef _OPENMP
      const int t = omp_get_thread_num();
#else
      const int t = 0;
#endif
      if (a[t] && b[t]) {
        int64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        const int64_t* restrict x = a[t];
        const int64_t* restrict y = b[t];
        size_t i = 0;
        for (; i + 4 <= n; i += 4) {
          s0 += x[i] + y[i];
          s1 += x[i + 1] + y[i + 1];
          s2 += x[i + 2] + y[i + 2];
          s3 += x[i + 3] + y[i + 3];
        }
        sink[t] = s0 + s1 + s2 + s3;
      }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (best < 0 || sec < best) best = sec;
  }
  int64_t keep = 0;
  for (int t = 0; t < num_threads; ++t) {
    keep += sink[t];
    if (a[t]) munmap(a[t], n ? n * 8 : 4096);
    if (b[t]) munmap(b[t], n ? n * 8 : 4096);
  }
  free(a);
  free(b);
  free(sink);
  unpin(cpus, ncpus);
  if (bad || best <= 0) return -1.0;
  (void)keep;
  return 2.0 * (double)n * 8.0 * (double)num_threads / best / 1e9;
}

/* ---- the host's streaming-read figure, placed on purpose (round 6) -----------------------------------------------------
 * orc_host_stream_read_gbps above spreads its threads by CPU NUMBER; on a two-socket EPYC whose numbers run socket 0 cores,
 * socket 1 cores, then their SMT siblings, 128 threads landed two to a core on half the cores and the figure fell 10 x from
 * its 64-thread value (profiles/r05_bench_default.json: 733 / 1 002 / 100 / 44 GB/s at 32 / 64 / 128 / 256 threads) -- placement,
 * not DRAM.  Here thread t gets the t-th entry of a list that holds ONE CPU per physical core first (its lowest-numbered
 * sibling), dealt round robin over the NUMA nodes, and only then the siblings; every thread maps and first-touches its own
 * pages after it is pinned; all threads are timed inside ONE parallel region between two barriers (no fork / join in the
 * figure).  per_node[i] (i < max_nodes): GB/s of the threads on node i.  Returns the total, or a negative number. */
static int read_first_int(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int v = -1;
  if (fscanf(f, "%d", &v) != 1) v = -1;
  fclose(f);
  return v;
}
static int cpu_numa_node(int cpu) {
  char path[128];
  for (int node = 0; node < 64; ++node) {
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/node%d", cpu, node);
    if (access(path, F_OK) == 0) return node;
  }
  return 0;
}
/* allowed CPUs, one per physical core first (round robin over the nodes), then the SMT siblings; *ncores = the first part.
 * Reads sysfs once per CPU (callers hold the result: never inside a parallel region). */
static int placement_order(int* out, int cap, int* ncores) {
  int cpus[CPU_SETSIZE];
  const int n = allowed_cpus(cpus, CPU_SETSIZE);
  static int prim[CPU_SETSIZE], sec[CPU_SETSIZE], pnode[CPU_SETSIZE], snode[CPU_SETSIZE];
  int np_ = 0, ns = 0, max_node = 0;
  char path[128];
  for (int i = 0; i < n; ++i) {
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpus[i]);
    const int first = read_first_int(path);
    /* the round-robin unit is the L3 domain (a CCD: its link to the memory controllers caps what its cores can stream), numbered so
     * that consecutive domains alternate between the NUMA nodes: domain key = (rank of the L3 inside its node) * 64 + node */
    const int numa = cpu_numa_node(cpus[i]);
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpus[i]);
    const int l3 = read_first_int(path);
    static int l3_first[CPU_SETSIZE], l3_node[CPU_SETSIZE], l3_rank[CPU_SETSIZE], nl3;
    if (i == 0) nl3 = 0;
    int di = -1;
    for (int q = 0; q < nl3; ++q) {
      if (l3_first[q] == (l3 < 0 ? cpus[i] : l3)) di = q;
    }
    if (di < 0) {
      di = nl3++;
      l3_first[di] = l3 < 0 ? cpus[i] : l3;
      l3_node[di] = numa;
      int r = 0;
      for (int q = 0; q < di; ++q) r += l3_node[q] == numa;
      l3_rank[di] = r;
    }
    const int node = (l3_rank[di] % 64) * 64 + (numa % 64); /* ("node" below = this domain key) */
    max_node = node > max_node ? node : max_node;
    if (first < 0 || first == cpus[i]) {
      pnode[np_] = node;
      prim[np_++] = cpus[i];
    } else {
      snode[ns] = node;
      sec[ns++] = cpus[i];
    }
  }
  int k = 0;
  for (int part = 0; part < 2; ++part) {
    const int* src = part == 0 ? prim : sec;
    const int* nodes = part == 0 ? pnode : snode;
    const int cnt = part == 0 ? np_ : ns;
    static int next[4096];  /* per domain key: where its next untaken CPU is searched from */
    for (int i = 0; i < 4096; ++i) next[i] = 0;
    int left = cnt;
    while (left > 0 && k < cap) {
      int took = 0;
      for (int node = 0; node <= max_node && node < 4096 && left > 0 && k < cap; ++node) { /* one CPU of every L3 domain per round */
        int i = next[node];
        while (i < cnt && nodes[i] != node) ++i;
        if (i < cnt) {
          out[k++] = src[i];
          --left;
          ++took;
          next[node] = i + 1;
        } else {
          next[node] = cnt;
        }
      }
      if (!took) break;
    }
    if (part == 0) *ncores = k;
  }
  return k;
}

int32_t orc_physical_core_count(void) {
  int order[CPU_SETSIZE], ncores = 0;
  (void)placement_order(order, CPU_SETSIZE, &ncores);
  return ncores;
}

enum { kStreamPasses = 6 };
double orc_host_stream_read_gbps_placed(int32_t num_threads, size_t bytes_per_thread, int32_t reps, double* per_node, int32_t max_nodes) {
  int order[CPU_SETSIZE], ncores = 0;
  const int ncpus = placement_order(order, CPU_SETSIZE, &ncores);
  if (num_threads < 1) num_threads = 1;
  if (ncpus <= 0) return -1.0;
  if (num_threads > ncpus) num_threads = ncpus;
  const size_t n = bytes_per_thread / 8;
  double* secs = (double*)calloc((size_t)num_threads, sizeof(double));
  int* node_of = (int*)calloc((size_t)num_threads, sizeof(int));
  int64_t* sink = (int64_t*)calloc((size_t)num_threads, sizeof(int64_t));
  if (!secs || !node_of || !sink) return -1.0;
  int bad = 0;
  if (reps < 1) reps = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
#endif
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    const int cpu = order[t % ncpus];
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpu, &one);
    (void)sched_setaffinity(0, sizeof(one), &one);
    node_of[t] = cpu_numa_node(cpu);
    int64_t* x = (int64_t*)map_pages(n * 8);
    int64_t* y = (int64_t*)map_pages(n * 8);
    if (!x || !y) {
      bad = 1;
    } else {
      for (size_t i = 0; i < n; ++i) { /* first touch by the reader, after the pin */
        x[i] = (int64_t)i;
        y[i] = (int64_t)(i ^ 5);
      }
    }
    double best = -1.0;
    for (int r = 0; r < reps + 1; ++r) { /* (the first trip is a warm-up) */
#ifdef _OPENMP
#pragma omp barrier
#endif
      struct timespec t0, t1;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      int64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      for (int pass = 0; pass < kStreamPasses && x && y; ++pass) { /* (several passes per timed trip: a barrier of 256 threads
                                                                      costs milliseconds, one pass over 8 GiB about as much) */
        for (size_t i = 0; i + 4 <= n; i += 4) {
          s0 += x[i] + y[i];
          s1 += x[i + 1] + y[i + 1];
          s2 += x[i + 2] + y[i + 2];
          s3 += x[i + 3] + y[i + 3];
        }
        __asm__ volatile("" : "+r"(s0), "+r"(s1), "+r"(s2), "+r"(s3) : : "memory");
      }
      sink[t] += s0 + s1 + s2 + s3;
#ifdef _OPENMP
#pragma omp barrier
#endif
      clock_gettime(CLOCK_MONOTONIC, &t1); /* (after the barrier: every thread's figure is the slowest thread's) */
      const double sec = ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec)) / kStreamPasses;
      if (r > 0 && (best < 0 || sec < best)) best = sec;
    }
    secs[t] = best;
    if (x) munmap(x, n ? n * 8 : 4096);
    if (y) munmap(y, n ? n * 8 : 4096);
  }
  int cpus_all[CPU_SETSIZE];
  const int nall = allowed_cpus(cpus_all, CPU_SETSIZE);
  { /* back to the allowed set */
    cpu_set_t all;
    CPU_ZERO(&all);
    for (int i = 0; i < ncpus; ++i) CPU_SET(order[i], &all);
    (void)sched_setaffinity(0, sizeof(all), &all);
    (void)nall;
  }
  double worst = 0.0;
  for (int t = 0; t < num_threads; ++t) worst = secs[t] > worst ? secs[t] : worst;
  for (int i = 0; per_node && i < max_nodes; ++i) per_node[i] = 0.0;
  if (!bad && worst > 0 && per_node) {
    for (int t = 0; t < num_threads; ++t) {
      if (node_of[t] < max_nodes) per_node[node_of[t]] += 2.0 * (double)n * 8.0 / worst / 1e9;
    }
  }
  int64_t keep = 0;
  for (int t = 0; t < num_threads; ++t) keep += sink[t];
  free(secs);
  free(node_of);
  free(sink);
  if (bad || worst <= 0) return -1.0;
  (void)keep;
  return 2.0 * (double)n * 8.0 * (double)num_threads / worst / 1e9;
}

/* ---- CPU baseline, JIT-shaped (bench.py `cpu_baseline.variants.jit_shaped`; never part of the product) -------------
 * What HDK's LLVM backend emits for `SELECT key, SUM(val) FROM t GROUP BY key` over a perfect-hash, row-wise layout,
 * written out by hand: the row function is the decoders (fixed_width_int_decode, QE/DecodersImpl.h:30-61) feeding
 * get_group_value_fast (QE/GroupByRuntime.cpp:198-213: off = (key - min) * row_size_quad; the first row of a group
 * writes the key) and agg_sum / agg_sum_skip_val (QE/RuntimeFunctions.cpp:456-461, 612-625), all inlined into one
 * loop per fragment with the plan's constants folded -- no plan interpretation per row, which is what
 * orc_run_plan_range spends its time on.  Execution scheme as in orc_run_plan_parallel: one kernel per fragment on a
 * thread pool with a private output buffer (QE/Execute.cpp:2776-2788), then the slot-wise reduction of the partials
 * (QE/Execute.cpp:1290-1317).  `first_touch` != 0: every fragment is first copied into memory allocated and written
 * by the thread that will scan it (NUMA-local pages, outside the timed region) -- numpy hands over buffers whose
 * pages all sit on the node of the thread that filled them.
 * key / val: 8-byte columns; buf: [entry_count][row_quads] quads, the projected key's slot at quad `key_slot` (or -1)
 * and the SUM slot at quad `sum_slot`, initialised by the caller; returns the seconds of the
 * timed region (scan + reduction) or a negative number on error. */
double orc_c2_jit_shaped(const int64_t* const* keys, const int64_t* const* vals, const int64_t* num_rows,
                         uint64_t num_fragments, int64_t min_key, uint32_t entry_count, uint32_t row_quads,
                         int32_t keyless, int32_t key_slot, int32_t sum_slot, int32_t skip_null, int64_t null_val,
                         const int64_t* init_buffer, int32_t num_threads, int32_t first_touch, int32_t reps,
                         int64_t* out) {
  const size_t rq = row_quads;
  const size_t quads = (size_t)entry_count * rq;
  if (num_threads < 1) num_threads = 1;
  if (reps < 1) reps = 1;
  int64_t** lk = (int64_t**)calloc(num_fragments, sizeof(int64_t*));
  int64_t** lv = (int64_t**)calloc(num_fragments, sizeof(int64_t*));
  /* (private buffers on their own cache lines: see orc_run_plan_parallel) */
  const size_t pstride = (quads + 31) & ~(size_t)31;
  int64_t* partials = NULL;
  if (posix_memalign((void**)&partials, 256, (num_fragments ? num_fragments : 1) * pstride * sizeof(int64_t)) != 0) partials = NULL;
  if (!lk || !lv || !partials) return -1.0;
  int bad = 0;
  int cpus[CPU_SETSIZE];
  const int ncpus = allowed_cpus(cpus, CPU_SETSIZE);
  static int place[CPU_SETSIZE];
  int place_cores = 0;
  const int place_n = first_touch ? placement_order(place, CPU_SETSIZE, &place_cores) : 0; /* (sysfs: once, outside the region) */
  (void)place_cores;
  if (first_touch) { /* threads placed and pinned: the pages they touch below stay local to them */
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
    if (place_n > 0) { /* one thread per physical core first, round robin over the NUMA nodes, SMT siblings last */
      cpu_set_t one;
      CPU_ZERO(&one);
      CPU_SET(place[omp_get_thread_num() % place_n], &one);
      (void)sched_setaffinity(0, sizeof(one), &one);
    }
#endif
  }
  /* same static fragment -> thread map for the copy and for the scans */
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(num_threads)
#endif
  for (int64_t f = 0; f < (int64_t)num_fragments; ++f) {
    if (first_touch) {
      const size_t bytes = (size_t)num_rows[f] * 8;
      lk[f] = (int64_t*)map_pages(bytes);
      lv[f] = (int64_t*)map_pages(bytes);
      if (!lk[f] || !lv[f]) {
        bad = 1;
      } else {
        memcpy(lk[f], keys[f], bytes);
        memcpy(lv[f], vals[f], bytes);
      }
    } else {
      lk[f] = (int64_t*)keys[f];
      lv[f] = (int64_t*)vals[f];
    }
  }
  double best = -1.0;
  for (int r = 0; r < reps && !bad; ++r) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(num_threads)
#endif
    for (int64_t f = 0; f < (int64_t)num_fragments; ++f) {
      int64_t* restrict buf = partials + (size_t)f * pstride;
      memcpy(buf, init_buffer, quads * sizeof(int64_t));
      const int64_t* restrict k = lk[f];
      const int64_t* restrict v = lv[f];
      const int64_t n = num_rows[f];
      if (skip_null) {
        for (int64_t i = 0; i < n; ++i) {
          int64_t* row = buf + (size_t)(k[i] - min_key) * rq; /* get_group_value_fast[_keyless] */
          if (!keyless && row[0] == HDK_EMPTY_KEY_64) row[0] = k[i];
          if (key_slot >= 0) row[key_slot] = k[i]; /* agg_id: the projected key's own slot */
          const int64_t x = v[i];
          if (x != null_val) { /* agg_sum_skip_val */
            row[sum_slot] = row[sum_slot] == null_val ? x : (int64_t)((uint64_t)row[sum_slot] + (uint64_t)x);
          }
        }
      } else {
        for (int64_t i = 0; i < n; ++i) {
          int64_t* row = buf + (size_t)(k[i] - min_key) * rq;
          if (!keyless && row[0] == HDK_EMPTY_KEY_64) row[0] = k[i];
          if (key_slot >= 0) row[key_slot] = k[i];
          row[sum_slot] = (int64_t)((uint64_t)row[sum_slot] + (uint64_t)v[i]); /* agg_sum */
        }
      }
    }
    memcpy(out, init_buffer, quads * sizeof(int64_t));
    for (uint64_t f = 0; f < num_fragments; ++f) { /* reduceOneSlot over the partials, in fragment order */
      const int64_t* p = partials + (size_t)f * pstride;
      for (uint32_t e = 0; e < entry_count; ++e) {
        /* isEmptyEntry (RS/ResultSetStorage.cpp:439-547): the key, or -- keyless -- the slot idx_target_as_key names
         * still at its init value (`keyless` = that quad + 1) */
        if (keyless ? p[(size_t)e * rq + keyless - 1] == init_buffer[keyless - 1] : p[(size_t)e * rq] == HDK_EMPTY_KEY_64) continue;
        if (!keyless) out[(size_t)e * rq] = p[(size_t)e * rq];
        if (key_slot >= 0) out[(size_t)e * rq + key_slot] = p[(size_t)e * rq + key_slot];
        const int64_t x = p[(size_t)e * rq + sum_slot];
        int64_t* s = out + (size_t)e * rq + sum_slot;
        if (skip_null) {
          if (x != null_val) *s = *s == null_val ? x : (int64_t)((uint64_t)*s + (uint64_t)x);
        } else {
          *s = (int64_t)((uint64_t)*s + (uint64_t)x);
        }
      }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (best < 0 || sec < best) best = sec;
  }
  if (first_touch) {
    for (uint64_t f = 0; f < num_fragments; ++f) {
      const size_t bytes = (size_t)num_rows[f] * 8;
      if (lk[f]) munmap(lk[f], bytes ? bytes : 4096);
      if (lv[f]) munmap(lv[f], bytes ? bytes : 4096);
    }
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
    unpin(cpus, ncpus); /* the pool's threads (and the caller) go back to the whole allowed set */
#endif
    unpin(cpus, ncpus);
  }
  free(lk);
  free(lv);
  free(partials);
  return bad ? -1.0 : best;
}
