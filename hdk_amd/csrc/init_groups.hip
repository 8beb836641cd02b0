// init_groups.hip -- output-buffer initialisation (kernel #1 of every group-by launch).
//
// New kernels for the contract of init_group_by_buffer_on_device /
// init_columnar_group_by_buffer_on_device (reference QE/GpuInitGroups.h:23-48; CUDA bodies
// QE/GpuInitGroups.cu:17-232 fill one entry per thread with scalar stores).  Here the fill is a
// streaming write: each lane writes whole 16-B quads of the destination in row order, so the 3.2 GB
// C5 table initialises at the HBM write rate.  The byte image produced is identical.
#include "host_common.h"
#include "device_common.h"

namespace hdk {

constexpr int kInitBlock = 256;

// Row-wise: the buffer is `entry_count` rows of `row_size_quad` int64 words:
//   [keys: key_count x key_width, padded to 8][values: init_vals[0..]]
// keyless: rows are values only, replicated `warp_size` times (interleaved bins are never
// requested by this library, warp_size arrives as 1).
__global__ __launch_bounds__(kInitBlock) void k_init_row_wise(int64_t* __restrict__ buf,
                                                              const int64_t* __restrict__ init_vals,
                                                              uint64_t total_quads, uint32_t row_size_quad,
                                                              uint32_t key_count, uint32_t key_width,
                                                              int keyless) {
  const uint32_t keys_quads = keyless ? 0 : (key_count * key_width + 7) / 8;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kInitBlock;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kInitBlock + threadIdx.x; i < total_quads; i += stride) {
    const int64_t v = init_row_quad(static_cast<uint32_t>(i % row_size_quad), keys_quads, key_count, key_width, init_vals);
    buf[i] = v;
  }
}

// the same fill for a buffer known only through GROUPBY_BUF (a device array of pointers: [0] is the buffer):
// hdk_hip_launch with HDK_HIP_LAUNCH_INIT_OUTPUT
__global__ __launch_bounds__(kInitBlock) void k_init_row_wise_indirect(int64_t* const* __restrict__ groupby_buf,
                                                                       const int64_t* __restrict__ init_vals,
                                                                       uint64_t total_quads, uint32_t row_size_quad,
                                                                       uint32_t key_count, uint32_t key_width,
                                                                       int keyless) {
  int64_t* __restrict__ buf = groupby_buf[0];
  const uint32_t keys_quads = keyless ? 0 : (key_count * key_width + 7) / 8;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kInitBlock;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kInitBlock + threadIdx.x; i < total_quads; i += stride) {
    buf[i] = init_row_quad(static_cast<uint32_t>(i % row_size_quad), keys_quads, key_count, key_width, init_vals);
  }
}

int32_t launch_init_row_wise_indirect(int64_t* const* groupby_buf, const int64_t* init_vals, uint32_t entry_count,
                                      uint32_t key_count, uint32_t key_width, uint32_t row_size_quad, int keyless,
                                      const hdk_hip_device_properties* props, hipStream_t s) {
  const uint64_t total_quads = static_cast<uint64_t>(entry_count) * row_size_quad;
  if (total_quads == 0) {
    return HDK_HIP_OK;
  }
  uint64_t blocks = (total_quads + kInitBlock - 1) / kInitBlock;
  const uint64_t cap = static_cast<uint64_t>(props->num_cu) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_init_row_wise_indirect, dim3(static_cast<unsigned>(blocks)), dim3(kInitBlock), 0, s, groupby_buf,
                     init_vals, total_quads, row_size_quad, key_count, key_width, keyless);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// The bytes between the end of a narrow column and the next 8-byte boundary.  The reference leaves them as the
// allocator handed them out (nothing reads them); zeroing them makes the buffer image deterministic, so that whole
// buffers can be compared and checksummed.
__device__ inline void zero_gap(int8_t* buf, size_t from, size_t to) {
  for (size_t b = from; b < to; ++b) {
    buf[b] = 0;
  }
}

// Columnar: a sequence of columns, each `entry_count` elements of `width` bytes, 8-byte aligned
// (init_columnar_group_by_buffer_gpu_impl, QE/GpuInitGroups.cu:17-108).  `init_vals` and
// `col_sizes` are DEVICE arrays, as in the reference.
__global__ __launch_bounds__(kInitBlock) void k_init_columnar(int8_t* __restrict__ buf,
                                                              const int64_t* __restrict__ init_vals,
                                                              uint32_t entry_count, uint32_t key_count,
                                                              uint32_t agg_col_count,
                                                              const int8_t* __restrict__ col_sizes,
                                                              int need_padding, int keyless, int key_size) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kInitBlock;
  const uint64_t start = static_cast<uint64_t>(blockIdx.x) * kInitBlock + threadIdx.x;
  size_t off = 0;
  if (!keyless) {
    for (uint32_t k = 0; k < key_count; ++k) {
      int8_t* col = buf + off;
      for (uint64_t i = start; i < entry_count; i += stride) {
        switch (key_size) {
          case 1: reinterpret_cast<int8_t*>(col)[i] = HDK_EMPTY_KEY_8; break;
          case 2: reinterpret_cast<int16_t*>(col)[i] = HDK_EMPTY_KEY_16; break;
          case 4: reinterpret_cast<int32_t*>(col)[i] = HDK_EMPTY_KEY_32; break;
          default: reinterpret_cast<int64_t*>(col)[i] = HDK_EMPTY_KEY_64; break;
        }
      }
      const size_t end = off + static_cast<size_t>(entry_count) * key_size;
      off = (end + 7) & ~static_cast<size_t>(7);
      if (start == 0) {
        zero_gap(buf, end, off);
      }
    }
  }
  int init_idx = 0;
  for (uint32_t c = 0; c < agg_col_count; ++c) {
    if (need_padding) {
      const size_t end = off;
      off = (off + 7) & ~static_cast<size_t>(7);
      if (start == 0) {
        zero_gap(buf, end, off);
      }
    }
    const int w = col_sizes[c];
    if (w == 0) {
      continue;
    }
    const int64_t v = init_vals[init_idx++];
    int8_t* col = buf + off;
    for (uint64_t i = start; i < entry_count; i += stride) {
      switch (w) {
        case 1: reinterpret_cast<int8_t*>(col)[i] = static_cast<int8_t>(v); break;
        case 2: reinterpret_cast<int16_t*>(col)[i] = static_cast<int16_t>(v); break;
        case 4: reinterpret_cast<int32_t*>(col)[i] = static_cast<int32_t>(v); break;
        default: reinterpret_cast<int64_t*>(col)[i] = v; break;
      }
    }
    off += static_cast<size_t>(entry_count) * w;
  }
  if (start == 0) {
    zero_gap(buf, off, (off + 7) & ~static_cast<size_t>(7));  // the buffer is a whole number of int64 words
  }
}

static inline size_t align8h(size_t x) { return (x + 7) & ~static_cast<size_t>(7); }

}  // namespace hdk

using namespace hdk;

extern "C" int32_t hdk_hip_init_group_by_buffer(int64_t* groups_buffer, const int64_t* init_vals,
                                                uint32_t groups_buffer_entry_count, uint32_t key_count,
                                                uint32_t key_width, uint32_t row_size_quad,
                                                int32_t keyless, int8_t warp_size, size_t block_size_x,
                                                size_t grid_size_x, int32_t device_id, void* stream) {
  (void)block_size_x;
  (void)grid_size_x;
  HDK_REQUIRE(groups_buffer && init_vals, "NULL buffer");
  HDK_REQUIRE(row_size_quad > 0, "row_size_quad must be positive");
  HDK_REQUIRE(keyless || key_width == 4 || key_width == 8, "key_width must be 4 or 8");
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  const uint64_t total_quads = static_cast<uint64_t>(groups_buffer_entry_count) * row_size_quad *
                               (keyless ? (warp_size > 0 ? warp_size : 1) : 1);
  if (total_quads == 0) {
    return HDK_HIP_OK;
  }
  const hdk_hip_device_properties* props = device_props(device_id);
  uint64_t blocks = (total_quads + kInitBlock - 1) / kInitBlock;
  const uint64_t cap = static_cast<uint64_t>(props->num_cu) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_init_row_wise, dim3(static_cast<unsigned>(blocks)), dim3(kInitBlock), 0, s,
                     groups_buffer, init_vals, total_quads, row_size_quad, key_count, key_width,
                     keyless ? 1 : 0);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

extern "C" int32_t hdk_hip_init_columnar_group_by_buffer(
    int64_t* groups_buffer, const int64_t* init_vals, uint32_t groups_buffer_entry_count,
    uint32_t key_count, uint32_t agg_col_count, const int8_t* col_sizes, int32_t need_padding,
    int32_t keyless, int8_t key_size, size_t block_size_x, size_t grid_size_x, int32_t device_id,
    void* stream) {
  (void)block_size_x;
  (void)grid_size_x;
  HDK_REQUIRE(groups_buffer && init_vals && (col_sizes || agg_col_count == 0), "NULL buffer");
  hipStream_t s;
  const int32_t st = device_enter(device_id, stream, &s);
  if (st) return st;
  const size_t n = groups_buffer_entry_count;
  if (n == 0) {
    return HDK_HIP_OK;
  }
  const hdk_hip_device_properties* props = device_props(device_id);
  uint64_t blocks = (n + kInitBlock - 1) / kInitBlock;
  const uint64_t cap = static_cast<uint64_t>(props->num_cu) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_init_columnar, dim3(static_cast<unsigned>(blocks)), dim3(kInitBlock), 0, s,
                     reinterpret_cast<int8_t*>(groups_buffer), init_vals, groups_buffer_entry_count,
                     key_count, agg_col_count, col_sizes, need_padding ? 1 : 0, keyless ? 1 : 0,
                     static_cast<int>(key_size));
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}
